// LDS-staged gather, translation unit 6 of 6 (gather_lds.h): the entropy pass that keeps the per-view group correlations as FP32 octets
// (MVS_CORR_F32, round 5) with fp32 source windows - the exact form of the kept-correlation pass 1: corr_aggregate_kernel<., true> then
// produces the volume the second gather (gl_aggregate_kernel with fp32 windows) would have written, at 32 B of streamed traffic per
// voxel and view instead of a second window staging + tap gather.  The coarse cascade stages (ndepth > model_th) of the default
// precision policy run it: their depth schedules the next stage's hypotheses, and the fp16 correlations' 2^-11 rounding is what the
// cascade amplifies on ill-conditioned hypothesis ranges (scripts/study_wide_range_gather.py).
#include "gather_lds.h"

namespace mvs {

template <int DT, int NOCT, int NS, bool TILED>
static int gl_launch_entropy_keep32_t(const void* feat, const float* hom, const float* hyp, float* ent, void* corr, int B, int V, int D, int H, int W,
                                      hipStream_t st) {
    if constexpr (NS == 1) {
        set_error("mvs_warp_corr_entropy_keep_fwd: D <= 4 is not built (the second gather is the faster pass 2 there)");
        return MVS_ERR_UNSUPPORTED;
    } else {
        return gl_launch_entropy_t<DT, NOCT, NS, TILED, true, false>(feat, hom, hyp, ent, B, V, D, H, W, 1, V, st, corr);
    }
}

int gl_launch_entropy_keep32(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, void* corr, int B, int V, int C,
                             int D, int H, int W, hipStream_t st) {
    GL_DISPATCH(gl_launch_entropy_keep32_t, feat, hom, hyp, ent, corr, B, V, D, H, W, st);
}

}  // namespace mvs
