// LDS-staged gather, translation unit 4 of 5 (gather_lds.h): the aggregation pass with fp32 windows (fp32 / split / fp16 volume out).
#include "gather_lds.h"

namespace mvs {

int gl_launch_aggregate_w16(const void* feat, int dtype, int layout, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                            int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st);      // gather_lds_aggregate_w16_kernels.hip

int gl_launch_aggregate(const void* feat, int dtype, int layout, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                        int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st) {
    if ((normalise & 4) && gl_window_f16_enabled())            // fp16 volume: fp16 windows
        return gl_launch_aggregate_w16(feat, dtype, layout, hom, hyp, vis, vol, vis_sum, normalise, B, V, C, D, H, W, vb, ve, st);
    GL_DISPATCH(gl_launch_aggregate_t, feat, hom, hyp, vis, vol, vis_sum, normalise, B, V, D, H, W, vb, ve, st);
}

}  // namespace mvs

namespace mvs { MVS_DEFINE_SAT_READER(sat_read_gather_agg) }
