// LDS-staged gather, translation unit 5 of 5 (gather_lds.h): the aggregation pass with fp16 windows (the fp16 volume formats).
#include "gather_lds.h"

namespace mvs {

template <int DT, int NOCT, int NS, bool TILED>
static int gl_launch_aggregate_w16_t(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                                     int normalise, int B, int V, int D, int H, int W, int vb, int ve, hipStream_t st) {
    return gl_launch_aggregate_t<DT, NOCT, NS, TILED, true>(feat, hom, hyp, vis, vol, vis_sum, normalise, B, V, D, H, W, vb, ve, st);
}

int gl_launch_aggregate_w16(const void* feat, int dtype, int layout, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                            int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st) {
    GL_DISPATCH(gl_launch_aggregate_w16_t, feat, hom, hyp, vis, vol, vis_sum, normalise, B, V, D, H, W, vb, ve, st);
}

}  // namespace mvs

namespace mvs { MVS_DEFINE_SAT_READER(sat_read_gather_agg16) }
