// LDS-staged gather, translation unit 2 of 5 (gather_lds.h): the entropy pass with fp16 windows (MVS_GATHER_F16).
#include "gather_lds.h"

namespace mvs {

template <int DT, int NOCT, int NS, bool TILED>
static int gl_launch_entropy_w16_t(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int D, int H, int W, int vb,
                                   int ve, hipStream_t st) {
    return gl_launch_entropy_t<DT, NOCT, NS, TILED, false, true>(feat, hom, hyp, ent, B, V, D, H, W, vb, ve, st);
}

int gl_launch_entropy_w16(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H, int W,
                          int vb, int ve, hipStream_t st) {
    GL_DISPATCH(gl_launch_entropy_w16_t, feat, hom, hyp, ent, B, V, D, H, W, vb, ve, st);
}

}  // namespace mvs
