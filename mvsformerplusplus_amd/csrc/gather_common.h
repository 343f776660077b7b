// Helpers shared by the two LDS-staged forms of the gather passes (gather_lds_kernels.hip: block-level windows, round 2;
// gather_wave_kernels.hip: wave-autonomous windows, round 3).
#pragma once
#include <type_traits>
#include "mvs_common.h"

namespace mvs {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 gl_as_vec(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned gl_as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// Wave-wide packed-u16 min / max with DPP (no LDS round trips): quad_perm, row_half_mirror, row_mirror leave every lane
// of a 16-lane row with the row's result, row_bcast:15 / :31 fold the four rows; the full result lands in lane 63.
template <bool MAX>
__device__ __forceinline__ u16x2 gl_wave_reduce(u16x2 v) {
#define GL_DPP_STEP(CTRL, ROWMASK)                                                                                             \
    {                                                                                                                          \
        const int cur = (int)gl_as_u32(v);                                                                                     \
        const u16x2 o = gl_as_vec((unsigned)__builtin_amdgcn_update_dpp(cur, cur, CTRL, ROWMASK, 0xf, false));                 \
        v = MAX ? __builtin_elementwise_max(v, o) : __builtin_elementwise_min(v, o);                                           \
    }
    GL_DPP_STEP(0xB1, 0xf)       // quad_perm:[1,0,3,2]
    GL_DPP_STEP(0x4E, 0xf)       // quad_perm:[2,3,0,1]
    GL_DPP_STEP(0x141, 0xf)      // row_half_mirror
    GL_DPP_STEP(0x140, 0xf)      // row_mirror
    GL_DPP_STEP(0x142, 0xa)      // row_bcast:15 -> rows 1, 3
    GL_DPP_STEP(0x143, 0xc)      // row_bcast:31 -> rows 2, 3
#undef GL_DPP_STEP
    return v;
}

// Feature layouts (include/mvs_hip.h MVS_LAYOUT_*):
//   TILED = false  planar NCHW [C][H*W]: channel c of position p at base[c * HW + p]            (what the reference's FPN emits)
//   TILED = true   octet-tiled channel-last [C/8][H*W][8]: the 8 channels of an octet are one 32-byte (fp32) / 16-byte
//                  (bf16, fp16) run - the hand-off layout of SURVEY.md section 8f #4: staging a window is a copy of whole
//                  runs (one or two 16-byte loads per position instead of eight 4- / 2-byte loads)
// `base` points at the octet (planar: channel 8*o of the view, tiled: tile plane o of the view).
template <bool TILED, typename T>
__device__ __forceinline__ void gl_load8(const T* __restrict__ base, unsigned HW, unsigned p, float* v) {
    if (!TILED) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = to_f32(base[(unsigned)c * HW + p]);
    } else if (sizeof(T) == 4) {
        const f32x4* q = reinterpret_cast<const f32x4*>(base) + (size_t)p * 2;
        const f32x4 a = q[0], b = q[1];
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        typedef T t8 __attribute__((ext_vector_type(8)));
        const t8 a = *(reinterpret_cast<const t8*>(base) + p);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = to_f32((T)a[c]);
    }
}
// offset (in elements of T) of octet o of a view: planar = 8 channel planes, tiled = one [HW][8] tile plane: the same number
__device__ __forceinline__ unsigned gl_octet_offset(int o, unsigned HW) { return (unsigned)o * 8u * HW; }

// softmax over depth and its entropy (cost_volume.py:91-92) with the hardware exp2 / log2 (1 ulp each; the exponent's
// argument is <= 0, so the scaling by log2(e) costs ~|x| * 6e-8 relative)
__device__ __forceinline__ void gl_softmax_entropy_store(const float* sim, int stride, int D, float* dst) {
    const float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, sim[d * stride]);
    float den = 0.0f;
    for (int d = 0; d < D; ++d) den += __builtin_amdgcn_exp2f((sim[d * stride] - m) * LOG2E);
    const float rden = 1.0f / den;
    float ent = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = __builtin_amdgcn_exp2f((sim[d * stride] - m) * LOG2E) * rden;
        ent -= pr * (__builtin_amdgcn_logf(pr + 1e-7f) * LN2);  // cost_volume.py:92
    }
    *dst = ent;
}

__device__ __forceinline__ Homography gl_load_homography(const float* p) {
    Homography hm;
#pragma unroll
    for (int i = 0; i < 9; ++i) hm.r[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) hm.t[i] = p[9 + i];
    return hm;
}

}  // namespace mvs
