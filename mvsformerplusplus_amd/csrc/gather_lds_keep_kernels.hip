// LDS-staged gather, translation unit 3 of 6 (gather_lds.h): the entropy pass that keeps the per-view group correlations as fp16, and
// the streaming pass 2 over kept correlations (fp16, or fp32 from gather_lds_keep32_kernels.hip).
#include "gather_lds.h"

namespace mvs {

// ------------------------------------------------------------------------------------------------
// pass 2, streaming form: volume = sum_v vis_v * corr_v / (sum_v vis_v + 1e-6)     cost_volume.py:97-101
// corr [B, V-1, D, HW, 8] fp16 (gl_entropy_kernel<KEEP>), vis [B, V-1, HW] -> vol [B, D, HW, 8] fp16.
// grid = (ceil(HW / 256), ceil(D / 4), B); a thread owns one pixel and four planes (a view's visibility weight is loaded
// once for the four), 16-byte loads and stores.
// ------------------------------------------------------------------------------------------------
// FMT: the volume's format - MVS_VOLUME_F16 (fp16 octet, 16 B per voxel), MVS_VOLUME_SPLIT ([hi x8 | lo x8] bf16 of the bf16x3 U-Net, 32 B) or
// MVS_VOLUME_F32 (32 B); CF32: the kept correlations are fp32 octets (MVS_CORR_F32) instead of fp16 (MVS_CORR_F16) - the stage's
// REGULARISER format is independent of the gather's
template <int FMT, bool CF32>
__global__ __launch_bounds__(256) void corr_aggregate_kernel(const void* __restrict__ corr, const float* __restrict__ vis,
                                                             void* __restrict__ vol, int NV, int D, unsigned HW) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= HW) return;
    const int b = (int)blockIdx.z, d0 = (int)blockIdx.y * 4;
    const float* vp = vis + (size_t)b * NV * HW + p;
    const size_t cp = (size_t)b * NV * D * HW + p;
    float acc[4][8];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[dd][g] = 0.0f;
    float vsum = 0.0f;
#pragma unroll 2
    for (int v = 0; v < NV; ++v) {
        const float w = vp[(size_t)v * HW];
        vsum += w;                                                                                   // cost_volume.py:98
        const size_t cv = cp + (size_t)v * D * HW;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = d0 + dd < D ? d0 + dd : D - 1;
            const size_t vox = cv + (size_t)(unsigned)d * HW;
            if constexpr (CF32) {
                const f32x4* c4 = reinterpret_cast<const f32x4*>(corr) + vox * 2;
                const f32x4 c0 = c4[0], c1 = c4[1];
#pragma unroll
                for (int g = 0; g < 4; ++g) { acc[dd][g] += w * c0[g]; acc[dd][4 + g] += w * c1[g]; }  // cost_volume.py:97
            } else {
                const h8 c = reinterpret_cast<const h8*>(corr)[vox];
#pragma unroll
                for (int g = 0; g < 8; ++g) acc[dd][g] += w * (float)c[g];                            // cost_volume.py:97
            }
        }
    }
    const float rdenom = 1.0f / (vsum + 1e-6f);                                                      // cost_volume.py:101
    float sat_amax = 0.0f;
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        if (d0 + dd >= D) continue;
        float r[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) r[g] = acc[dd][g] * rdenom;
        const size_t vox = (size_t)b * D * HW + (size_t)(unsigned)(d0 + dd) * HW + p;
        if constexpr (FMT == MVS_VOLUME_F16) {
            h8 hv;
#pragma unroll
            for (int g = 0; g < 8; ++g) hv[g] = (_Float16)fminf(fmaxf(r[g], -65504.0f), 65504.0f);
            sat::track(sat_amax, r[0], r[1], r[2], r[3]);
            sat::track(sat_amax, r[4], r[5], r[6], r[7]);
            reinterpret_cast<h8*>(vol)[vox] = hv;
        } else {
            f32x4* o = reinterpret_cast<f32x4*>(vol) + vox * 2;
            if constexpr (FMT == MVS_VOLUME_SPLIT) {             // [hi x8 | lo x8] bf16: the same 32 bytes (conv_bf16x3_kernels.hip)
                unsigned hw[4], lw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint16_t h0 = from_f32<uint16_t>(r[2 * j]), h1 = from_f32<uint16_t>(r[2 * j + 1]);
                    const uint16_t l0 = from_f32<uint16_t>(r[2 * j] - to_f32(h0)), l1 = from_f32<uint16_t>(r[2 * j + 1] - to_f32(h1));
                    hw[j] = (unsigned)h0 | ((unsigned)h1 << 16);
                    lw[j] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
                o[0] = f32x4{__builtin_bit_cast(float, hw[0]), __builtin_bit_cast(float, hw[1]), __builtin_bit_cast(float, hw[2]), __builtin_bit_cast(float, hw[3])};
                o[1] = f32x4{__builtin_bit_cast(float, lw[0]), __builtin_bit_cast(float, lw[1]), __builtin_bit_cast(float, lw[2]), __builtin_bit_cast(float, lw[3])};
            } else {
                o[0] = f32x4{r[0], r[1], r[2], r[3]};
                o[1] = f32x4{r[4], r[5], r[6], r[7]};
            }
        }
    }
    if constexpr (FMT == MVS_VOLUME_F16) sat::commit(sat_amax);
}

// KEEP form, fp16 correlations: every depth-chunk geometry (round 5 instantiates NS = 1, D <= 4, too: whether the streamed pass 2 pays there -
// 16 B per voxel and view written and read against a second gather - is the CALLER's policy, StageNet.keep_min_depth)
template <int DT, int NOCT, int NS, bool TILED>
static int gl_launch_entropy_keep_t(const void* feat, const float* hom, const float* hyp, float* ent, void* corr, int B, int V, int D, int H, int W,
                                    hipStream_t st) {
    return gl_launch_entropy_t<DT, NOCT, NS, TILED, true, true>(feat, hom, hyp, ent, B, V, D, H, W, 1, V, st, corr);
}

int gl_launch_entropy_keep32(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, void* corr, int B, int V, int C,
                             int D, int H, int W, hipStream_t st);      // gather_lds_keep32_kernels.hip

int gl_launch_entropy_keep(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, void* corr, int corr_format, int B,
                           int V, int C, int D, int H, int W, hipStream_t st) {
    if (corr_format == MVS_CORR_F32) return gl_launch_entropy_keep32(feat, dtype, layout, hom, hyp, ent, corr, B, V, C, D, H, W, st);
    GL_DISPATCH(gl_launch_entropy_keep_t, feat, hom, hyp, ent, corr, B, V, D, H, W, st);
}

template <bool CF32>
static int launch_corr_aggregate_t(const void* c, const float* vis, void* vol, int volume_format, int B, int V, int D, int H, int W, hipStream_t st) {
    const unsigned HW = (unsigned)H * (unsigned)W;
    const dim3 grid(ceil_div(HW, 256), ceil_div(D, 4), B);
    switch (volume_format) {
        case MVS_VOLUME_F16: hipLaunchKernelGGL((corr_aggregate_kernel<MVS_VOLUME_F16, CF32>), grid, dim3(256), 0, st, c, vis, vol, V - 1, D, HW); break;
        case MVS_VOLUME_SPLIT: hipLaunchKernelGGL((corr_aggregate_kernel<MVS_VOLUME_SPLIT, CF32>), grid, dim3(256), 0, st, c, vis, vol, V - 1, D, HW); break;
        default: hipLaunchKernelGGL((corr_aggregate_kernel<MVS_VOLUME_F32, CF32>), grid, dim3(256), 0, st, c, vis, vol, V - 1, D, HW); break;
    }
    return check_launch("corr_aggregate_kernel");
}

int launch_corr_aggregate(const void* corr, int corr_format, const float* vis, void* vol, int volume_format, int B, int V, int D, int H, int W,
                          hipStream_t st) {
    return corr_format == MVS_CORR_F32 ? launch_corr_aggregate_t<true>(corr, vis, vol, volume_format, B, V, D, H, W, st)
                                       : launch_corr_aggregate_t<false>(corr, vis, vol, volume_format, B, V, D, H, W, st);
}

}  // namespace mvs

namespace mvs { MVS_DEFINE_SAT_READER(sat_read_gather_keep) }
