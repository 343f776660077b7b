// Producer-side feature emitter: the LAST convolution of the reference's feature side written
// straight into the layout the gather passes read.
//
// Reference (restated, never copied): the per-stage feature maps that reach StageNet come out of 3x3 Conv2d layers -
//   * FMT_with_pathway.smooth_1 / _2 / _3 (models/FMT.py:195-197: Conv2d(C, C, 3, padding=1, bias=False), C = 32 / 16 / 8) for
//     stages 2-4 of the shipped model (models/FMT.py:231-233, torch.stack over views into [B,V,C,H,W]);
//   * FPNDecoder.out1 / out2 / out3 (models/module.py:257-270: Conv2d(64, C, 3, padding=1) + BatchNorm2d + Swish) where a model feeds
//     the FPN heads to the cost volume directly (casmvs-style networks).
// Both emit planar NCHW; StageNet upcasts per view (cost_volume.py:67) and the gather kernels then read it with one strided plane per
// channel - or, in the hand-off layout [B,V,C/8,H,W,8] (gather_lds.h TILED), one 16- / 32-byte run per pixel and channel octet.  Round 2
// built the consumer side plus a converter pass (mvs_pack_features: one extra read + write of every feature map).  This kernel is the
// producer side: the convolution's epilogue adds the bias, applies the activation (none | Swish, BatchNorm folded on the host), rounds
// to the hand-off dtype (bf16 like the reference's autocast, test.py:250; or fp16 / fp32) and stores octet tiles directly - the
// planar [B,V,C,H,W] tensor is never materialised and no transpose pass runs.
//
// Form: implicit GEMM on v_mfma_f32_16x16x32_bf16 with the three-term split-bf16 product (fp32-equivalent, like the regularisers'
// "bf16x3" mode: the input may be fp32 and the result must not depend on it being rounded), D[cout, pixel] += W[cout, k] X[k, pixel],
// k = (tap, cin).  A workgroup owns 4 rows x 64 columns of output pixels (one row per wave, four 16-pixel column blocks per wave);
// the input tile + halo (6 x 66 pixels) is read from the planar input with lane-consecutive loads along x (one plane per channel:
// coalesced), split once into hi | lo bf16 and kept in LDS channel-last ([octet plane][pixel][hi x8 | lo x8]) so that a B operand
// (8 consecutive input channels of one pixel) is one ds_read_b128 per half; packed weights (packing.pack_conv_weights_bf16x3 with kd = 1)
// come per step from global / L2 in lane order.  Input channels are staged in passes of <= 32.
// Roofline: HBM (Cin x 4 B in, Cout x 2 B out per pixel; 64 -> 8 at 1152 x 1536: 453 MB + 28 MB per view) for Cin <= 16, MFMA-issue
// for the 64-channel FPN heads (9.2 KFLOP x 3 terms per pixel).  Not part of the timed path (features are its inputs).
#include "mvs_common.h"
#include "split_format.h"

namespace mvs {

constexpr int FC_TH = 4, FC_TW = 64, FC_IH = FC_TH + 2, FC_IW = FC_TW + 2, FC_NPIX = FC_IH * FC_IW;   // 6 x 66 = 396 staged pixels
constexpr int FC_PLANE = FC_NPIX * 32 + 32;            // bytes of one octet plane (+ one slot: planes start on different bank rows)

__device__ __forceinline__ float fc_load(const void* p, int dtype, size_t i) {
    if (dtype == MVS_DTYPE_F32) return static_cast<const float*>(p)[i];
    if (dtype == MVS_DTYPE_BF16) return to_f32(static_cast<const uint16_t*>(p)[i]);
    return (float)static_cast<const _Float16*>(p)[i];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv2d3x3_tiles_kernel(const void* __restrict__ x, int in_dtype, const void* __restrict__ wp,
                                                              const float* __restrict__ bias, int act, void* __restrict__ out, int out_dtype,
                                                              int H, int W, long long in_batch_stride, long long out_batch_stride,
                                                              int tiles_x, int ntiles) {
    constexpr int CH = CIN < 32 ? CIN : 32, NPASS = CIN / CH, OPT = CH / 8, NOCT = 9 * OPT, NSTEP = (NOCT + 3) / 4;
    constexpr int MREP = (COUT + 15) / 16, NREP = FC_TW / 16;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles), n = (int)blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * FC_TH, x0 = tx * FC_TW;
    const size_t HW = (size_t)H * (size_t)W;
    const size_t xin = (size_t)n * (size_t)in_batch_stride;

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();                            // every wave has read the previous pass's image
        // ---- stage: one work-item = one staged pixel x one channel octet; consecutive work-items = consecutive pixels of a row ----
        for (int e = tid; e < FC_NPIX * OPT; e += 256) {
            const int oc = e / FC_NPIX, pix = e - oc * FC_NPIX;
            const int iy = pix / FC_IW, ix = pix - iy * FC_IW;
            const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            float v[8];
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {         // zero padding (padding=1) outside the image
                const size_t base = xin + (size_t)(pass * CH + oc * 8) * HW + (size_t)gy * W + gx;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fc_load(x, in_dtype, base + (size_t)k * HW);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.0f;
            }
            bf16x8 hi, lo;
            split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
            char* dst = ldsb + oc * FC_PLANE + pix * 32;
            *reinterpret_cast<bf16x8*>(dst) = hi;
            *reinterpret_cast<bf16x8*>(dst + 16) = lo;
        }
        __syncthreads();
        // ---- contract: step = four channel octets (one per lane group), octet q = 4 step + g -> (tap, oc) = divmod(q, OPT) ----
        const bf16x8* wq = reinterpret_cast<const bf16x8*>(wp) + (size_t)pass * NSTEP * MREP * 2 * 64 + lane;
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            const int q = 4 * step + g;
            const bool live = q < NOCT;                           // the last step may run past the 9 x OPT octets: zero operand (the packed weights are zero there too)
            const int tap = live ? q / OPT : 0, oc = live ? q - tap * OPT : 0;
            const int ky = tap / 3, kx = tap - ky * 3;
            const char* src = ldsb + oc * FC_PLANE + ((wave + ky) * FC_IW + li + kx) * 32;
            bf16x8 ah[MREP], al[MREP], bh[NREP], bl[NREP];
#pragma unroll
            for (int mb = 0; mb < MREP; ++mb) {
                ah[mb] = wq[(size_t)((step * MREP + mb) * 2 + 0) * 64];
                al[mb] = wq[(size_t)((step * MREP + mb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int nb = 0; nb < NREP; ++nb) {
                bf16x8 h = *reinterpret_cast<const bf16x8*>(src + nb * 16 * 32);
                bf16x8 l = *reinterpret_cast<const bf16x8*>(src + nb * 16 * 32 + 16);
                if (!live) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { h[k] = (__bf16)0.0f; l[k] = (__bf16)0.0f; }
                }
                bh[nb] = h;
                bl[nb] = l;
            }
#pragma unroll
            for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
                for (int nb = 0; nb < NREP; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: lane (pixel li, group g) holds output channels 16 mb + 4 g .. + 3 of its pixel = half an octet: bias, activation,
    //      hand-off dtype, one 8-byte (16-bit dtypes) / 16-byte (fp32) store into [C/8][H][W][8] ----
    const int y = y0 + wave;
    if (y >= H) return;
    const size_t obase = (size_t)n * (size_t)out_batch_stride;
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int xx = x0 + nb * 16 + li;
        if (xx >= W) continue;
#pragma unroll
        for (int mb = 0; mb < MREP; ++mb) {
            const int co = 16 * mb + 4 * g;
            if (co >= COUT) continue;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = acc[mb][nb][k] + (bias ? bias[co + k] : 0.0f);
                if (act == 1) t = t / (1.0f + expf(-t));          // Swish (module.py Swish: x * sigmoid(x))
                v[k] = t;
            }
            const size_t o = obase + (((size_t)(co >> 3) * H + y) * W + xx) * 8 + (co & 7);
            if (out_dtype == MVS_DTYPE_F32) {
                *reinterpret_cast<float4*>(static_cast<float*>(out) + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (out_dtype == MVS_DTYPE_BF16) {
                typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<u16x4*>(static_cast<uint16_t*>(out) + o) =
                    u16x4{from_f32<uint16_t>(v[0]), from_f32<uint16_t>(v[1]), from_f32<uint16_t>(v[2]), from_f32<uint16_t>(v[3])};
            } else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const float m = 65504.0f;
                *reinterpret_cast<h4*>(static_cast<_Float16*>(out) + o) =
                    h4{(_Float16)fminf(fmaxf(v[0], -m), m), (_Float16)fminf(fmaxf(v[1], -m), m), (_Float16)fminf(fmaxf(v[2], -m), m),
                       (_Float16)fminf(fmaxf(v[3], -m), m)};
            }
        }
    }
}

template <int CIN, int COUT>
static int launch_feat_conv(const void* x, int in_dtype, const void* wp, const float* bias, int act, void* out, int out_dtype, int N, int H, int W,
                            long long in_bs, long long out_bs, hipStream_t st) {
    constexpr int CH = CIN < 32 ? CIN : 32, OPT = CH / 8;
    const int tiles_x = (int)ceil_div(W, FC_TW), tiles_y = (int)ceil_div(H, FC_TH);
    const size_t lds = (size_t)OPT * FC_PLANE;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d3x3_tiles_kernel<CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv2d3x3_tiles_kernel<CIN, COUT>), dim3(tiles_x * tiles_y, N), dim3(256), lds, st, x, in_dtype, wp, bias, act, out, out_dtype, H, W,
                       in_bs, out_bs, tiles_x, tiles_x * tiles_y);
    return check_launch("conv2d3x3_tiles_kernel");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_feature_conv_is_built(int Cin, int Cout) {
    return (Cin == Cout && (Cin == 8 || Cin == 16 || Cin == 32)) || (Cin == 64 && (Cout == 8 || Cout == 16 || Cout == 32)) ? 1 : 0;
}

extern "C" int mvs_conv2d3x3_tiles_fwd(const void* x, int in_dtype, const void* w_packed, const float* bias, int act, void* tiled, int out_dtype, int N,
                                       int Cin, int Cout, int H, int W, long long in_batch_stride, long long out_batch_stride, void* stream) {
    if (!x || !w_packed || !tiled || N < 1 || H < 1 || W < 1) { set_error("mvs_conv2d3x3_tiles_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (in_dtype < MVS_DTYPE_F32 || in_dtype > MVS_DTYPE_F16 || out_dtype < MVS_DTYPE_F32 || out_dtype > MVS_DTYPE_F16) { set_error("mvs_conv2d3x3_tiles_fwd: unknown dtype"); return MVS_ERR_ARG; }
    if (act != 0 && act != 1) { set_error("mvs_conv2d3x3_tiles_fwd: activation must be 0 (none) or 1 (Swish)"); return MVS_ERR_ARG; }
    if (in_batch_stride < (long long)Cin * H * W || out_batch_stride < (long long)Cout * H * W) { set_error("mvs_conv2d3x3_tiles_fwd: batch strides shorter than one image"); return MVS_ERR_ARG; }
    if (!mvs_feature_conv_is_built(Cin, Cout)) {
        set_error("mvs_conv2d3x3_tiles_fwd: built for the reference's feature heads - (Cin, Cout) = (8,8), (16,16), (32,32) [FMT.py:195-197] and "
                  "(64,8), (64,16), (64,32) [module.py:257-270]; got (%d, %d)", Cin, Cout);
        return MVS_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
#define MVS_FC(CI, CO) if (Cin == CI && Cout == CO) return launch_feat_conv<CI, CO>(x, in_dtype, w_packed, bias, act, tiled, out_dtype, N, H, W, in_batch_stride, out_batch_stride, st);
    MVS_FC(8, 8) MVS_FC(16, 16) MVS_FC(32, 32) MVS_FC(64, 8) MVS_FC(64, 16) MVS_FC(64, 32)
#undef MVS_FC
    return MVS_ERR_UNSUPPORTED;
}
