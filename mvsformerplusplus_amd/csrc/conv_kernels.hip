// 3D-conv regulariser kernels: implicit-GEMM Conv3d / ConvTranspose3d
// with folded BatchNorm + ReLU (+ skip add) on v_mfma_f32_16x16x4_f32, i.e. exact fp32 contraction - bf16
// inputs can break the 1e-3 depth bar when the logits are peaky.
//
// GEMM view of one workgroup (256 threads = 4 waves):
//     D[cout, voxel] += W[cout, k] * X[k, voxel],    k = (tap, cin)
//   * A operand = packed weights (rows = 16 output channels), read straight from global/L2 in the
//     exact per-lane order the MFMA wants (1 KiB contiguous per wave-instruction, see packing.py)
//   * B operand = activations of 16 consecutive output x-positions ("n-block"), read from an LDS copy of the
//     input tile + halo, channel-last with the voxel stride padded to CH+4 floats so that the 16 lanes of
//     a ds_read_b128 group fall into different 16-byte bank slots
//   * one ds_read_b128 / global_load_dwordx4 feeds FOUR MFMAs: lane group g = lane>>4 supplies k-quad
//     kq = 4*step + g and component s of its float4 is used by MFMA s (the MFMA's k index is only a label:
//     any bijection works as long as A and B agree)
//   * D fragment: lane (voxel = lane&15, g) holds output channels 16*mb + 4*g + 0..3 -> one float4 store into
//     the channel-last output, fused with bias (folded BN), ReLU and the U-Net skip add
// Activations are channel-last fp32 [B, D, H, W, C].
#include "conv_cfg.h"

#include <stdlib.h>

namespace mvs {

// operands of contraction step t: MREP weight fragments (global, 1 KiB contiguous per wave) + NREP activation
// fragments (LDS).  Lane group g owns k-quad 4t+g: (tap, cq) = divmod(4t+g, QC); padded quads carry zero weights.
template <class Cfg>
__device__ __forceinline__ void conv_load_step(int t, int g, const float4* wq, const float* lds, const int* voxbase,
                                               float4* a, float4* b) {
    constexpr int QC = Cfg::QC;
    int tap, cq;
    if (QC >= 4) { tap = (4 * t) / QC; cq = (4 * t) % QC + g; }       // tap is wave-uniform
    else { const int kq = 4 * t + g; tap = kq / QC; cq = kq % QC; }
    tap = tap < Cfg::NTAP ? tap : Cfg::NTAP - 1;
    const int kd = tap / 9, r9 = tap - kd * 9, kh = r9 / 3, kw = r9 - kh * 3;
    const int tapoff = ((kd * Cfg::IH + kh) * Cfg::IW + kw) * Cfg::S + cq * 4;
#pragma unroll
    for (int mb = 0; mb < Cfg::MREP; ++mb) a[mb] = wq[(size_t)(t * Cfg::MREP + mb) * 64];
#pragma unroll
    for (int nb = 0; nb < Cfg::NREP; ++nb) b[nb] = *reinterpret_cast<const float4*>(lds + voxbase[nb] + tapoff);
}

template <int MREP, int NREP>
__device__ __forceinline__ void conv_mfma_step(const float4* a, const float4* b, f32x4 (*acc)[NREP]) {
    // component-outer order: consecutive MFMAs target different accumulators (v_mfma_f32_16x16x4_f32 issues every
    // 32 cycles but a dependent accumulator is ready only after 40)
#define MVS_MFMA_ROUND(COMP)                                                                                          \
    _Pragma("unroll") for (int mb = 0; mb < MREP; ++mb) _Pragma("unroll") for (int nb = 0; nb < NREP; ++nb)          \
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].COMP, b[nb].COMP, acc[mb][nb], 0, 0, 0);
    MVS_MFMA_ROUND(x)
    MVS_MFMA_ROUND(y)
    MVS_MFMA_ROUND(z)
    MVS_MFMA_ROUND(w)
#undef MVS_MFMA_ROUND
}

template <class Cfg>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(const float* __restrict__ x, const float* wp,
                                                          const float* __restrict__ bias, float* __restrict__ y, int D, int H, int W,
                                                          int OD, int OH, int OW, int relu, int tiles_x, int tiles_y, int ntiles) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, SD = Cfg::SD, SH = Cfg::SH, SW = Cfg::SW, TD = Cfg::TD, TH = Cfg::TH, CH = Cfg::CH;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, S = Cfg::S, QC = Cfg::QC, NSTEP = Cfg::NSTEP, MREP = Cfg::MREP, NREP = Cfg::NREP;
    HIP_DYNAMIC_SHARED(float4, lds4)                          // float4 => the LDS base is known 16-byte aligned (ds_read_b128)
    float* lds = reinterpret_cast<float*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
    const int b = (int)blockIdx.y;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, tz = tile / tiles_y;
    const int oz0 = tz * TD, oy0 = ty * TH, ox0 = tx * 16;
    const int iz0 = oz0 * SD - Cfg::PD, iy0 = oy0 * SH - 1, ix0 = ox0 * SW - 1;

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    int voxbase[NREP];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int nbg = wave * NREP + nb;
        const int oz = nbg / TH, oy = nbg % TH;
        voxbase[nb] = (((oz * SD) * IH + oy * SH) * IW + li * SW) * S;
    }

    const float* xb = x + (size_t)b * D * H * W * CIN;
    for (int pass = 0; pass < Cfg::NPASS; ++pass) {
        if (pass > 0) __syncthreads();                       // everyone is done reading the previous chunk
        // ---- stage the input tile (+halo) of channel chunk `pass` into LDS, zero-filled outside the volume ----
        for (int e = tid; e < Cfg::NVOX * QC; e += 256) {
            const int vox = e / QC, cq = e - vox * QC;
            const int dx = vox % IW;
            const int t2 = vox / IW;
            const int dy = t2 % IH, dz = t2 / IH;
            const int z = iz0 + dz, yy = iy0 + dy, xx = ix0 + dx;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
                v = *reinterpret_cast<const float4*>(xb + (((size_t)z * H + yy) * W + xx) * CIN + pass * CH + cq * 4);
            *reinterpret_cast<float4*>(lds + vox * S + cq * 4) = v;
        }
        __syncthreads();
        // ---- contraction over (tap, cin) of this chunk ----
        // Software pipeline with two named register sets (no copies): the weight (global/L2) and activation (LDS)
        // operands of step t+1 are requested before the MFMAs of step t are issued, so their latency hides under
        // 4*MREP*NREP MFMAs (>= 512 cycles) instead of being exposed in front of every step.
        const float4* wq = reinterpret_cast<const float4*>(wp) + (size_t)pass * NSTEP * MREP * 64 + lane;
        float4 a0[MREP], b0[NREP], a1[MREP], b1[NREP];
        conv_load_step<Cfg>(0, g, wq, lds, voxbase, a0, b0);
#pragma unroll 1
        for (int t = 0; t + 1 < NSTEP; t += 2) {
            conv_load_step<Cfg>(t + 1, g, wq, lds, voxbase, a1, b1);
            __builtin_amdgcn_sched_barrier(0);                 // keep the prefetch ABOVE the MFMAs it hides under
            conv_mfma_step<MREP, NREP>(a0, b0, acc);
            conv_load_step<Cfg>(t + 2 < NSTEP ? t + 2 : NSTEP - 1, g, wq, lds, voxbase, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            conv_mfma_step<MREP, NREP>(a1, b1, acc);
        }
        if (NSTEP & 1) conv_mfma_step<MREP, NREP>(a0, b0, acc);    // odd step count: the last prefetched step
    }

    // ---- epilogue: + bias (folded BN), ReLU, channel-last float4 store ----
    float* yb = y + (size_t)b * OD * OH * OW * COUT;
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int nbg = wave * NREP + nb;
        const int oz = oz0 + nbg / TH, oy = oy0 + nbg % TH, ox = ox0 + li;
        if (oz >= OD || oy >= OH || ox >= OW) continue;
        float* o = yb + (((size_t)oz * OH + oy) * OW + ox) * COUT;
#pragma unroll
        for (int mb = 0; mb < MREP; ++mb) {
            const int co = 16 * mb + 4 * g;
            if (co >= COUT) continue;
            const float4 bb = *reinterpret_cast<const float4*>(bias + co);
            float4 v = make_float4(acc[mb][nb][0] + bb.x, acc[mb][nb][1] + bb.y, acc[mb][nb][2] + bb.z, acc[mb][nb][3] + bb.w);
            if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            *reinterpret_cast<float4*>(o + co) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose3d k=3, padding 1, stride (SD,2,2), output_padding (SD-1,1,1) + BN + ReLU (+ skip)
//                                                          module.py:129-165, 381-383, 466-481
// Output voxel o = stride*m + parity.  Per axis with stride 2: parity 0 <- (k=1, input m); parity 1 <-
// (k=0, input m+1) and (k=2, input m).  Depth axis with stride 1: o = m <- (k=0, m+1), (k=1, m), (k=2, m-1).
// The input tile (all CIN channels, +1 halo) is staged once and reused by the 4 / 8 parity classes.
// ------------------------------------------------------------------------------------------------
// operands of step `st` of parity class (pd,ph,pw): decode (tap, q), then MREP weight + NREP activation fragments
template <class Cfg>
__device__ __forceinline__ void deconv_load_step(int st, int pd, int ph, int pw, const float4* wq, const float* lds, const int* voxbase,
                                                 float4* a, float4* b) {
    constexpr int SD = Cfg::SD, NQ = Cfg::NQ;
    const int q = st % NQ;
    int ti = st / NQ;
    const int nkw = pw ? 2 : 1, nkh = ph ? 2 : 1;
    const int a_w = ti % nkw;
    ti /= nkw;
    const int a_h = ti % nkh, a_d = ti / nkh;
    // (kernel index, input offset) per axis: stride 2 parity 0 <- (1, 0); parity 1 <- (0, +1), (2, 0); stride 1 <- (k, 1 - k)
    const int kd = (SD == 2) ? (pd ? 2 * a_d : 1) : a_d;
    const int od = (SD == 2) ? (pd ? 1 - a_d : 0) : 1 - a_d;
    const int kh = ph ? 2 * a_h : 1, oh = ph ? 1 - a_h : 0;
    const int kw = pw ? 2 * a_w : 1, ow = pw ? 1 - a_w : 0;
    const int tap = (kd * 3 + kh) * 3 + kw;
    const int ldsoff = ((od * Cfg::LH + oh) * Cfg::LW + ow) * Cfg::S + q * 16;
#pragma unroll
    for (int mb = 0; mb < Cfg::MREP; ++mb) a[mb] = wq[(size_t)((tap * NQ + q) * Cfg::MREP + mb) * 64];
#pragma unroll
    for (int nb = 0; nb < Cfg::NREP; ++nb) b[nb] = *reinterpret_cast<const float4*>(lds + voxbase[nb] + ldsoff);
}

template <class Cfg>
__global__ __launch_bounds__(256) void deconv3d_mfma_kernel(const float* __restrict__ x, const float* wp,
                                                            const float* __restrict__ bias, const float* __restrict__ skip,
                                                            float* __restrict__ y, int D, int H, int W, int tiles_x, int tiles_y,
                                                            int ntiles) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, SD = Cfg::SD, TDM = Cfg::TDM, THM = Cfg::THM;
    constexpr int LH = Cfg::LH, LW = Cfg::LW, S = Cfg::S, QC = Cfg::QC, NQ = Cfg::NQ, MREP = Cfg::MREP, NREP = Cfg::NREP;
    HIP_DYNAMIC_SHARED(float4, lds4)
    float* lds = reinterpret_cast<float*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
    const int b = (int)blockIdx.y;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, tz = tile / tiles_y;
    const int mz0 = tz * TDM, my0 = ty * THM, mx0 = tx * 16;
    const int OD = D * SD, OH = 2 * H, OW = 2 * W;

    // ---- stage input tile: z in [mz0 - ZO, ...), y in [my0, my0 + THM], x in [mx0, mx0 + 16] ----
    const float* xb = x + (size_t)b * D * H * W * CIN;
    for (int e = tid; e < Cfg::NVOX * QC; e += 256) {
        const int vox = e / QC, cq = e - vox * QC;
        const int dx = vox % LW;
        const int t2 = vox / LW;
        const int dy = t2 % LH, dz = t2 / LH;
        const int z = mz0 - Cfg::ZO + dz, yy = my0 + dy, xx = mx0 + dx;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (z >= 0 && z < D && yy < H && xx < W)
            v = *reinterpret_cast<const float4*>(xb + (((size_t)z * H + yy) * W + xx) * CIN + cq * 4);
        *reinterpret_cast<float4*>(lds + vox * S + cq * 4) = v;
    }
    __syncthreads();

    int voxbase[NREP];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int nbg = wave * NREP + nb;
        const int mz = nbg / THM, my = nbg % THM;
        voxbase[nb] = (((mz + Cfg::ZO) * LH + my) * LW + li) * S + g * 4;
    }
    float* yb = y + (size_t)b * OD * OH * OW * COUT;
    const float* sb = skip ? skip + (size_t)b * OD * OH * OW * COUT : nullptr;

    constexpr int NCLS = (SD == 2 ? 2 : 1) * 4;
    for (int cls = 0; cls < NCLS; ++cls) {
        const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (SD == 2) ? (cls >> 2) : 0;
        f32x4 acc[MREP][NREP];
#pragma unroll
        for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
            for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

        // flat, software-pipelined step list of this class: step = (tap of the class, 16-channel block q)
        const int nkd = (SD == 2) ? (pd ? 2 : 1) : 3;
        const int nkh = ph ? 2 : 1, nkw = pw ? 2 : 1;
        const int nst = nkd * nkh * nkw * NQ;
        const float4* wq = reinterpret_cast<const float4*>(wp) + lane;
        float4 a0[MREP], b0[NREP], a1[MREP], b1[NREP];
        deconv_load_step<Cfg>(0, pd, ph, pw, wq, lds, voxbase, a0, b0);
#pragma unroll 1
        for (int st = 0; st + 1 < nst; st += 2) {
            deconv_load_step<Cfg>(st + 1, pd, ph, pw, wq, lds, voxbase, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            conv_mfma_step<MREP, NREP>(a0, b0, acc);
            deconv_load_step<Cfg>(st + 2 < nst ? st + 2 : nst - 1, pd, ph, pw, wq, lds, voxbase, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            conv_mfma_step<MREP, NREP>(a1, b1, acc);
        }
        if (nst & 1) conv_mfma_step<MREP, NREP>(a0, b0, acc);
        // ---- epilogue of this parity class: relu(acc + bias) + skip ----
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) {
            const int nbg = wave * NREP + nb;
            const int mz = mz0 + nbg / THM, my = my0 + nbg % THM, mx = mx0 + li;
            if (mz >= D || my >= H || mx >= W) continue;
            const int oz = mz * SD + pd, oy = 2 * my + ph, ox = 2 * mx + pw;
            const size_t off = (((size_t)oz * OH + oy) * OW + ox) * COUT;
#pragma unroll
            for (int mb = 0; mb < MREP; ++mb) {
                const int co = 16 * mb + 4 * g;
                if (co >= COUT) continue;
                const float4 bb = *reinterpret_cast<const float4*>(bias + co);
                float4 v = make_float4(fmaxf(acc[mb][nb][0] + bb.x, 0.0f), fmaxf(acc[mb][nb][1] + bb.y, 0.0f),
                                       fmaxf(acc[mb][nb][2] + bb.z, 0.0f), fmaxf(acc[mb][nb][3] + bb.w, 0.0f));
                if (sb) {
                    const float4 sk = *reinterpret_cast<const float4*>(sb + off + co);
                    v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;     // skip is added AFTER the ReLU (module.py:402-405)
                }
                *reinterpret_cast<float4*>(yb + off + co) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// visibility CNN ends (cost_volume.py:36): 1 -> 16 3x3 conv + BN + ReLU, and 8 -> 1 1x1 conv + sigmoid
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vis_conv1_kernel(const float* __restrict__ ent, const float* __restrict__ w1 /*[9][16]*/,
                                                        const float* __restrict__ b1 /*[16]*/, float* __restrict__ out /*[N,H,W,16]*/,
                                                        int H, int W) {
    const int HW = H * W;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int n = (int)blockIdx.y;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    const float* e = ent + (size_t)n * HW;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int yy = y + kh - 1, xx = x + kw - 1;
            const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? e[yy * W + xx] : 0.0f;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += v * w1[(kh * 3 + kw) * 16 + c];
        }
    float4* o = reinterpret_cast<float4*>(out + ((size_t)n * HW + p) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = make_float4(fmaxf(acc[4 * q] + b1[4 * q], 0.0f), fmaxf(acc[4 * q + 1] + b1[4 * q + 1], 0.0f),
                           fmaxf(acc[4 * q + 2] + b1[4 * q + 2], 0.0f), fmaxf(acc[4 * q + 3] + b1[4 * q + 3], 0.0f));
}

__global__ __launch_bounds__(256) void vis_out_kernel(const float* __restrict__ x /*[N,H,W,8]*/, const float* __restrict__ w4,
                                                      const float* __restrict__ b4, float* __restrict__ vis, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float4* f = reinterpret_cast<const float4*>(x + i * 8);
    const float4 a = f[0], c = f[1];
    float v = a.x * w4[0];
    v += a.y * w4[1]; v += a.z * w4[2]; v += a.w * w4[3];
    v += c.x * w4[4]; v += c.y * w4[5]; v += c.z * w4[6]; v += c.w * w4[7];
    v += b4[0];
    vis[i] = 1.0f / (1.0f + expf(-v));
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <class Cfg>
static int launch_conv(const float* x, const float* wp, const float* bias, float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
    const int OD = (D + 2 * Cfg::PD - Cfg::KD) / Cfg::SD + 1, OH = (H - 1) / Cfg::SH + 1, OW = (W - 1) / Cfg::SW + 1;
    const int tx = (int)ceil_div(OW, 16), ty = (int)ceil_div(OH, Cfg::TH), tz = (int)ceil_div(OD, Cfg::TD);
    const int ntiles = tx * ty * tz;
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_mfma_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL((conv3d_mfma_kernel<Cfg>), dim3(ntiles, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, y, D, H, W, OD, OH, OW, relu, tx, ty, ntiles);
    return check_launch("conv3d_mfma_kernel");
}

template <class Cfg>
static int launch_deconv(const float* x, const float* wp, const float* bias, const float* skip, float* y, int B, int D, int H, int W, hipStream_t st) {
    const int tx = (int)ceil_div(W, 16), ty = (int)ceil_div(H, Cfg::THM), tz = (int)ceil_div(D, Cfg::TDM);
    const int ntiles = tx * ty * tz;
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3d_mfma_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL((deconv3d_mfma_kernel<Cfg>), dim3(ntiles, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, skip, y, D, H, W, tx, ty, ntiles);
    return check_launch("deconv3d_mfma_kernel");
}

// activation / weight form of one MFMA layer under `prec`: -1 = not an MFMA precision, 0 = fp32 activations (split on the fly), 1 = split
// bf16 pairs in HBM, 2 = fp16 activations + fp16 hi / lo weights, 3 = fp16 activations + ONE fp16 weight term.  MVS_PREC_F16MIX drops w_lo
// on the layers with 32 / 64 channels on both sides or 64 on one (conv4-conv7: where the second term costs most and changes nothing
// measurable, scripts/study_weight_precision.py), MVS_PREC_F16 on every layer.
static int mfma_form(int prec, int Cin, int Cout) {
    switch (prec) {
        case MVS_PREC_BF16X3: return 0;
        case MVS_PREC_BF16X3_SPLIT: return 1;
        case MVS_PREC_F16X2: return 2;
        case MVS_PREC_F16: return 3;
        case MVS_PREC_F16MIX: return ((Cin < Cout ? Cin : Cout) >= 32 || (Cin > Cout ? Cin : Cout) >= 64) ? 3 : 2;
        default: return -1;
    }
}

int conv3d_dispatch(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W, int kd,
                    int sd, int sh, int sw, int relu, int prec, hipStream_t st) {
    if (mfma_form(prec, Cin, Cout) >= 0)
        return conv3d_dispatch_bf16x3(x, wp, bias, y, B, Cin, Cout, D, H, W, kd, sd, sh, sw, relu, st, nullptr, mfma_form(prec, Cin, Cout));
    if (prec != MVS_PREC_FP32) { set_error("conv3d: unknown precision %d", prec); return MVS_ERR_ARG; }
#define MVS_X(CI, CO, KD, SD, SH, SW, TD, TH, CH)                                                     \
    if (Cin == CI && Cout == CO && kd == KD && sd == SD && sh == SH && sw == SW)                      \
        return launch_conv<ConvCfg<CI, CO, KD, SD, SH, SW, TD, TH, CH>>(x, static_cast<const float*>(wp), bias, y, B, D, H, W, relu, st);
    MVS_CONV_TABLE(MVS_X)
#undef MVS_X
    set_error("conv3d: no kernel for Cin=%d Cout=%d kernel=(%d,3,3) stride=(%d,%d,%d)", Cin, Cout, kd, sd, sh, sw);
    return MVS_ERR_UNSUPPORTED;
}

int deconv3d_dispatch(const float* x, const void* wp, const float* bias, const float* skip, float* y, int B, int Cin, int Cout, int D,
                      int H, int W, int sd, int prec, hipStream_t st, const float* prob_w = nullptr, const float* prob_b = nullptr,
                      float* logits = nullptr) {
    if (mfma_form(prec, Cin, Cout) >= 0)
        return deconv3d_dispatch_bf16x3(x, wp, bias, skip, y, B, Cin, Cout, D, H, W, sd, st, prob_w, prob_b, logits, 1, mfma_form(prec, Cin, Cout));
    if (prob_w != nullptr) { set_error("deconv3d: the fused prob head exists for the bf16x3 contraction only"); return MVS_ERR_UNSUPPORTED; }
    if (prec != MVS_PREC_FP32) { set_error("deconv3d: unknown precision %d", prec); return MVS_ERR_ARG; }
#define MVS_X(CI, CO, SD, TDM, THM)                                                                     \
    if (Cin == CI && Cout == CO && sd == SD)                                                            \
        return launch_deconv<DeconvCfg<CI, CO, SD, TDM, THM>>(x, static_cast<const float*>(wp), bias, skip, y, B, D, H, W, st);
    MVS_DECONV_TABLE(MVS_X)
#undef MVS_X
    set_error("deconv3d: no kernel for Cin=%d Cout=%d stride=(%d,2,2)", Cin, Cout, sd);
    return MVS_ERR_UNSUPPORTED;
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_conv3d_bn_relu_fwd(const float* x_cl, const void* w_packed, const float* bias, float* y_cl, int B, int Cin, int Cout,
                                      int D, int H, int W, int kd, int sd, int sh, int sw, int relu, int precision, void* stream) {
    if (!x_cl || !w_packed || !bias || !y_cl || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_conv3d_bn_relu_fwd: bad arguments"); return MVS_ERR_ARG; }
    return conv3d_dispatch(x_cl, w_packed, bias, y_cl, B, Cin, Cout, D, H, W, kd, sd, sh, sw, relu, precision, (hipStream_t)stream);
}

extern "C" int mvs_deconv3d_linear_fwd(const float* x_cl, const void* w_packed, const float* bias, float* y_cl, int B, int Cin, int Cout, int D,
                                       int H, int W, int sd, int precision, void* stream) {
    if (!x_cl || !w_packed || !bias || !y_cl || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_deconv3d_linear_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (precision != MVS_PREC_BF16X3) { set_error("mvs_deconv3d_linear_fwd: only MVS_PREC_BF16X3"); return MVS_ERR_UNSUPPORTED; }
    return deconv3d_dispatch_bf16x3(x_cl, w_packed, bias, nullptr, y_cl, B, Cin, Cout, D, H, W, sd, (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}

extern "C" int mvs_conv3d_logits_fwd(const float* x_cl, const void* w_packed, const float* bias, float* logits, int B, int D, int H, int W,
                                     int precision, void* stream) {
    if (!x_cl || !w_packed || !bias || !logits || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_conv3d_logits_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (precision == MVS_PREC_F16 || precision == MVS_PREC_F16MIX) precision = MVS_PREC_F16X2;      // the one-row head keeps both weight terms
    if (precision != MVS_PREC_BF16X3 && precision != MVS_PREC_BF16X3_SPLIT && precision != MVS_PREC_F16X2) { set_error("mvs_conv3d_logits_fwd: only MVS_PREC_BF16X3 / _SPLIT / MVS_PREC_F16X2 / _F16 / _F16MIX (use mvs_prob_regress_fwd for the exact-fp32 head)"); return MVS_ERR_UNSUPPORTED; }
    return conv3d_dispatch_bf16x3(x_cl, w_packed, bias, nullptr, B, 8, 16, D, H, W, 3, 1, 1, 1, 0, (hipStream_t)stream, logits,
                                  precision == MVS_PREC_F16X2 ? 2 : (precision == MVS_PREC_BF16X3_SPLIT ? 1 : 0));
}

extern "C" int mvs_deconv3d_bn_relu_add_fwd(const float* x_cl, const void* w_packed, const float* bias, const float* skip_cl, float* y_cl,
                                            int B, int Cin, int Cout, int D, int H, int W, int sd, int precision, void* stream) {
    if (!x_cl || !w_packed || !bias || !y_cl || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_deconv3d_bn_relu_add_fwd: bad arguments"); return MVS_ERR_ARG; }
    return deconv3d_dispatch(x_cl, w_packed, bias, skip_cl, y_cl, B, Cin, Cout, D, H, W, sd, precision, (hipStream_t)stream);
}

// the two ends of the visibility CNN as separate entry points (mvs_vis_weight_fwd chains them with the MFMA convs)
extern "C" int mvs_vis_conv1_fwd(const float* entropy, const float* w1, const float* b1, float* out_cl16, int N, int H, int W, void* stream) {
    if (!entropy || !w1 || !b1 || !out_cl16 || N < 1 || H < 1 || W < 1) { set_error("mvs_vis_conv1_fwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(vis_conv1_kernel, dim3(ceil_div((long long)H * W, 256), N), dim3(256), 0, (hipStream_t)stream, entropy, w1, b1, out_cl16, H, W);
    return check_launch("vis_conv1_kernel");
}

extern "C" int mvs_vis_out_fwd(const float* x_cl8, const float* w4, const float* b4, float* vis, int N, int H, int W, void* stream) {
    if (!x_cl8 || !w4 || !b4 || !vis || N < 1 || H < 1 || W < 1) { set_error("mvs_vis_out_fwd: bad arguments"); return MVS_ERR_ARG; }
    const size_t total = (size_t)N * H * W;
    hipLaunchKernelGGL(vis_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_cl8, w4, b4, vis, total);
    return check_launch("vis_out_kernel");
}

extern "C" size_t mvs_vis_workspace_bytes(int N, int H, int W, int precision) {
    if (precision == MVS_PREC_BF16X3 || precision == MVS_PREC_F16X2 || precision == MVS_PREC_F16 || precision == MVS_PREC_F16MIX) return 16;   // the row-streaming kernel keeps every intermediate in LDS
    return (size_t)N * H * W * 32 * sizeof(float);
}

extern "C" int mvs_vis_weight_fwd(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                                  const float* b3, const float* w4, const float* b4, float* vis, void* workspace, size_t workspace_bytes,
                                  int N, int H, int W, int precision, void* stream) {
    if (!entropy || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !vis || !workspace || N < 1 || H < 1 || W < 1) { set_error("mvs_vis_weight_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (workspace_bytes < mvs_vis_workspace_bytes(N, H, W, precision)) { set_error("mvs_vis_weight_fwd: workspace too small (%zu < %zu)", workspace_bytes, mvs_vis_workspace_bytes(N, H, W, precision)); return MVS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const size_t HW = (size_t)H * W;
    if (precision == MVS_PREC_BF16X3 || precision == MVS_PREC_F16X2 || precision == MVS_PREC_F16 || precision == MVS_PREC_F16MIX)
        // one row-streaming launch, every intermediate in LDS (vis_kernels.hip); F16 / F16MIX: one weight term
        return vis_weight_stream_bf16x3(entropy, w1, b1, w2, b2, w3, b3, w4, b4, vis, N, H, W, st,
                                        precision == MVS_PREC_F16X2 ? 1 : (precision == MVS_PREC_BF16X3 ? 0 : 2));
    float* t1 = static_cast<float*>(workspace);          // [N,H,W,16]
    float* t2 = t1 + (size_t)N * HW * 16;                // [N,H,W,16]
    hipLaunchKernelGGL(vis_conv1_kernel, dim3(ceil_div((long long)HW, 256), N), dim3(256), 0, st, entropy, w1, b1, t1, H, W);
    int rc = check_launch("vis_conv1_kernel");
    if (rc != MVS_OK) return rc;
    rc = conv3d_dispatch(t1, w2, b2, t2, 1, 16, 16, N, H, W, 1, 1, 1, 1, 1, precision, st);      // views ride on the depth axis, kd = 1
    if (rc != MVS_OK) return rc;
    rc = conv3d_dispatch(t2, w3, b3, t1, 1, 16, 8, N, H, W, 1, 1, 1, 1, 1, precision, st);       // -> [N,H,W,8] in t1
    if (rc != MVS_OK) return rc;
    const size_t total = (size_t)N * HW;
    hipLaunchKernelGGL(vis_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, t1, w4, b4, vis, total);
    return check_launch("vis_out_kernel");
}

// level sizes of the U-Net: CostRegNet halves D,H,W per level, CostRegNet3D halves H,W only
static void regnet_level(int kind, int lvl, int D, int H, int W, int* d, int* h, int* w) {
    *d = (kind == MVS_REG_COSTREGNET) ? (D >> lvl) : D;
    *h = H >> lvl;
    *w = W >> lvl;
}

extern "C" size_t mvs_regnet_workspace_bytes(int kind, int B, int D, int H, int W) {
    size_t total = 0;
    for (int lvl = 1; lvl <= 3; ++lvl) {
        int d, h, w;
        regnet_level(kind, lvl, D, H, W, &d, &h, &w);
        total += (size_t)2 * B * d * h * w * (8u << lvl);
    }
    return total * sizeof(float);
}

extern "C" int mvs_deconv3d_prob_fwd(const float* x_cl, const void* w_packed, const float* bias, const float* skip_cl, const float* prob_w,
                                     const float* prob_b, float* logits, int B, int Cin, int D, int H, int W, int sd, int precision,
                                     void* stream) {
    if (!x_cl || !w_packed || !bias || !prob_w || !prob_b || !logits || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_deconv3d_prob_fwd: bad arguments"); return MVS_ERR_ARG; }
    return deconv3d_dispatch(x_cl, w_packed, bias, skip_cl, nullptr, B, Cin, 8, D, H, W, sd, precision, (hipStream_t)stream, prob_w, prob_b, logits);
}

static int regnet_run(int kind, const float* volume_cl, const void* const* w_packed, const float* const* bias, float* feat_cl,
                      const float* prob_w, const float* prob_b, float* logits, void* workspace, size_t workspace_bytes, int B, int D, int H,
                      int W, int precision, void* stream, const char* who) {
    if (!volume_cl || !w_packed || !bias || !workspace || B < 1) { set_error("%s: bad arguments", who); return MVS_ERR_ARG; }
    if (kind != MVS_REG_COSTREGNET && kind != MVS_REG_COSTREGNET3D) { set_error("%s: unknown regulariser kind %d", who, kind); return MVS_ERR_ARG; }
    if ((H % 8) || (W % 8) || (kind == MVS_REG_COSTREGNET && (D % 8))) {
        set_error("%s: spatial size %dx%dx%d must be divisible by 8 (U-Net skip adds, module.py:403-405)", who, D, H, W);
        return MVS_ERR_ARG;
    }
    if (workspace_bytes < mvs_regnet_workspace_bytes(kind, B, D, H, W)) { set_error("%s: workspace too small", who); return MVS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int sd = (kind == MVS_REG_COSTREGNET) ? 2 : 1;
    int d1, h1, w1, d2, h2, w2, d3, h3, w3;
    regnet_level(kind, 1, D, H, W, &d1, &h1, &w1);
    regnet_level(kind, 2, D, H, W, &d2, &h2, &w2);
    regnet_level(kind, 3, D, H, W, &d3, &h3, &w3);
    float* ws = static_cast<float*>(workspace);
    const size_t n1 = (size_t)B * d1 * h1 * w1 * 16, n2 = (size_t)B * d2 * h2 * w2 * 32, n3 = (size_t)B * d3 * h3 * w3 * 64;
    float *c1 = ws, *c2 = c1 + n1, *c3 = c2 + n1, *c4 = c3 + n2, *c5 = c4 + n2, *c6 = c5 + n3;
    int rc;
#define MVS_TRY(expr) do { rc = (expr); if (rc != MVS_OK) return rc; } while (0)
    MVS_TRY(conv3d_dispatch(volume_cl, w_packed[0], bias[0], c1, B, 8, 16, D, H, W, 3, sd, 2, 2, 1, precision, st));       // conv1
    MVS_TRY(conv3d_dispatch(c1, w_packed[1], bias[1], c2, B, 16, 16, d1, h1, w1, 3, 1, 1, 1, 1, precision, st));           // conv2
    MVS_TRY(conv3d_dispatch(c2, w_packed[2], bias[2], c3, B, 16, 32, d1, h1, w1, 3, sd, 2, 2, 1, precision, st));          // conv3
    MVS_TRY(conv3d_dispatch(c3, w_packed[3], bias[3], c4, B, 32, 32, d2, h2, w2, 3, 1, 1, 1, 1, precision, st));           // conv4
    MVS_TRY(conv3d_dispatch(c4, w_packed[4], bias[4], c5, B, 32, 64, d2, h2, w2, 3, sd, 2, 2, 1, precision, st));          // conv5
    MVS_TRY(conv3d_dispatch(c5, w_packed[5], bias[5], c6, B, 64, 64, d3, h3, w3, 3, 1, 1, 1, 1, precision, st));           // conv6
    MVS_TRY(deconv3d_dispatch(c6, w_packed[6], bias[6], c4, c3, B, 64, 32, d3, h3, w3, sd, precision, st));                // conv4 + conv7 -> c3
    MVS_TRY(deconv3d_dispatch(c3, w_packed[7], bias[7], c2, c1, B, 32, 16, d2, h2, w2, sd, precision, st));                // conv2 + conv9 -> c1
    MVS_TRY(deconv3d_dispatch(c1, w_packed[8], bias[8], volume_cl, feat_cl, B, 16, 8, d1, h1, w1, sd, precision, st, prob_w, prob_b,
                              logits));                                                                                    // conv0 + conv11 (+ prob)
#undef MVS_TRY
    return MVS_OK;
}

extern "C" int mvs_regnet_fwd(int kind, const float* volume_cl, const void* const* w_packed, const float* const* bias, float* feat_cl,
                              void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int precision, void* stream) {
    if (!feat_cl) { set_error("mvs_regnet_fwd: bad arguments"); return MVS_ERR_ARG; }
    return regnet_run(kind, volume_cl, w_packed, bias, feat_cl, nullptr, nullptr, nullptr, workspace, workspace_bytes, B, D, H, W, precision, stream,
                      "mvs_regnet_fwd");
}

extern "C" int mvs_regnet_logits_fwd(int kind, const float* volume_cl, const void* const* w_packed, const float* const* bias,
                                     const float* prob_w, const float* prob_b, float* logits, void* workspace, size_t workspace_bytes, int B,
                                     int D, int H, int W, int precision, void* stream) {
    if (!prob_w || !prob_b || !logits) { set_error("mvs_regnet_logits_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (mfma_form(precision, 16, 8) < 0) { set_error("mvs_regnet_logits_fwd: the fused 1x1x1 head exists for the MFMA precisions only (MVS_PREC_BF16X3 / _SPLIT / F16X2 / F16 / F16MIX)"); return MVS_ERR_UNSUPPORTED; }
    return regnet_run(kind, volume_cl, w_packed, bias, nullptr, prob_w, prob_b, logits, workspace, workspace_bytes, B, D, H, W, precision, stream,
                      "mvs_regnet_logits_fwd");
}
