// Split-bf16 ("bf16x3") variants of the implicit-GEMM Conv3d / ConvTranspose3d kernels of conv_kernels.hip.
//
// Why: with exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 157 TFLOP/s) the 238 GFLOP of regulariser work per reference view
// are compute-bound at >= 1.5 ms, 2.4x the HBM time of the whole path.  Plain bf16 inputs (2.5 PFLOP/s) break the 1e-3
// depth bar when the logits are peaky.  The classic remedy keeps fp32 accuracy on the bf16
// pipe: split every fp32 operand x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together) and
// contract three bf16 products, a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (the dropped lo*lo term is 2^-16 relative
// smaller), accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  3 MFMAs of 16 cycles cover the k-range of 8 fp32 MFMAs of
// 32 cycles: 5.3x less matrix-pipe time at ~2^-16 relative product error (plain bf16: 2^-8).
//
// Layout differences from the fp32 kernels (tiles, halo, epilogues and tile tables are shared, conv_cfg.h):
//   * activations stay fp32 channel-last in HBM; they are split once while the tile is staged into LDS as
//     [voxel][octet of 8 channels][hi x8 | lo x8] (same bytes per voxel as fp32)
//   * weights are split on the host (packing.pack_conv_weights_bf16x3) in per-lane MFMA operand order
//   * one contraction step = 32 k-values = 4 channel octets; lane group g = lane>>4 owns octet 4*step + g
#include "conv_cfg.h"

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];                 // round to nearest even (v_cvt_pk_bf16_f32)
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// three-term split product, term-outer so that consecutive MFMAs hit different accumulators
template <int MREP, int NREP>
__device__ __forceinline__ void bf_mfma_step(const bf16x8* ah, const bf16x8* al, const bf16x8* bh, const bf16x8* bl, f32x4 (*acc)[NREP]) {
#pragma unroll
    for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
        for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Conv3d
// ------------------------------------------------------------------------------------------------
template <class Cfg>
struct BfConv {
    static constexpr int OPT = Cfg::CH / 8;                              // octets per tap and voxel in one pass
    static constexpr int NSTEP = (Cfg::NTAP * OPT + 3) / 4;
    static constexpr int SB = Cfg::S * 4;                                // bytes per LDS voxel (CH*4 + 16)
    static_assert(Cfg::CH % 8 == 0, "split-bf16 path stages whole octets");
};

template <class Cfg>
__device__ __forceinline__ void bf_conv_load_step(int t, int g, const bf16x8* wq, const char* ldsb, const int* voxbase, bf16x8* ah,
                                                  bf16x8* al, bf16x8* bh, bf16x8* bl) {
    constexpr int OPT = BfConv<Cfg>::OPT;
    const int o = 4 * t + g;
    int tap = o / OPT;
    const int oc = o - tap * OPT;
    tap = tap < Cfg::NTAP ? tap : Cfg::NTAP - 1;                          // padded octets carry zero weights
    const int kd = tap / 9, r9 = tap - kd * 9, kh = r9 / 3, kw = r9 - kh * 3;
    const int tapoff = ((kd * Cfg::IH + kh) * Cfg::IW + kw) * BfConv<Cfg>::SB + oc * 32;
#pragma unroll
    for (int mb = 0; mb < Cfg::MREP; ++mb) {
        ah[mb] = wq[(size_t)((t * Cfg::MREP + mb) * 2) * 64];
        al[mb] = wq[(size_t)((t * Cfg::MREP + mb) * 2 + 1) * 64];
    }
#pragma unroll
    for (int nb = 0; nb < Cfg::NREP; ++nb) {
        bh[nb] = *reinterpret_cast<const bf16x8*>(ldsb + voxbase[nb] + tapoff);
        bl[nb] = *reinterpret_cast<const bf16x8*>(ldsb + voxbase[nb] + tapoff + 16);
    }
}

template <class Cfg>
__global__ __launch_bounds__(256) void conv3d_mfma_bf16x3_kernel(const float* __restrict__ x, const void* wp, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int D, int H, int W, int OD, int OH, int OW,
                                                                 int relu, int tiles_x, int tiles_y, int ntiles) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, SD = Cfg::SD, SH = Cfg::SH, SW = Cfg::SW, TD = Cfg::TD, TH = Cfg::TH, CH = Cfg::CH;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, MREP = Cfg::MREP, NREP = Cfg::NREP;
    constexpr int OPT = BfConv<Cfg>::OPT, NSTEP = BfConv<Cfg>::NSTEP, SB = BfConv<Cfg>::SB;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
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
        voxbase[nb] = (((oz * SD) * IH + oy * SH) * IW + li * SW) * SB;
    }

    const float* xb = x + (size_t)b * D * H * W * CIN;
    for (int pass = 0; pass < Cfg::NPASS; ++pass) {
        if (pass > 0) __syncthreads();
        // ---- stage + split: 8 channels of one voxel per work-item iteration ----
        for (int e = tid; e < Cfg::NVOX * OPT; e += 256) {
            const int vox = e / OPT, oc = e - vox * OPT;
            const int dx = vox % IW;
            const int t2 = vox / IW;
            const int dy = t2 % IH, dz = t2 / IH;
            const int z = iz0 + dz, yy = iy0 + dy, xx = ix0 + dx;
            float4 u = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v = u;
            if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float4* src = reinterpret_cast<const float4*>(xb + (((size_t)z * H + yy) * W + xx) * CIN + pass * CH + oc * 8);
                u = src[0];
                v = src[1];
            }
            bf16x8 hi, lo;
            split8(u, v, hi, lo);
            *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32) = hi;
            *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32 + 16) = lo;
        }
        __syncthreads();
        // ---- software-pipelined contraction (see conv_kernels.hip) ----
        const bf16x8* wq = reinterpret_cast<const bf16x8*>(wp) + (size_t)pass * NSTEP * MREP * 2 * 64 + lane;
        bf16x8 ah0[MREP], al0[MREP], bh0[NREP], bl0[NREP], ah1[MREP], al1[MREP], bh1[NREP], bl1[NREP];
        bf_conv_load_step<Cfg>(0, g, wq, ldsb, voxbase, ah0, al0, bh0, bl0);
#pragma unroll 1
        for (int t = 0; t + 1 < NSTEP; t += 2) {
            bf_conv_load_step<Cfg>(t + 1, g, wq, ldsb, voxbase, ah1, al1, bh1, bl1);
            __builtin_amdgcn_sched_barrier(0);
            bf_mfma_step<MREP, NREP>(ah0, al0, bh0, bl0, acc);
            bf_conv_load_step<Cfg>(t + 2 < NSTEP ? t + 2 : NSTEP - 1, g, wq, ldsb, voxbase, ah0, al0, bh0, bl0);
            __builtin_amdgcn_sched_barrier(0);
            bf_mfma_step<MREP, NREP>(ah1, al1, bh1, bl1, acc);
        }
        if (NSTEP & 1) bf_mfma_step<MREP, NREP>(ah0, al0, bh0, bl0, acc);
    }

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
// ConvTranspose3d (parity classes as in conv_kernels.hip)
// ------------------------------------------------------------------------------------------------
template <class Cfg>
struct BfDeconv {
    static constexpr int OPT = Cfg::CIN / 8;
    static constexpr int SB = Cfg::S * 4;
};

template <class Cfg>
__device__ __forceinline__ void bf_deconv_load_step(int st, int ntap, int pd, int ph, int pw, int g, const bf16x8* wq, const char* ldsb,
                                                    const int* voxbase, bf16x8* ah, bf16x8* al, bf16x8* bh, bf16x8* bl) {
    constexpr int SD = Cfg::SD, OPT = BfDeconv<Cfg>::OPT;
    const int o = 4 * st + g;
    int ti = o / OPT;
    const int oc = o - ti * OPT;
    ti = ti < ntap ? ti : ntap - 1;                                        // padded octets carry zero weights
    const int nkw = pw ? 2 : 1, nkh = ph ? 2 : 1;
    const int a_w = ti % nkw;
    ti /= nkw;
    const int a_h = ti % nkh, a_d = ti / nkh;
    const int od = (SD == 2) ? (pd ? 1 - a_d : 0) : 1 - a_d;
    const int oh = ph ? 1 - a_h : 0, ow = pw ? 1 - a_w : 0;
    const int ldsoff = ((od * Cfg::LH + oh) * Cfg::LW + ow) * BfDeconv<Cfg>::SB + oc * 32;
#pragma unroll
    for (int mb = 0; mb < Cfg::MREP; ++mb) {
        ah[mb] = wq[(size_t)((st * Cfg::MREP + mb) * 2) * 64];
        al[mb] = wq[(size_t)((st * Cfg::MREP + mb) * 2 + 1) * 64];
    }
#pragma unroll
    for (int nb = 0; nb < Cfg::NREP; ++nb) {
        bh[nb] = *reinterpret_cast<const bf16x8*>(ldsb + voxbase[nb] + ldsoff);
        bl[nb] = *reinterpret_cast<const bf16x8*>(ldsb + voxbase[nb] + ldsoff + 16);
    }
}

template <class Cfg>
__global__ __launch_bounds__(256) void deconv3d_mfma_bf16x3_kernel(const float* __restrict__ x, const void* wp, const float* __restrict__ bias,
                                                                   const float* __restrict__ skip, float* __restrict__ y, int D, int H, int W,
                                                                   int tiles_x, int tiles_y, int ntiles) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, SD = Cfg::SD, TDM = Cfg::TDM, THM = Cfg::THM;
    constexpr int LH = Cfg::LH, LW = Cfg::LW, MREP = Cfg::MREP, NREP = Cfg::NREP;
    constexpr int OPT = BfDeconv<Cfg>::OPT, SB = BfDeconv<Cfg>::SB;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
    const int b = (int)blockIdx.y;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, tz = tile / tiles_y;
    const int mz0 = tz * TDM, my0 = ty * THM, mx0 = tx * 16;
    const int OD = D * SD, OH = 2 * H, OW = 2 * W;

    const float* xb = x + (size_t)b * D * H * W * CIN;
    for (int e = tid; e < Cfg::NVOX * OPT; e += 256) {
        const int vox = e / OPT, oc = e - vox * OPT;
        const int dx = vox % LW;
        const int t2 = vox / LW;
        const int dy = t2 % LH, dz = t2 / LH;
        const int z = mz0 - Cfg::ZO + dz, yy = my0 + dy, xx = mx0 + dx;
        float4 u = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v = u;
        if (z >= 0 && z < D && yy < H && xx < W) {
            const float4* src = reinterpret_cast<const float4*>(xb + (((size_t)z * H + yy) * W + xx) * CIN + oc * 8);
            u = src[0];
            v = src[1];
        }
        bf16x8 hi, lo;
        split8(u, v, hi, lo);
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32) = hi;
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32 + 16) = lo;
    }
    __syncthreads();

    int voxbase[NREP];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int nbg = wave * NREP + nb;
        const int mz = nbg / THM, my = nbg % THM;
        voxbase[nb] = (((mz + Cfg::ZO) * LH + my) * LW + li) * SB;
    }
    float* yb = y + (size_t)b * OD * OH * OW * COUT;
    const float* sb = skip ? skip + (size_t)b * OD * OH * OW * COUT : nullptr;
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wp) + lane;          // advanced class by class

    constexpr int NCLS = (SD == 2 ? 2 : 1) * 4;
    for (int cls = 0; cls < NCLS; ++cls) {
        const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (SD == 2) ? (cls >> 2) : 0;
        f32x4 acc[MREP][NREP];
#pragma unroll
        for (int mb = 0; mb < MREP; ++mb)
#pragma unroll
            for (int nb = 0; nb < NREP; ++nb) acc[mb][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        const int ntap = ((SD == 2) ? (pd ? 2 : 1) : 3) * (ph ? 2 : 1) * (pw ? 2 : 1);
        const int nst = (ntap * OPT + 3) / 4;
        bf16x8 ah0[MREP], al0[MREP], bh0[NREP], bl0[NREP], ah1[MREP], al1[MREP], bh1[NREP], bl1[NREP];
        bf_deconv_load_step<Cfg>(0, ntap, pd, ph, pw, g, wq, ldsb, voxbase, ah0, al0, bh0, bl0);
#pragma unroll 1
        for (int st = 0; st + 1 < nst; st += 2) {
            bf_deconv_load_step<Cfg>(st + 1, ntap, pd, ph, pw, g, wq, ldsb, voxbase, ah1, al1, bh1, bl1);
            __builtin_amdgcn_sched_barrier(0);
            bf_mfma_step<MREP, NREP>(ah0, al0, bh0, bl0, acc);
            bf_deconv_load_step<Cfg>(st + 2 < nst ? st + 2 : nst - 1, ntap, pd, ph, pw, g, wq, ldsb, voxbase, ah0, al0, bh0, bl0);
            __builtin_amdgcn_sched_barrier(0);
            bf_mfma_step<MREP, NREP>(ah1, al1, bh1, bl1, acc);
        }
        if (nst & 1) bf_mfma_step<MREP, NREP>(ah0, al0, bh0, bl0, acc);
        wq += (size_t)nst * MREP * 2 * 64;

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
                    v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
                }
                *reinterpret_cast<float4*>(yb + off + co) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <class Cfg>
static int launch_conv_bf(const float* x, const void* wp, const float* bias, float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
    const int OD = (D + 2 * Cfg::PD - Cfg::KD) / Cfg::SD + 1, OH = (H - 1) / Cfg::SH + 1, OW = (W - 1) / Cfg::SW + 1;
    const int tx = (int)ceil_div(OW, 16), ty = (int)ceil_div(OH, Cfg::TH), tz = (int)ceil_div(OD, Cfg::TD);
    const int ntiles = tx * ty * tz;
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_mfma_bf16x3_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL((conv3d_mfma_bf16x3_kernel<Cfg>), dim3(ntiles, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, y, D, H, W, OD, OH, OW, relu, tx, ty, ntiles);
    return check_launch("conv3d_mfma_bf16x3_kernel");
}

template <class Cfg>
static int launch_deconv_bf(const float* x, const void* wp, const float* bias, const float* skip, float* y, int B, int D, int H, int W, hipStream_t st) {
    const int tx = (int)ceil_div(W, 16), ty = (int)ceil_div(H, Cfg::THM), tz = (int)ceil_div(D, Cfg::TDM);
    const int ntiles = tx * ty * tz;
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3d_mfma_bf16x3_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL((deconv3d_mfma_bf16x3_kernel<Cfg>), dim3(ntiles, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, skip, y, D, H, W, tx, ty, ntiles);
    return check_launch("deconv3d_mfma_bf16x3_kernel");
}

int conv3d_dispatch_bf16x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W,
                           int kd, int sd, int sh, int sw, int relu, hipStream_t st) {
#define MVS_X(CI, CO, KD, SD, SH, SW, TD, TH, CH)                                                     \
    if (Cin == CI && Cout == CO && kd == KD && sd == SD && sh == SH && sw == SW)                      \
        return launch_conv_bf<ConvCfg<CI, CO, KD, SD, SH, SW, TD, TH, CH>>(x, wp, bias, y, B, D, H, W, relu, st);
    MVS_CONV_TABLE(MVS_X)
#undef MVS_X
    set_error("conv3d(bf16x3): no kernel for Cin=%d Cout=%d kernel=(%d,3,3) stride=(%d,%d,%d)", Cin, Cout, kd, sd, sh, sw);
    return MVS_ERR_UNSUPPORTED;
}

int deconv3d_dispatch_bf16x3(const float* x, const void* wp, const float* bias, const float* skip, float* y, int B, int Cin, int Cout,
                             int D, int H, int W, int sd, hipStream_t st) {
#define MVS_X(CI, CO, SD, TDM, THM)                                                                     \
    if (Cin == CI && Cout == CO && sd == SD) return launch_deconv_bf<DeconvCfg<CI, CO, SD, TDM, THM>>(x, wp, bias, skip, y, B, D, H, W, st);
    MVS_DECONV_TABLE(MVS_X)
#undef MVS_X
    set_error("deconv3d(bf16x3): no kernel for Cin=%d Cout=%d stride=(%d,2,2)", Cin, Cout, sd);
    return MVS_ERR_UNSUPPORTED;
}

}  // namespace mvs
