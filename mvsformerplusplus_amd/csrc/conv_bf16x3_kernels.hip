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

// software-pipelined contraction of one staged channel chunk (see conv_kernels.hip for the pipeline rationale)
template <class Cfg>
__device__ __forceinline__ void bf_conv_contract(const bf16x8* wq, const char* ldsb, const int* voxbase, int g, f32x4 (*acc)[Cfg::NREP]) {
    constexpr int MREP = Cfg::MREP, NREP = Cfg::NREP, NSTEP = BfConv<Cfg>::NSTEP;
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
        bf_conv_contract<Cfg>(reinterpret_cast<const bf16x8*>(wp) + (size_t)pass * NSTEP * MREP * 2 * 64 + lane, ldsb, voxbase, g, acc);
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
// Visibility CNN fused into two launches (cost_volume.py:36,93):
//   vis_front: entropy [N,H,W] -> ConvBnReLU(1,16) computed on the fly while the 18x18 input tile of ConvBnReLU(16,16)
//              is staged (the 16-channel first activation never reaches HBM) -> [N,H,W,16]
//   vis_back : ConvBnReLU(16,8) with Conv2d(8,1,1) + Sigmoid folded into the epilogue (the 8 output channels of a
//              pixel live in two lane groups of the MFMA result: one wave shuffle) -> vis [N,H,W]
// ------------------------------------------------------------------------------------------------
typedef ConvCfg<16, 16, 1, 1, 1, 1, 1, 16, 16> VisCfgA;
typedef ConvCfg<16, 8, 1, 1, 1, 1, 1, 16, 16> VisCfgB;

__global__ __launch_bounds__(256) void vis_front_bf16x3_kernel(const float* __restrict__ ent, const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const void* wp2, const float* __restrict__ bias2, float* __restrict__ y, int H, int W,
                                                               int tiles_x, int ntiles_per_view) {
    typedef VisCfgA Cfg;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, NREP = Cfg::NREP, SB = BfConv<Cfg>::SB, EW = IW + 2;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    float* ent_s = reinterpret_cast<float*>(ldsb + Cfg::NVOX * SB);          // (IH+2) x (IW+2) entropy tile
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles_per_view);
    const int n = (int)blockIdx.y;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const float* e = ent + (size_t)n * H * W;
    for (int i = tid; i < (IH + 2) * EW; i += 256) {
        const int dy = i / EW, dx = i - dy * EW;
        const int yy = oy0 - 2 + dy, xx = ox0 - 2 + dx;
        ent_s[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? e[(size_t)yy * W + xx] : 0.0f;
    }
    __syncthreads();
    // first layer (1 -> 16, 3x3, folded BN, ReLU) for the 18x18 tile, split into bf16 hi/lo, 8 channels per iteration.
    // i & 1 == tid & 1: a work-item always produces the same 8 channels, so their 72 weights + 8 biases sit in registers
    // (indexing w1 by a lane-dependent octet made every tap a vector load: TA 79 % busy in the PMC profile)
    float w1r[9][8], b1r[8];
    {
        const int oc = tid & 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 a = *reinterpret_cast<const float4*>(w1 + t * 16 + oc * 8), c = *reinterpret_cast<const float4*>(w1 + t * 16 + oc * 8 + 4);
            w1r[t][0] = a.x; w1r[t][1] = a.y; w1r[t][2] = a.z; w1r[t][3] = a.w; w1r[t][4] = c.x; w1r[t][5] = c.y; w1r[t][6] = c.z; w1r[t][7] = c.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) b1r[j] = b1[oc * 8 + j];
    }
    for (int i = tid; i < Cfg::NVOX * 2; i += 256) {
        const int vox = i >> 1, oc = i & 1;
        const int dx = vox % IW, dy = vox / IW;
        const int yy = oy0 - 1 + dy, xx = ox0 - 1 + dx;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {                        // outside the image the 16->16 conv sees zero padding
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float ev = ent_s[(dy + kh) * EW + dx + kw];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ev * w1r[kh * 3 + kw][j];
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] + b1r[j], 0.0f);
        }
        bf16x8 hi, lo;
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32) = hi;
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32 + 16) = lo;
    }
    __syncthreads();
    f32x4 acc[1][NREP];
    int voxbase[NREP];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        acc[0][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        voxbase[nb] = ((wave * NREP + nb) * IW + li) * SB;
    }
    bf_conv_contract<Cfg>(reinterpret_cast<const bf16x8*>(wp2) + lane, ldsb, voxbase, g, acc);
    const float4 bb = *reinterpret_cast<const float4*>(bias2 + 4 * g);
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        const int oy = oy0 + wave * NREP + nb, ox = ox0 + li;
        if (oy >= H || ox >= W) continue;
        const float4 r = make_float4(fmaxf(acc[0][nb][0] + bb.x, 0.0f), fmaxf(acc[0][nb][1] + bb.y, 0.0f), fmaxf(acc[0][nb][2] + bb.z, 0.0f),
                                     fmaxf(acc[0][nb][3] + bb.w, 0.0f));
        *reinterpret_cast<float4*>(y + (((size_t)n * H + oy) * W + ox) * 16 + 4 * g) = r;
    }
}

__global__ __launch_bounds__(256) void vis_back_bf16x3_kernel(const float* __restrict__ x, const void* wp3, const float* __restrict__ bias3,
                                                              const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ vis,
                                                              int H, int W, int tiles_x, int ntiles_per_view) {
    typedef VisCfgB Cfg;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, NREP = Cfg::NREP, SB = BfConv<Cfg>::SB;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int tile = (int)xcd_remap(blockIdx.x, (unsigned)ntiles_per_view);
    const int n = (int)blockIdx.y;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const float* xb = x + (size_t)n * H * W * 16;
    for (int i = tid; i < Cfg::NVOX * 2; i += 256) {
        const int vox = i >> 1, oc = i & 1;
        const int dx = vox % IW, dy = vox / IW;
        const int yy = oy0 - 1 + dy, xx = ox0 - 1 + dx;
        float4 u = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v = u;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float4* src = reinterpret_cast<const float4*>(xb + ((size_t)yy * W + xx) * 16 + oc * 8);
            u = src[0];
            v = src[1];
        }
        bf16x8 hi, lo;
        split8(u, v, hi, lo);
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32) = hi;
        *reinterpret_cast<bf16x8*>(ldsb + vox * SB + oc * 32 + 16) = lo;
    }
    __syncthreads();
    f32x4 acc[1][NREP];
    int voxbase[NREP];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        acc[0][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        voxbase[nb] = ((wave * NREP + nb) * IW + li) * SB;
    }
    bf_conv_contract<Cfg>(reinterpret_cast<const bf16x8*>(wp3) + lane, ldsb, voxbase, g, acc);
    // epilogue: relu(acc + bias3) for channels 4g..4g+3 (g < 2), dot with the 1x1 weights, add the other lane group, sigmoid
    const int cg = g < 2 ? g : 0;
    const float4 bb = *reinterpret_cast<const float4*>(bias3 + 4 * cg);
    const float4 ww = *reinterpret_cast<const float4*>(w4 + 4 * cg);
    const float bias4 = b4[0];
#pragma unroll
    for (int nb = 0; nb < NREP; ++nb) {
        float part = fmaxf(acc[0][nb][0] + bb.x, 0.0f) * ww.x;
        part += fmaxf(acc[0][nb][1] + bb.y, 0.0f) * ww.y;
        part += fmaxf(acc[0][nb][2] + bb.z, 0.0f) * ww.z;
        part += fmaxf(acc[0][nb][3] + bb.w, 0.0f) * ww.w;
        if (g >= 2) part = 0.0f;                                             // rows 8..15 of the 16-row MFMA are padding
        part += __shfl_xor(part, 16);
        const int oy = oy0 + wave * NREP + nb, ox = ox0 + li;
        if (g == 0 && oy < H && ox < W) vis[((size_t)n * H + oy) * W + ox] = 1.0f / (1.0f + expf(-(part + bias4)));
    }
}

int vis_weight_fused_bf16x3(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                            const float* w4, const float* b4, float* vis, float* scratch16, int N, int H, int W, hipStream_t st) {
    const int tx = (int)ceil_div(W, 16), ty = (int)ceil_div(H, 16);
    const size_t ldsA = VisCfgA::LDS_BYTES + (VisCfgA::IH + 2) * (VisCfgA::IW + 2) * sizeof(float);
    hipLaunchKernelGGL(vis_front_bf16x3_kernel, dim3(tx * ty, N), dim3(256), ldsA, st, entropy, w1, b1, w2, b2, scratch16, H, W, tx, tx * ty);
    int rc = check_launch("vis_front_bf16x3_kernel");
    if (rc != MVS_OK) return rc;
    hipLaunchKernelGGL(vis_back_bf16x3_kernel, dim3(tx * ty, N), dim3(256), VisCfgB::LDS_BYTES, st, scratch16, w3, b3, w4, b4, vis, H, W, tx, tx * ty);
    return check_launch("vis_back_bf16x3_kernel");
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
                                                                   const float* __restrict__ skip, float* __restrict__ y,
                                                                   const float* __restrict__ prob_w, const float* __restrict__ prob_b,
                                                                   float* __restrict__ logits, int D, int H, int W, int tiles_x,
                                                                   int tiles_y, int ntiles) {
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

    // COUT == 8: the two x-parity classes of a (pd, ph) pair ride in one MFMA (rows 0-7: pw = 0, rows 8-15: pw = 1, tap set
    // of pw = 1; weights packed accordingly, packing.pack_deconv_weights_bf16x3)
    constexpr bool PAIR = COUT == 8;
    constexpr int NCLS = (SD == 2 ? 2 : 1) * 4;
    // The skip tensor is the largest read of the layer (same size as the output) and it is only needed by the epilogue: all of its
    // float4s are requested here, before the contraction, so that their HBM latency runs under the MFMA work instead of stalling
    // every class's epilogue (PMC: these kernels sat 66-78 % of their wave cycles in s_waitcnt).
    constexpr int NIT = PAIR ? NCLS / 2 : NCLS;
    float4 skp[NIT][NREP][MREP];
    if (sb) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cls = PAIR ? 2 * it : it;
            const int pw = PAIR ? 1 : (cls & 1), ph = (cls >> 1) & 1, pd = (SD == 2) ? (cls >> 2) : 0;
#pragma unroll
            for (int nb = 0; nb < NREP; ++nb) {
                const int nbg = wave * NREP + nb;
                const int mz = mz0 + nbg / THM, my = my0 + nbg % THM, mx = mx0 + li;
                const bool inside = mz < D && my < H && mx < W;
                const int oz = mz * SD + pd, oy = 2 * my + ph, ox = PAIR ? 2 * mx + (g >> 1) : 2 * mx + pw;
#pragma unroll
                for (int mb = 0; mb < MREP; ++mb) {
                    const int co = PAIR ? 4 * (g & 1) : 16 * mb + 4 * g;
                    skp[it][nb][mb] = (inside && co < COUT) ? *reinterpret_cast<const float4*>(sb + (((size_t)oz * OH + oy) * OW + ox) * COUT + co)
                                                            : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        }
    }
#pragma unroll
    for (int cls = 0; cls < NCLS; cls += PAIR ? 2 : 1) {
        const int it = PAIR ? cls / 2 : cls;
        const int pw = PAIR ? 1 : (cls & 1), ph = (cls >> 1) & 1, pd = (SD == 2) ? (cls >> 2) : 0;
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
            const bool inside = mz < D && my < H && mx < W;
            if (!inside && !(PAIR && prob_w != nullptr)) continue;     // the fused head shuffles: every lane takes part, stores are guarded
            if (PAIR) {
                // lane groups 0/1: channels 0-3 / 4-7 of output voxel 2mx; groups 2/3: the same of voxel 2mx + 1
                const int oz = mz * SD + pd, oy = 2 * my + ph, ox = 2 * mx + (g >> 1), co = 4 * (g & 1);
                const size_t off = (((size_t)oz * OH + oy) * OW + ox) * COUT + co;
                const float4 bb = *reinterpret_cast<const float4*>(bias + co);
                float4 v = make_float4(fmaxf(acc[0][nb][0] + bb.x, 0.0f), fmaxf(acc[0][nb][1] + bb.y, 0.0f), fmaxf(acc[0][nb][2] + bb.z, 0.0f),
                                       fmaxf(acc[0][nb][3] + bb.w, 0.0f));
                if (sb && inside) {
                    const float4 sk = skp[it][nb][0];
                    v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
                }
                if (prob_w != nullptr) {
                    // fused 1x1x1 `prob` head (module.py:486,502): logit = sum_c w[c] * feat[c] + b; the voxel's 8 channels sit in
                    // two lane groups (g, g ^ 1): one cross-lane add.  The 8-channel feature volume never reaches HBM.
                    const float4 pw4 = *reinterpret_cast<const float4*>(prob_w + co);
                    float part = v.x * pw4.x;
                    part += v.y * pw4.y;
                    part += v.z * pw4.z;
                    part += v.w * pw4.w;
                    part += __shfl_xor(part, 16);
                    if ((g & 1) == 0 && inside) logits[(size_t)b * OD * OH * OW + ((size_t)oz * OH + oy) * OW + ox] = part + prob_b[0];
                } else {
                    *reinterpret_cast<float4*>(yb + off) = v;
                }
                continue;
            }
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
                    const float4 sk = skp[it][nb][mb];
                    v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
                }
                *reinterpret_cast<float4*>(yb + off + co) = v;
            }
        }
    }
}

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
static int launch_deconv_bf(const float* x, const void* wp, const float* bias, const float* skip, float* y, const float* prob_w,
                            const float* prob_b, float* logits, int B, int D, int H, int W, hipStream_t st) {
    const int tx = (int)ceil_div(W, 16), ty = (int)ceil_div(H, Cfg::THM), tz = (int)ceil_div(D, Cfg::TDM);
    const int ntiles = tx * ty * tz;
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3d_mfma_bf16x3_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL((deconv3d_mfma_bf16x3_kernel<Cfg>), dim3(ntiles, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, skip, y, prob_w, prob_b,
                       logits, D, H, W, tx, ty, ntiles);
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
                             int D, int H, int W, int sd, hipStream_t st, const float* prob_w, const float* prob_b, float* logits) {
    if (prob_w != nullptr && Cout != 8) { set_error("deconv3d(bf16x3): the fused prob head needs Cout == 8"); return MVS_ERR_UNSUPPORTED; }
#define MVS_X(CI, CO, SD, TDM, THM)                                                                     \
    if (Cin == CI && Cout == CO && sd == SD)                                                            \
        return launch_deconv_bf<DeconvCfg<CI, CO, SD, TDM, THM>>(x, wp, bias, skip, y, prob_w, prob_b, logits, B, D, H, W, st);
    MVS_DECONV_TABLE(MVS_X)
#undef MVS_X
    set_error("deconv3d(bf16x3): no kernel for Cin=%d Cout=%d stride=(%d,2,2)", Cin, Cout, sd);
    return MVS_ERR_UNSUPPORTED;
}

}  // namespace mvs
