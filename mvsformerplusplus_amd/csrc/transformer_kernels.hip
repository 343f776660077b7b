// Stage-1 transformer regulariser of the shipped config:
//   PureTransformerCostReg  module.py:602-646      patch embedding (Conv3d k = stride = (2,4,4)) + LayerNorm3D, 6 post-norm
//   FlashAttnBlock          module.py:535-583      blocks [softmax attention, 4 heads x 16; FFN 64 -> 256 -> 64, exact GELU],
//   attention               dino/layers/attention.py:76-101,141-170    ConvTranspose3d back to the volume + LayerNorm3D + 1x1x1 prob
//   Frustoconical PE        position_encoding.py:138-189
//
// Everything between the cost volume and the logits is token-local GEMM work on <= 64 channels plus one softmax
// attention over all N tokens (N = 27 648 at 1152x1536).  Three kernel families:
//   * tr_gemm_kernel      64 tokens x K inputs per work-group, split-bf16 ("bf16x3") contraction on v_mfma_f32_16x16x32_bf16
//                         exactly like the 3D convolutions (conv_bf16x3_kernels.hip): activations split hi/lo while the tile
//                         is staged into LDS, weights split on the host in per-lane operand order.  Prologues: token rows /
//                         patch gather (+ position encoding).  Epilogues: bias, GELU, residual + LayerNorm, LayerNorm,
//                         q/k/v operand writer, up-projection + LayerNorm3D + prob.
//   * tr_attention_kernel flash attention, one wave per 16 queries of one head, keys streamed through LDS 64 at a time.
//                         q.k uses the 32-wide MFMA k-axis for [hi | lo] halves: two MFMAs give the full
//                         (k_hi + k_lo).(q_hi + q_lo); p.v is the usual three-term split.  Scores leave the matrix core in
//                         exactly the per-lane order the p.v product wants them in, so P never moves between lanes.
//   * pos3d_*             normalised frustum coordinates of every hypothesis voxel (two-pass min/max + write).
#include <stdlib.h>

#include "mvs_common.h"
#include "attention_f16.h"

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void tr_split8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}

__device__ __forceinline__ float wave_sum_groups(float v) {      // sum over the four lane groups g = lane >> 4
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace

enum { PRO_TOKENS = 0, PRO_PATCH = 1 };
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RES_LN = 2, EPI_LN = 3, EPI_QKV = 4, EPI_UP = 5, EPI_QKV16 = 6 };

struct TrArgs {
    const float* x;          // PRO_TOKENS: [B, n, K] token rows;  PRO_PATCH: cost volume [B, D, H, W, 8] channel-last
    const float* pos;        // PRO_PATCH: normalised frustum position [B, 3, D, H, W] or nullptr
    const float* pe_w;       // PRO_PATCH: pe_proj.weight [8][24]
    float pe_div[4];         // PRO_PATCH: frequencies of PositionEncoding3D (position_encoding.py:171)
    const void* w;           // packed split-bf16 weights, packing.pack_linear_bf16x3
    const float* bias;       // [N] or nullptr  (EPI_UP: up.0.bias [8])
    const float* res;        // EPI_RES_LN: residual rows [B, n, 64]
    const float* gamma;      // EPI_RES_LN: layer scale [1]
    const float* ln_w;       // LayerNorm weight / bias ([64]; EPI_UP: [8])
    const float* ln_b;
    float eps;
    float* y;                // [B, n, N]  (EPI_UP: logits [B, D, H, W])
    __bf16* q;               // EPI_QKV: [B, heads, npad, 32] = [hi16 | lo16], pre-multiplied by qscale
    __bf16* k;               //          [B, heads, npad, 32]
    __bf16* vt;              //          [B, heads, 2 (hi, lo), 16, npad]
    _Float16* q16;           // EPI_QKV16 (attention_f16_kernels.hip): Q [B, heads, npad, 16] fp16, pre-multiplied by qscale
    _Float16* k16;           //            KP [B, heads, npad/32, 4, 16, 2, 4] fp16: dims 4g..4g+3 of key 32 step + 16 t + j at [step][g][j][t]
    __bf16* v16;             //            VP [B, heads, npad/32, 4, 16, 8] bf16: v[key 32 step + 16 (e >> 2) + 4g + (e & 3)][d] at [step][g][d][e]
    float qscale;
    int heads, npad;
    const float* prob_w;     // EPI_UP: prob.weight [8], prob.bias [1]
    const float* prob_b;
    int n;                   // tokens per batch item
    int N;                   // output features
    int D, H, W, rd, rh, rw, Ht, Wt;     // volume and patch geometry (PRO_PATCH, EPI_UP); token = (td*Ht + th)*Wt + tw
};

// --------------------------------------------------------------------------------------------------
// token GEMM: y[tok][f] = sum_k W[f][k] * x[tok][k]  (+ epilogue).  grid = (ceil(n / 64), B), 256 threads.
// --------------------------------------------------------------------------------------------------
template <int K, int PRO, int EPI>
__global__ __launch_bounds__(256) void tr_gemm_kernel(const TrArgs a) {
    constexpr int OCT = K / 8, NSTEP = K / 32, SB = K * 4 + 16;          // LDS bytes per token: [octet][hi8 | lo8] + pad
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = (int)blockIdx.y, tok0 = (int)blockIdx.x * 64;
    const int n = a.n;

    // ---- stage + split the 64 x K activation tile ----
    if (PRO == PRO_TOKENS) {
        const float* xb = a.x + (size_t)b * n * K;
        for (int e = tid; e < 64 * OCT; e += 256) {
            const int tk = e / OCT, oc = e - tk * OCT, tok = tok0 + tk;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tok < n) {
                const float4* p = reinterpret_cast<const float4*>(xb + (size_t)tok * K + oc * 8);
                const float4 u = p[0], w = p[1];
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; v[4] = w.x; v[5] = w.y; v[6] = w.z; v[7] = w.w;
            }
            bf16x8 hi, lo;
            tr_split8(v, hi, lo);
            *reinterpret_cast<bf16x8*>(ldsb + tk * SB + oc * 32) = hi;
            *reinterpret_cast<bf16x8*>(ldsb + tk * SB + oc * 32 + 16) = lo;
        }
    } else {
        // one octet = the 8 channels of one voxel of the token's rd x rh x rw patch (k = patch_voxel * 8 + channel)
        float* pe_w = reinterpret_cast<float*>(ldsb + 64 * SB);                       // [8][24]
        if (a.pos != nullptr) {
            for (int i = tid; i < 192; i += 256) pe_w[i] = a.pe_w[i];
            __syncthreads();
        }
        const size_t HW = (size_t)a.H * a.W;
        for (int e = tid; e < 64 * OCT; e += 256) {
            const int tk = e / OCT, pv = e - tk * OCT, tok = tok0 + tk;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tok < n) {
                const int tw = tok % a.Wt, t2 = tok / a.Wt, th = t2 % a.Ht, td = t2 / a.Ht;
                const int kw = pv % a.rw, p2 = pv / a.rw, kh = p2 % a.rh, kd = p2 / a.rh;
                const int d = td * a.rd + kd, h = th * a.rh + kh, w = tw * a.rw + kw;
                const size_t vox = ((size_t)(b * a.D + d) * a.H + h) * a.W + w;
                const float4* p = reinterpret_cast<const float4*>(a.x + vox * 8);
                const float4 u = p[0], q = p[1];
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
                if (a.pos != nullptr) {                                              // x + pe_proj(PositionEncoding3D(position3d))  module.py:633
                    float pe[24];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const float pp = a.pos[((size_t)(b * 3 + ax) * a.D + d) * HW + (size_t)h * a.W + w] * 4.0f;
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const float ang = pp * a.pe_div[f];
                            pe[ax * 8 + 2 * f] = sinf(ang);
                            pe[ax * 8 + 2 * f + 1] = cosf(ang);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float s = 0.0f;
#pragma unroll
                        for (int i = 0; i < 24; ++i) s += pe_w[c * 24 + i] * pe[i];
                        v[c] += s;
                    }
                }
            }
            bf16x8 hi, lo;
            tr_split8(v, hi, lo);
            *reinterpret_cast<bf16x8*>(ldsb + tk * SB + pv * 32) = hi;
            *reinterpret_cast<bf16x8*>(ldsb + tk * SB + pv * 32 + 16) = lo;
        }
    }
    __syncthreads();

    // ---- contract: 4 output tiles (64 features) at a time, A = weights from global / L2, B = tokens from LDS ----
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(a.w) + lane;
    const int MT = a.N / 16;
    const char* brow = ldsb + (wave * 16 + j) * SB + g * 32;
    const int tok = tok0 + wave * 16 + j;
    const bool valid = tok < n;
    for (int mc = 0; mc < MT; mc += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(brow + t * 128);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(brow + t * 128 + 16);
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                ah[mb] = wq[(size_t)((t * MT + mc + mb) * 2) * 64];
                al[mb] = wq[(size_t)((t * MT + mc + mb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh, acc[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl, acc[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh, acc[mb], 0, 0, 0);
        }

        // lane (j, g) now holds features 16*(mc + mb) + 4g + r, r = 0..3, of token j
        if (EPI == EPI_BIAS || EPI == EPI_GELU) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int f0 = 16 * (mc + mb) + 4 * g;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mb][r] + (a.bias != nullptr ? a.bias[f0 + r] : 0.0f);
                    if (EPI == EPI_GELU) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
                }
                if (valid) *reinterpret_cast<float4*>(a.y + ((size_t)b * n + tok) * a.N + f0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else if (EPI == EPI_RES_LN || EPI == EPI_LN) {
            // N == 64: the token's whole row is in this chunk, spread over the four lanes that share j
            float v[16];
            const float gm = EPI == EPI_RES_LN ? a.gamma[0] : 1.0f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int f0 = 16 * mb + 4 * g;
                float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (EPI == EPI_RES_LN && valid) rs = *reinterpret_cast<const float4*>(a.res + ((size_t)b * n + tok) * 64 + f0);
                const float rr[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lin = acc[mb][r] + a.bias[f0 + r];
                    v[mb * 4 + r] = EPI == EPI_RES_LN ? rr[r] + gm * lin : lin;
                }
            }
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += v[i];
            const float mean = wave_sum_groups(s) * (1.0f / 64.0f);
            float qq = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] -= mean; qq += v[i] * v[i]; }
            const float inv = 1.0f / sqrtf(wave_sum_groups(qq) * (1.0f / 64.0f) + a.eps);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int f0 = 16 * mb + 4 * g;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = v[mb * 4 + r] * inv * a.ln_w[f0 + r] + a.ln_b[f0 + r];
                if (valid) *reinterpret_cast<float4*>(a.y + ((size_t)b * n + tok) * 64 + f0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else if (EPI == EPI_QKV) {
            // tile = which * heads + head; the lane holds dims 4g..4g+3 of that (which, head) for token j.  Rows >= n of
            // the padded operand buffers receive zeros (their staged inputs were zero and qkv has no bias).
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int tile = mc + mb, which = tile / a.heads, hh = tile - which * a.heads;
                const size_t hb = (size_t)b * a.heads + hh;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = which == 0 ? acc[mb][r] * a.qscale : acc[mb][r];
                bf16x4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const __bf16 h = (__bf16)v[r];
                    hi[r] = h;
                    lo[r] = (__bf16)(v[r] - (float)h);
                }
                if (which < 2) {
                    __bf16* dst = (which == 0 ? a.q : a.k) + (hb * a.npad + tok) * 32 + 4 * g;
                    *reinterpret_cast<bf16x4*>(dst) = hi;
                    *reinterpret_cast<bf16x4*>(dst + 16) = lo;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a.vt[((hb * 2 + 0) * 16 + 4 * g + r) * (size_t)a.npad + tok] = hi[r];
                        a.vt[((hb * 2 + 1) * 16 + 4 * g + r) * (size_t)a.npad + tok] = lo[r];
                    }
                }
            }
        } else if (EPI == EPI_QKV16) {
            // the 16-bit operands of tr_attention16_kernel; the lane holds dims 4g..4g+3 of (which, head) for token j.  q and k are fp16,
            // clamped to the fp16 range (|x| <= 65504); v is bf16; rows >= n of the padded buffers receive zeros like EPI_QKV.
            typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int tile = mc + mb, which = tile / a.heads, hh = tile - which * a.heads;
                const size_t hb = (size_t)b * a.heads + hh;
                if (which < 2) {
                    f16x4_t hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = which == 0 ? acc[mb][r] * a.qscale : acc[mb][r];
                        hv[r] = (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
                    }
                    if (which == 0) *reinterpret_cast<f16x4_t*>(a.q16 + (hb * a.npad + tok) * 16 + 4 * g) = hv;
                    else *reinterpret_cast<f16x4_t*>(a.k16 + (((hb * (a.npad >> 5) + (tok >> 5)) * 4 + g) * 16 + (tok & 15)) * 8 + ((tok >> 4) & 1) * 4) = hv;
                } else {
                    const int kk = tok & 31, e = 4 * (kk >> 4) + (kk & 3), gk = (kk & 15) >> 2;
                    __bf16* dst = a.v16 + ((hb * (a.npad >> 5) + (tok >> 5)) * 4 + gk) * 128 + e;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(4 * g + r) * 8] = (__bf16)acc[mb][r];
                }
            }
        } else if (EPI == EPI_UP) {
            // feature = patch_voxel * 8 + channel: lanes g and g ^ 1 hold the two halves of one voxel's 8 channels
            const int tw = tok % a.Wt, t2 = tok / a.Wt, th = t2 % a.Ht, td = t2 / a.Ht;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int pv = 2 * (mc + mb) + (g >> 1), c0 = 4 * (g & 1);
                float v[4];
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[mb][r] + a.bias[c0 + r]; s += v[r]; }
                s += __shfl_xor(s, 16);
                const float mean = s * 0.125f;
                float qq = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] -= mean; qq += v[r] * v[r]; }
                qq += __shfl_xor(qq, 16);
                const float inv = 1.0f / sqrtf(qq * 0.125f + a.eps);
                float lg = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) lg += (v[r] * inv * a.ln_w[c0 + r] + a.ln_b[c0 + r]) * a.prob_w[c0 + r];
                lg += __shfl_xor(lg, 16);
                if (valid && (g & 1) == 0) {
                    const int kw = pv % a.rw, p2 = pv / a.rw, kh = p2 % a.rh, kd = p2 / a.rh;
                    const int d = td * a.rd + kd, h = th * a.rh + kh, w = tw * a.rw + kw;
                    a.y[((size_t)(b * a.D + d) * a.H + h) * a.W + w] = lg + a.prob_b[0];
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------
// flash attention over all n tokens.  grid = (npad / (64 QT), heads, B); wave w of a block owns 16 QT queries.
//
// Per 32 keys a wave issues 4 score MFMAs + 3 p.v MFMAs and ~8 v_exp_f32 per lane; everything else is kept off the
// common path:
//   * the running maximum m of a query is only raised when some score of the step exceeds it by more than 2^kLazy
//     (one wave ballot per step, no cross-lane traffic otherwise); probabilities may therefore reach 2^kLazy, which the
//     fp32 accumulators and the hi/lo split absorb.  "- m" is folded into the score MFMAs' accumulator input.
//   * K / V^T blocks of 64 keys are double-buffered in LDS: global loads of block i+1 are issued before the math of
//     block i and written after it, one barrier per block
//   * K rows are stored chunk-rotated so that the 16 lanes of a group hit 16 different bank quads
// --------------------------------------------------------------------------------------------------
constexpr int kVRow = 64 * 2 + 16;          // bytes of one (hi|lo, dim) row of the staged V^T block, skewed
constexpr float kLazy = 8.0f;

// P1: probabilities enter p.v as ONE bf16 term (2^-9 relative rounding, unbiased; v keeps hi + lo, scores stay four-term):
// -20 VALU instructions and -1 MFMA per step.  MVS_PREC_BF16P.
// QT: query tiles of 16 per wave; the K and V^T operands of a step are read from LDS once and used by all QT tiles, whose
// independent score / softmax / p.v chains interleave in the issue stream.
constexpr int kAttnQT = 1;      // measured: 2 tiles per wave 1.25 ms vs 1.12 ms per layer at 27 648 tokens (fewer, fatter waves lose more than the shared LDS reads gain)

template <bool P1, int QT>
__global__ __launch_bounds__(256) void tr_attention_kernel(const __bf16* __restrict__ Q, const __bf16* __restrict__ Kb,
                                                           const __bf16* __restrict__ Vt, float* __restrict__ out, int n, int npad,
                                                           int heads) {
    __shared__ float4 kl4[2][64 * 64 / 16];              // 64 keys x [hi16 | lo16] bf16, 16-byte chunks rotated by key >> 2
    __shared__ float4 vl4[2][32 * kVRow / 16];           // [hi | lo][16 dims][64 keys] bf16
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int hh = (int)blockIdx.y, b = (int)blockIdx.z;
    const size_t hb = (size_t)b * heads + hh;
    const int q0 = ((int)blockIdx.x * 4 + wave) * 16 * QT;

    // B operands of the score product: lane groups 0/1 carry dims 0-7 / 8-15, groups 2/3 repeat them (they meet k_lo)
    bf16x8 qh[QT], ql[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const __bf16* qrow = Q + (hb * npad + q0 + 16 * qt + j) * 32;
        qh[qt] = *reinterpret_cast<const bf16x8*>(qrow + 8 * (g & 1));
        ql[qt] = *reinterpret_cast<const bf16x8*>(qrow + 16 + 8 * (g & 1));
    }

    // staging roles: K chunk tid of the block (key = tid >> 2, 16-byte chunk tid & 3); V^T row tid >> 3, chunk tid & 7
    const float4* ksrc = reinterpret_cast<const float4*>(Kb + hb * npad * 32) + tid;
    const int kdst = 4 * (tid >> 2) + (((tid & 3) + (tid >> 4)) & 3);
    const float4* vsrc = reinterpret_cast<const float4*>(Vt + (hb * 32 + (tid >> 3)) * (size_t)npad) + (tid & 7);
    const int vdst = (tid >> 3) * kVRow + (tid & 7) * 16;
    // operand addresses inside a block
    const int koff0 = (4 * j + ((g + (j >> 2)) & 3)) * 16;                     // key j of a tile; tiles are 16 keys = 1024 bytes apart
    const int voff = j * kVRow + 8 * g;

    float4 kreg = ksrc[0], vreg = vsrc[0];
    kl4[0][kdst] = kreg;
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(vl4[0]) + vdst) = vreg;
    __syncthreads();

    f32x4 o[QT];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { o[qt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; m[qt] = 0.0f; l[qt] = 0.0f; }
    int buf = 0;
    for (int kb = 0; kb < npad; kb += 64, buf ^= 1) {
        const bool more = kb + 64 < npad;
        if (more) {                                                            // in flight during this block's math
            kreg = ksrc[(size_t)(kb + 64) * 4];
            vreg = vsrc[(kb + 64) / 8];
        }
        const char* kl = reinterpret_cast<const char*>(kl4[buf]);
        const char* vl = reinterpret_cast<const char*>(vl4[buf]);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            // operands shared by the wave's query tiles: A = [k_hi | k_lo] of key 32*sb + 16*tile + j; V^T row of dim j with the
            // key slots e <-> (tile e / 4, row 4g + e % 4), the order the scores come in
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kl + sb * 2048 + koff0);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kl + sb * 2048 + 1024 + koff0);
            const char* vr = vl + voff + sb * 64;
            const bf16x4 vh0 = *reinterpret_cast<const bf16x4*>(vr), vh1 = *reinterpret_cast<const bf16x4*>(vr + 32);
            const bf16x4 vl0 = *reinterpret_cast<const bf16x4*>(vr + 16 * kVRow), vl1 = *reinterpret_cast<const bf16x4*>(vr + 16 * kVRow + 32);
            const bf16x8 vh = __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vlo = __builtin_shufflevector(vl0, vl1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bool first = kb == 0 && sb == 0;                             // m starts at 0: the first step sets it, whatever the sign
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                // S^T tiles (keys x queries) minus the running maximum
                f32x4 s0 = (f32x4){-m[qt], -m[qt], -m[qt], -m[qt]}, s1 = s0;
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qh[qt], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qh[qt], s1, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, ql[qt], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, ql[qt], s1, 0, 0, 0);
                float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                if (kb + 64 > n) {                                               // padded keys of the last block
                    const int key0 = kb + 32 * sb + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (key0 + r >= n) s[r] = -INFINITY;
                        if (key0 + 16 + r >= n) s[4 + r] = -INFINITY;
                    }
                }
                const float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
                if (first || __any(mx > kLazy)) {
                    // raise the maximum of every query of the tile to its step maximum (shared by the 4 lanes of the query)
                    float d = first ? mx : fmaxf(mx, 0.0f);
                    d = fmaxf(d, __shfl_xor(d, 16));
                    d = fmaxf(d, __shfl_xor(d, 32));
                    const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
                    m[qt] += d;
                    l[qt] *= alpha;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qt][r] *= alpha;
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] -= d;
                }
                float p[8], ps = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { p[e] = __builtin_amdgcn_exp2f(s[e]); ps += p[e]; }
                l[qt] += ps;
                bf16x8 ph, pl;
                if (P1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ph[e] = (__bf16)p[e];
                } else {
                    tr_split8(p, ph, pl);
                }
                o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vlo, ph, o[qt], 0, 0, 0);
                if (!P1) o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[qt], 0, 0, 0);
                o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[qt], 0, 0, 0);
            }
        }
        if (more) {
            kl4[buf ^ 1][kdst] = kreg;
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(vl4[buf ^ 1]) + vdst) = vreg;
        }
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float lt = wave_sum_groups(l[qt]);
        const int tok = q0 + 16 * qt + j;
        if (tok < n) {
            const float inv = 1.0f / lt;
            *reinterpret_cast<float4*>(out + (((size_t)b * n + tok) * heads + hh) * 16 + 4 * g) =
                make_float4(o[qt][0] * inv, o[qt][1] * inv, o[qt][2] * inv, o[qt][3] * inv);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// get_position_3d (position_encoding.py:138-163): X = K^-1 [x, y, 1] * depth, x / y min-max normalised over the whole
// volume (range of stage 1 reused by later stages), z clamped and normalised by the range of depth_values.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inverse3x3(const float* k, float* inv) {
    const float c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
    const float det = k[0] * c00 + k[1] * c01 + k[2] * c02;
    const float r = 1.0f / det;
    inv[0] = c00 * r; inv[1] = (k[2] * k[7] - k[1] * k[8]) * r; inv[2] = (k[1] * k[5] - k[2] * k[4]) * r;
    inv[3] = c01 * r; inv[4] = (k[0] * k[8] - k[2] * k[6]) * r; inv[5] = (k[2] * k[3] - k[0] * k[5]) * r;
    inv[6] = c02 * r; inv[7] = (k[1] * k[6] - k[0] * k[7]) * r; inv[8] = (k[0] * k[4] - k[1] * k[3]) * r;
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : fminf(v, w);
    }
    const int wave = (int)threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : fminf(r, sh[i]);
    return r;
}

// partial[block] = {min Y, max Y, min X, max X} of the block's voxels
__global__ __launch_bounds__(256) void pos3d_partial_kernel(const float* __restrict__ Km, const float* __restrict__ hyp,
                                                            float* __restrict__ partial, int B, int D, int H, int W) {
    __shared__ float sh[4];
    const size_t HW = (size_t)H * W, total = (size_t)B * D * HW;
    float mn_y = INFINITY, mx_y = -INFINITY, mn_x = INFINITY, mx_x = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i % HW;
        const int b = (int)(i / (HW * D));
        float inv[9];
        inverse3x3(Km + b * 9, inv);
        const float x = (float)(pix % W), y = (float)(pix / W), d = hyp[i];
        const float X = (inv[0] * x + inv[1] * y + inv[2]) * d, Y = (inv[3] * x + inv[4] * y + inv[5]) * d;
        mn_x = fminf(mn_x, X); mx_x = fmaxf(mx_x, X); mn_y = fminf(mn_y, Y); mx_y = fmaxf(mx_y, Y);
    }
    const float r0 = block_reduce(mn_y, false, sh), r1 = block_reduce(mx_y, true, sh);
    const float r2 = block_reduce(mn_x, false, sh), r3 = block_reduce(mx_x, true, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = r0; partial[blockIdx.x * 4 + 1] = r1;
        partial[blockIdx.x * 4 + 2] = r2; partial[blockIdx.x * 4 + 3] = r3;
    }
}

// range = {height_min, height_max, width_min, width_max, depth_min, depth_max}; the first four only when nblk > 0
__global__ __launch_bounds__(256) void pos3d_final_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ depth_values,
                                                          int ndv, float* __restrict__ range) {
    __shared__ float sh[4];
    float v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = (int)threadIdx.x; i < nblk; i += 256)
        for (int c = 0; c < 4; ++c) v[c] = (c & 1) ? fmaxf(v[c], partial[i * 4 + c]) : fminf(v[c], partial[i * 4 + c]);
    float dmn = INFINITY, dmx = -INFINITY;
    for (int i = (int)threadIdx.x; i < ndv; i += 256) { dmn = fminf(dmn, depth_values[i]); dmx = fmaxf(dmx, depth_values[i]); }
    float r[6];
    for (int c = 0; c < 4; ++c) r[c] = block_reduce(v[c], (c & 1) != 0, sh);
    r[4] = block_reduce(dmn, false, sh);
    r[5] = block_reduce(dmx, true, sh);
    if (threadIdx.x == 0) {
        if (nblk > 0) for (int c = 0; c < 4; ++c) range[c] = r[c];
        range[4] = r[4];
        range[5] = r[5];
    }
}

__global__ __launch_bounds__(256) void pos3d_write_kernel(const float* __restrict__ Km, const float* __restrict__ hyp,
                                                          const float* __restrict__ range, float* __restrict__ pos, int B, int D, int H,
                                                          int W) {
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW, total = (size_t)B * DHW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t pix = i % HW, r = i % DHW;
    const int b = (int)(i / DHW);
    float inv[9];
    inverse3x3(Km + b * 9, inv);
    const float x = (float)(pix % W), y = (float)(pix / W), d = hyp[i];
    const float X = (inv[0] * x + inv[1] * y + inv[2]) * d, Y = (inv[3] * x + inv[4] * y + inv[5]) * d;
    const float Z = (inv[6] * x + inv[7] * y + inv[8]) * d;
    const float hmin = range[0], hmax = range[1], wmin = range[2], wmax = range[3], dmin = range[4], dmax = range[5];
    float* pb = pos + (size_t)b * 3 * DHW + r;
    pb[0] = (X - wmin) / (wmax - wmin + 1e-5f);
    pb[DHW] = (Y - hmin) / (hmax - hmin + 1e-5f);
    pb[2 * DHW] = (fminf(fmaxf(Z, dmin), dmax) - dmin) / (dmax - dmin + 1e-5f);
}

// get_position_3d(normalize=False): the frustum points themselves, K^-1 [x, y, 1] * depth (position_encoding.py:146-149)
__global__ __launch_bounds__(256) void pos3d_raw_kernel(const float* __restrict__ Km, const float* __restrict__ hyp, float* __restrict__ pos, int B,
                                                        int D, int H, int W) {
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW, total = (size_t)B * DHW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t pix = i % HW, r = i % DHW;
    const int b = (int)(i / DHW);
    float inv[9];
    inverse3x3(Km + b * 9, inv);
    const float x = (float)(pix % W), y = (float)(pix / W), d = hyp[i];
    float* pb = pos + (size_t)b * 3 * DHW + r;
    pb[0] = (inv[0] * x + inv[1] * y + inv[2]) * d;
    pb[DHW] = (inv[3] * x + inv[4] * y + inv[5]) * d;
    pb[2 * DHW] = (inv[6] * x + inv[7] * y + inv[8]) * d;
}

// PositionEncoding3D as a tensor of its own (position_encoding.py:164-189; the hot path evaluates it inside the patch embedding and never
// writes it): pe[b, ax*C + 2f, n] = sin(pos[b, ax, n] * rescale * div[f]), pe[b, ax*C + 2f + 1, n] = cos(...), div[f] = exp(2f * (-ln 1e4 / C))
__global__ __launch_bounds__(256) void pos_encoding3d_kernel(const float* __restrict__ pos, const float* __restrict__ div_term, float* __restrict__ pe,
                                                             int C, float rescale, size_t N) {
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int f = (int)blockIdx.y, b = (int)blockIdx.z;
    if (n >= N) return;
    const float div = div_term[f];          // the caller's table: the reference's own fp32 values (position_encoding.py:169)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float ang = pos[((size_t)b * 3 + ax) * N + n] * rescale * div;
        float* o = pe + ((size_t)b * 3 * C + (size_t)ax * C + 2 * f) * N + n;
        o[0] = sinf(ang);
        o[N] = cosf(ang);
    }
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
template <int K, int PRO, int EPI>
static int launch_gemm(const TrArgs& a, int B, hipStream_t st, const char* what) {
    const size_t lds = (size_t)64 * (K * 4 + 16) + (PRO == PRO_PATCH ? 192 * sizeof(float) : 0);
    if (lds > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tr_gemm_kernel<K, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (EPI == EPI_QKV || EPI == EPI_QKV16) ? a.npad / 64 : (a.n + 63) / 64;
    hipLaunchKernelGGL((tr_gemm_kernel<K, PRO, EPI>), dim3(tiles, B), dim3(256), lds, st, a);
    return check_launch(what);
}

static bool only_bf16x3(int precision, const char* fn) {
    if (precision == MVS_PREC_BF16X3) return true;
    set_error("%s: the transformer kernels implement the split-bf16 (fp32-equivalent) contraction only, precision=%d", fn, precision);
    return false;
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_position3d_fwd(const float* K, const float* hyp, const float* depth_values, int n_depth_values, float* range,
                                  int compute_range, float* workspace, size_t workspace_bytes, float* position3d, int B, int D, int H,
                                  int W, void* stream) {
    if (!K || !hyp || !depth_values || !range || !position3d || B < 1 || D < 1 || H < 1 || W < 1 || n_depth_values < 1) {
        set_error("mvs_position3d_fwd: bad arguments");
        return MVS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * D * H * W;
    int nblk = 0;
    if (compute_range) {
        nblk = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
        if (!workspace || workspace_bytes < mvs_position3d_workspace_bytes()) { set_error("mvs_position3d_fwd: workspace too small"); return MVS_ERR_WORKSPACE; }
        hipLaunchKernelGGL(pos3d_partial_kernel, dim3(nblk), dim3(256), 0, st, K, hyp, workspace, B, D, H, W);
        const int rc = check_launch("pos3d_partial_kernel");
        if (rc != MVS_OK) return rc;
    }
    hipLaunchKernelGGL(pos3d_final_kernel, dim3(1), dim3(256), 0, st, workspace, nblk, depth_values, n_depth_values, range);
    int rc = check_launch("pos3d_final_kernel");
    if (rc != MVS_OK) return rc;
    hipLaunchKernelGGL(pos3d_write_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, K, hyp, range, position3d, B, D, H, W);
    return check_launch("pos3d_write_kernel");
}

extern "C" size_t mvs_position3d_workspace_bytes(void) { return (size_t)1024 * 4 * sizeof(float); }

extern "C" int mvs_position3d_raw_fwd(const float* K, const float* hyp, float* position3d, int B, int D, int H, int W, void* stream) {
    if (!K || !hyp || !position3d || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_position3d_raw_fwd: bad arguments"); return MVS_ERR_ARG; }
    const size_t total = (size_t)B * D * H * W;
    hipLaunchKernelGGL(pos3d_raw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, hyp, position3d, B, D, H, W);
    return check_launch("pos3d_raw_kernel");
}

extern "C" int mvs_position_encoding3d_fwd(const float* position3d, const float* div_term, float* pe, int B, int C, float rescale, long long N,
                                           void* stream) {
    if (!position3d || !div_term || !pe || B < 1 || B > 65535 || C < 2 || (C & 1) || C / 2 > 65535 || N < 1) { set_error("mvs_position_encoding3d_fwd: bad arguments (C even, >= 2)"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(pos_encoding3d_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)(C / 2), (unsigned)B), dim3(256), 0, (hipStream_t)stream, position3d,
                       div_term, pe, C, rescale, (size_t)N);
    return check_launch("pos_encoding3d_kernel");
}

extern "C" int mvs_tr_embed_fwd(const float* volume_cl, const float* position3d, const float* pe_w, const float* pe_div,
                                const void* w_packed, const float* bias, const float* ln_w, const float* ln_b, float* tokens, int B,
                                int D, int H, int W, int rd, int rh, int rw, int precision, void* stream) {
    if (!volume_cl || !w_packed || !bias || !ln_w || !ln_b || !tokens || (position3d && (!pe_w || !pe_div)) || B < 1 || rd < 1 || rh < 1 || rw < 1) {
        set_error("mvs_tr_embed_fwd: bad arguments");
        return MVS_ERR_ARG;
    }
    if (!only_bf16x3(precision, "mvs_tr_embed_fwd")) return MVS_ERR_UNSUPPORTED;
    if (D % rd || H % rh || W % rw) { set_error("mvs_tr_embed_fwd: volume %dx%dx%d is not a multiple of the patch %dx%dx%d", D, H, W, rd, rh, rw); return MVS_ERR_ARG; }
    TrArgs a = {};
    a.x = volume_cl; a.pos = position3d; a.pe_w = pe_w;
    if (position3d) for (int i = 0; i < 4; ++i) a.pe_div[i] = pe_div[i];
    a.w = w_packed; a.bias = bias; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = 1e-6f; a.y = tokens;
    a.D = D; a.H = H; a.W = W; a.rd = rd; a.rh = rh; a.rw = rw; a.Ht = H / rh; a.Wt = W / rw;
    a.n = (D / rd) * a.Ht * a.Wt; a.N = 64;
    const int K = 8 * rd * rh * rw;
    if (K == 256) return launch_gemm<256, PRO_PATCH, EPI_LN>(a, B, (hipStream_t)stream, "tr_gemm_kernel<embed>");
    if (K == 512) return launch_gemm<512, PRO_PATCH, EPI_LN>(a, B, (hipStream_t)stream, "tr_gemm_kernel<embed>");
    set_error("mvs_tr_embed_fwd: patch %dx%dx%d (K = %d) is not instantiated (shipped: 2x4x4; also 4x4x4)", rd, rh, rw, K);
    return MVS_ERR_UNSUPPORTED;
}

extern "C" int mvs_tr_linear_fwd(const float* x, const void* w_packed, const float* bias, int epilogue, const float* residual,
                                 const float* gamma, const float* ln_w, const float* ln_b, float ln_eps, float* y, int B, int n, int K,
                                 int N, int precision, void* stream) {
    if (!x || !w_packed || !y || B < 1 || n < 1) { set_error("mvs_tr_linear_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (!only_bf16x3(precision, "mvs_tr_linear_fwd")) return MVS_ERR_UNSUPPORTED;
    TrArgs a = {};
    a.x = x; a.w = w_packed; a.bias = bias; a.res = residual; a.gamma = gamma; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = ln_eps; a.y = y;
    a.n = n; a.N = N;
    hipStream_t st = (hipStream_t)stream;
    if (N % 64 != 0) { set_error("mvs_tr_linear_fwd: N = %d must be a multiple of 64", N); return MVS_ERR_UNSUPPORTED; }
    if (epilogue == MVS_TR_EPI_BIAS && K == 64) return launch_gemm<64, PRO_TOKENS, EPI_BIAS>(a, B, st, "tr_gemm_kernel<bias>");
    if (epilogue == MVS_TR_EPI_GELU && K == 64 && bias) return launch_gemm<64, PRO_TOKENS, EPI_GELU>(a, B, st, "tr_gemm_kernel<gelu>");
    if (epilogue == MVS_TR_EPI_RES_LN && N == 64 && bias && residual && gamma && ln_w && ln_b) {
        if (K == 64) return launch_gemm<64, PRO_TOKENS, EPI_RES_LN>(a, B, st, "tr_gemm_kernel<res_ln>");
        if (K == 256) return launch_gemm<256, PRO_TOKENS, EPI_RES_LN>(a, B, st, "tr_gemm_kernel<res_ln>");
    }
    set_error("mvs_tr_linear_fwd: combination epilogue=%d K=%d N=%d is not instantiated or misses an operand", epilogue, K, N);
    return MVS_ERR_UNSUPPORTED;
}

extern "C" size_t mvs_tr_attention_operand_bytes(int B, int n, int heads) {
    // enough for either operand format: [hi16 | lo16] bf16 rows padded to 64 tokens, or 16 fp16 per row padded to kAttnPad tokens
    const size_t npad = ((size_t)n + kAttnPad - 1) / kAttnPad * kAttnPad;
    return (size_t)B * heads * npad * 32 * 2;            // each of q, k, vt
}

extern "C" int mvs_tr_qkv_fwd(const float* x, const void* w_packed, void* q, void* k, void* vt, float softmax_scale, int B, int n,
                              int heads, int precision, int operand_format, void* stream) {
    if (!x || !w_packed || !q || !k || !vt || B < 1 || n < 1) { set_error("mvs_tr_qkv_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (!only_bf16x3(precision, "mvs_tr_qkv_fwd")) return MVS_ERR_UNSUPPORTED;
    if (heads != 4) { set_error("mvs_tr_qkv_fwd: built for 4 heads of 16 channels (shipped transformer_config), got %d heads", heads); return MVS_ERR_UNSUPPORTED; }
    TrArgs a = {};
    a.x = x; a.w = w_packed;
    a.qscale = softmax_scale * 1.44269504088896340736f;    // scores in base 2
    a.heads = heads; a.n = n; a.N = 3 * 16 * heads;
    if (operand_format == MVS_PREC_ATTN16) {
        a.q16 = static_cast<_Float16*>(q); a.k16 = static_cast<_Float16*>(k); a.v16 = static_cast<__bf16*>(vt);
        a.npad = (n + kAttnPad - 1) / kAttnPad * kAttnPad;
        // the GEMM runs over all npad rows: rows >= n are staged as zeros and written as zeros (no bias)
        return launch_gemm<64, PRO_TOKENS, EPI_QKV16>(a, B, (hipStream_t)stream, "tr_gemm_kernel<qkv16>");
    }
    if (operand_format != MVS_PREC_BF16X3 && operand_format != MVS_PREC_BF16P) {
        set_error("mvs_tr_qkv_fwd: operand_format must be MVS_PREC_BF16X3 (split-bf16 operands) or MVS_PREC_ATTN16, got %d", operand_format);
        return MVS_ERR_UNSUPPORTED;
    }
    a.q = static_cast<__bf16*>(q); a.k = static_cast<__bf16*>(k); a.vt = static_cast<__bf16*>(vt);
    a.npad = (n + 64 * kAttnQT - 1) / (64 * kAttnQT) * (64 * kAttnQT);
    return launch_gemm<64, PRO_TOKENS, EPI_QKV>(a, B, (hipStream_t)stream, "tr_gemm_kernel<qkv>");
}

extern "C" int mvs_tr_attention_fwd(const void* q, const void* k, const void* vt, float* out, int B, int n, int heads, int precision,
                                    void* stream) {
    if (!q || !k || !vt || !out || B < 1 || n < 1 || heads < 1) { set_error("mvs_tr_attention_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (precision == MVS_PREC_ATTN16) {
        // measurement / test switch: tile shapes 1 .. 6 of the SAME algorithm, every one numerically valid (tests/test_emu_parity.py runs them
        // all in one process, hence no caching; the wrong-result timing ablations need a -DMVS_ATTN_ABLATIONS build, attention_f16_kernels.hip)
        const char* ev = getenv("MVS_ATTN_VARIANT");
        const int variant = ev ? atoi(ev) : 0;
        return launch_attention16(q, k, vt, out, B, n, heads, variant, (hipStream_t)stream);
    }
    if (precision != MVS_PREC_BF16P && !only_bf16x3(precision, "mvs_tr_attention_fwd")) return MVS_ERR_UNSUPPORTED;
    const int npad = (n + 64 * kAttnQT - 1) / (64 * kAttnQT) * (64 * kAttnQT);
    const dim3 grid(npad / (64 * kAttnQT), heads, B);
    if (precision == MVS_PREC_BF16P)
        hipLaunchKernelGGL((tr_attention_kernel<true, kAttnQT>), grid, dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(q),
                           static_cast<const __bf16*>(k), static_cast<const __bf16*>(vt), out, n, npad, heads);
    else
        hipLaunchKernelGGL((tr_attention_kernel<false, kAttnQT>), grid, dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(q),
                           static_cast<const __bf16*>(k), static_cast<const __bf16*>(vt), out, n, npad, heads);
    return check_launch("tr_attention_kernel");
}

extern "C" int mvs_tr_up_prob_fwd(const float* tokens, const void* w_packed, const float* up_bias, const float* ln_w, const float* ln_b,
                                  const float* prob_w, const float* prob_b, float* logits, int B, int D, int H, int W, int rd, int rh,
                                  int rw, int precision, void* stream) {
    if (!tokens || !w_packed || !up_bias || !ln_w || !ln_b || !prob_w || !prob_b || !logits || B < 1 || rd < 1 || rh < 1 || rw < 1) {
        set_error("mvs_tr_up_prob_fwd: bad arguments");
        return MVS_ERR_ARG;
    }
    if (!only_bf16x3(precision, "mvs_tr_up_prob_fwd")) return MVS_ERR_UNSUPPORTED;
    if (D % rd || H % rh || W % rw || (8 * rd * rh * rw) % 64) { set_error("mvs_tr_up_prob_fwd: unsupported patch %dx%dx%d for volume %dx%dx%d", rd, rh, rw, D, H, W); return MVS_ERR_UNSUPPORTED; }
    TrArgs a = {};
    a.x = tokens; a.w = w_packed; a.bias = up_bias; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = 1e-6f; a.prob_w = prob_w; a.prob_b = prob_b; a.y = logits;
    a.D = D; a.H = H; a.W = W; a.rd = rd; a.rh = rh; a.rw = rw; a.Ht = H / rh; a.Wt = W / rw;
    a.n = (D / rd) * a.Ht * a.Wt; a.N = 8 * rd * rh * rw;
    return launch_gemm<64, PRO_TOKENS, EPI_UP>(a, B, (hipStream_t)stream, "tr_gemm_kernel<up>");
}

// ------------------------------------------------------------------------------------------------
// Backward of the attention core (training path; the reference differentiates
// F.scaled_dot_product_attention / flash-attn, dino/layers/attention.py:141-170):
//     o = softmax(scale * q k^T) v      ->      dq, dk, dv   from   d_o
// with  p_ij = exp(s_ij - lse_i),  D_i = sum_c d_o[i][c] o[i][c],  ds_ij = p_ij (d_o_i . v_j - D_i):
//     dq_i = scale * sum_j ds_ij k_j        dk_j = scale * sum_i ds_ij q_i        dv_j = sum_i p_ij d_o_i
// fp32 throughout (the reference's flash-attn backward keeps q, k, v, p in bf16).  Two launches, no atomics, deterministic:
//   pass Q: one work-item per QUERY row walks all keys twice (log-sum-exp, then dq); writes lse (base 2) and D for pass K
//   pass K: one work-item per KEY row walks all queries; dk and dv
// The row a work-item walks over is the same for the whole wave: its 16 + 16 floats arrive through scalar loads (one
// s_load_dwordx16 each) and enter the v_fma as SGPR operands - per (query, key) pair ~50 VALU instructions on registers, no LDS.
// Training token counts are a few thousand (DTU 512 x 640, D = 32 -> 2560 tokens): 0.3 ms per launch; at cfg2's 27 648 tokens ~7 ms.
// qkv is the projection's plain output [B][n][3][heads][16] (q | k | v), o / d_o are [B][n][heads*16], d_qkv has qkv's layout.
// ------------------------------------------------------------------------------------------------
constexpr float kLog2e = 1.4426950408889634f;

__global__ __launch_bounds__(256) void tr_attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                                                            float* __restrict__ d_qkv, float* __restrict__ lse2, float* __restrict__ dsum, int n,
                                                            int heads, float scale) {
    const int b = (int)blockIdx.z, h = (int)blockIdx.y;
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const bool valid = i < n;
    const int ii = valid ? i : n - 1;
    const int RS = 3 * heads * 16, OS = heads * 16;
    const float* base = qkv + (size_t)b * n * RS;
    const float* qrow = base + (size_t)ii * RS + h * 16;
    const float* orow = o + ((size_t)b * n + ii) * OS + h * 16;
    const float* grow = d_o + ((size_t)b * n + ii) * OS + h * 16;
    float qs[16], g[16], Dq = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        qs[c] = qrow[c] * (scale * kLog2e);                        // scores in base 2
        g[c] = grow[c];
        Dq += g[c] * orow[c];
    }
    const float* kbase = base + OS + h * 16;                        // key j at kbase + j * RS, value j at kbase + OS + j * RS
    float m = -INFINITY, l = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float* kj = kbase + (size_t)j * RS;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += qs[c] * kj[c];
        const float mn = fmaxf(m, s);
        l = l * __builtin_amdgcn_exp2f(m - mn) + __builtin_amdgcn_exp2f(s - mn);
        m = mn;
    }
    const float L = m + __builtin_amdgcn_logf(l);                  // log2 of the row's sum of 2^s
    float dq[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) dq[c] = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float* kj = kbase + (size_t)j * RS;
        const float* vj = kj + OS;
        float s = 0.0f, dp = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) { s += qs[c] * kj[c]; dp += g[c] * vj[c]; }
        const float ds = __builtin_amdgcn_exp2f(s - L) * (dp - Dq);
#pragma unroll
        for (int c = 0; c < 16; ++c) dq[c] += ds * kj[c];
    }
    if (valid) {
        float* dst = d_qkv + ((size_t)b * n + i) * RS + h * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c] = dq[c] * scale;
        lse2[((size_t)b * heads + h) * n + i] = L;
        dsum[((size_t)b * heads + h) * n + i] = Dq;
    }
}

__global__ __launch_bounds__(256) void tr_attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o, const float* __restrict__ lse2,
                                                             const float* __restrict__ dsum, float* __restrict__ d_qkv, int n, int heads, float scale) {
    const int b = (int)blockIdx.z, h = (int)blockIdx.y;
    const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const bool valid = j < n;
    const int jj = valid ? j : n - 1;
    const int RS = 3 * heads * 16, OS = heads * 16;
    const float* base = qkv + (size_t)b * n * RS;
    const float* krow = base + (size_t)jj * RS + OS + h * 16;
    float ks[16], v[16], dk[16], dv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        ks[c] = krow[c] * (scale * kLog2e);
        v[c] = krow[OS + c];
        dk[c] = 0.0f;
        dv[c] = 0.0f;
    }
    const float* qbase = base + h * 16;                             // query i at qbase + i * RS
    const float* gbase = d_o + (size_t)b * n * OS + h * 16;
    const float* Lr = lse2 + ((size_t)b * heads + h) * n;
    const float* Dr = dsum + ((size_t)b * heads + h) * n;
    for (int i = 0; i < n; ++i) {
        const float* qi = qbase + (size_t)i * RS;
        const float* gi = gbase + (size_t)i * OS;
        float s = 0.0f, dp = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) { s += qi[c] * ks[c]; dp += gi[c] * v[c]; }
        const float p = __builtin_amdgcn_exp2f(s - Lr[i]);
        const float ds = p * (dp - Dr[i]);
#pragma unroll
        for (int c = 0; c < 16; ++c) { dv[c] += p * gi[c]; dk[c] += ds * qi[c]; }
    }
    if (valid) {
        float* dst = d_qkv + ((size_t)b * n + j) * RS + OS + h * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) { dst[c] = dk[c] * scale; dst[OS + c] = dv[c]; }
    }
}

extern "C" int mvs_tr_attention_bwd(const float* qkv, const float* o, const float* d_o, float* d_qkv, float* lse_ws, float* dsum_ws, int B, int n,
                                    int heads, float softmax_scale, void* stream) {
    if (!qkv || !o || !d_o || !d_qkv || !lse_ws || !dsum_ws || B < 1 || n < 1 || heads < 1) { set_error("mvs_tr_attention_bwd: bad arguments"); return MVS_ERR_ARG; }
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)heads, (unsigned)B);
    hipLaunchKernelGGL(tr_attn_bwd_q_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, o, d_o, d_qkv, lse_ws, dsum_ws, n, heads, softmax_scale);
    int rc = check_launch("tr_attn_bwd_q_kernel");
    if (rc != MVS_OK) return rc;
    hipLaunchKernelGGL(tr_attn_bwd_kv_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, d_o, lse_ws, dsum_ws, d_qkv, n, heads, softmax_scale);
    return check_launch("tr_attn_bwd_kv_kernel");
}
