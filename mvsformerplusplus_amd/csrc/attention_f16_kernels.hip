// Flash attention of the stage-1 transformer regulariser in ONE fp16 term per operand (round 4).
//
// Reference: FlashAttnBlock -> flash_attn_qkvpacked_func / F.scaled_dot_product_attention
// (models/dino/layers/attention.py:76-101,141-170, module.py:535-583).  The reference's own GPU path runs this product with q, k, v
// and the probabilities in bf16 (flash-attn, fp32 accumulation).  Here the four operands are fp16 (11 significant bits instead of 8,
// the values are O(1..10) after the LayerNorms: measured on the oracle before the build, scripts/study_attention_precision.py:
// refined depth 8e-7 relative L1 from the fp32 oracle on plain inputs, 3.5e-5 on the x30-logits stress set; bf16 operands 7e-6 / 4e-4),
// scores, softmax statistics and both accumulations fp32.  The split-bf16 form of rounds 1-3 (tr_attention_kernel, 4 + 3 MFMAs and
// ~70 VALU instructions per 32 keys and query tile) stays as attention_precision "bf16x3".
//
// Shape: head_dim 16, n = 27 648 tokens at cfg2, 4 heads: per layer 3.06 G exponentials and 196 GFLOP.  With K = head_dim = 16 the
// score product of 16 keys x 16 queries is exactly one v_mfma_f32_16x16x16_f16, and its result layout (lane (j, g): keys 4g..4g+3 of
// query j) IS the B-operand layout of the p.v product when the 32 keys of a step are taken in the order
//     k-slot 8g + e  <->  key 16 (e >> 2) + 4g + (e & 3)            (e = 0..7)
// so P never moves between lanes: exp2 -> v_cvt_pk_f16_f32 -> v_mfma_f32_16x16x32_f16 with V^T pre-permuted to that order by the
// qkv projection's epilogue (tr_gemm_kernel<EPI_QKV16>).  Per 32 keys and 16 queries: 2 + 1 MFMAs, 8 v_exp_f32, 4 v_cvt_pk,
// 8 adds, one compare.
//
// Operand buffers (written by mvs_tr_qkv_fwd with operand format MVS_PREC_F16), npad = n rounded up to kAttnPad:
//   Q   [B, heads, npad, 16]                 fp16, pre-multiplied by softmax_scale * log2(e)
//   KP  [B, heads, npad/16, 4 (g), 16 (j), 4] fp16: dims 4g..4g+3 of key 16 tile + j     = the A operand of the score MFMA, lane-linear
//   VP  [B, heads, npad/32, 4 (g), 16 (d), 8] fp16: v[key(8g + e)][d]                    = the A operand of the p.v MFMA, lane-linear
// A block of KB keys is KB x 32 contiguous bytes of KP and of VP: the LDS image is a straight copy, made by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no ds_write), double-buffered, one barrier per block.  Lane-linear images make every
// ds_read_b64 / ds_read_b128 of a wave one contiguous 512 / 1024-byte run: conflict-free.
//
// Softmax bookkeeping: the accumulator input of the score MFMA is (kBias - m), so the MFMA result is already the exponent.  m starts
// as the row maximum over the first 32 keys; afterwards it is raised only when a step's probabilities sum to more than 2^(kBias + kLazy)
// (one compare + wave vote per step; the rare path rescales o and l).  kBias = 6 shifts the working range of the fp16 probabilities
// up: p <= 2^14 < 65504 in the common path, fp16 subnormals start 20 binades below the running maximum and flush 31 below
// (27 648 flushed keys together < 2e-5 of the row sum).
#include "mvs_common.h"
#include "attention_f16.h"

namespace mvs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float kBias = 6.0f;
constexpr float kLazy = 8.0f;
constexpr float kTrig = 16384.0f;            // 2^(kBias + kLazy)

__device__ __forceinline__ float max_over_groups(float v) {      // max over the four lanes (g = lane >> 4) that share a query
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

}  // namespace

// grid = (heads * npad / (64 QT), B): block -> head = block % heads (consecutive blocks go to consecutive XCDs, so with 4 heads every
// XCD's L2 holds the K / V of one head only: 1.8 MB at cfg2), query block = block / heads; wave w owns 16 QT queries.
template <int QT, int KB, bool LSUM_MFMA>
__global__ __launch_bounds__(256) void tr_attention_f16_kernel(const _Float16* __restrict__ Q, const _Float16* __restrict__ KP,
                                                               const _Float16* __restrict__ VP, float* __restrict__ out, int n, int npad,
                                                               int heads) {
    constexpr int BUF = KB * 64;                      // bytes of one buffer: K image (KB x 32) then V image (KB x 32)
    constexpr int PIECES = KB * 32 / 1024 / 4;        // 1-KiB DMA pieces per wave, block and operand
    static_assert(KB % 128 == 0, "a block is staged as 1-KiB pieces, 4 waves");
    __shared__ float4 lds4[2 * BUF / 16];
    char* lds = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int hh = (int)blockIdx.x % heads, qb = (int)blockIdx.x / heads, b = (int)blockIdx.y;
    const size_t hb = (size_t)b * heads + hh;
    const int q0 = (qb * 4 + wave) * 16 * QT;

    f16x4 qf[QT];                                      // B operand of the score product: dims 4g..4g+3 of query j
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) qf[qt] = *reinterpret_cast<const f16x4*>(Q + (hb * npad + q0 + 16 * qt + j) * 16 + 4 * g);

    const char* ksrc = reinterpret_cast<const char*>(KP + hb * npad * 16) + lane * 16;
    const char* vsrc = reinterpret_cast<const char*>(VP + hb * npad * 16) + lane * 16;
    auto stage = [&](int kb, int buf) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int piece = wave * PIECES + p;
            MVS_GLOBAL_LOAD_LDS16(ksrc + (size_t)kb * 32 + piece * 1024, lds + buf * BUF + piece * 1024);
            MVS_GLOBAL_LOAD_LDS16(vsrc + (size_t)kb * 32 + piece * 1024, lds + buf * BUF + KB * 32 + piece * 1024);
        }
    };
    const int koff = g * 128 + j * 8;                  // inside a 16-key tile image
    const int voff = lane * 16;                        // inside a 32-key step image

    stage(0, 0);
    MVS_WAIT_VMEM();
    __syncthreads();

    // running maximum from the first 32 keys (key 0 always exists).  cin = (kBias - m) in all four accumulator-input registers of a tile.
    float l[QT];
    f32x4 cin[QT], o[QT], ls[QT];
    {
        const f16x4 k0 = *reinterpret_cast<const f16x4*>(lds + koff);
        const f16x4 k1 = *reinterpret_cast<const f16x4*>(lds + 512 + koff);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 s0 = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
            s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(k0, qf[qt], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(k1, qf[qt], s1, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * g + r < n) mx = fmaxf(mx, s0[r]);
                if (16 + 4 * g + r < n) mx = fmaxf(mx, s1[r]);
            }
            const float c = kBias - max_over_groups(mx);
            cin[qt] = (f32x4){c, c, c, c};
            l[qt] = 0.0f;
            o[qt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            ls[qt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    f16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

    // One block of KB keys.  The score MFMAs of step st + 1 are issued BEFORE the exponentials of step st (software pipeline inside the
    // block): the matrix pipe works on the next scores while the VALU turns the current ones into probabilities.  The common path of a
    // step is one basic block; the rescale branch is taken when some query's probabilities sum to more than kTrig.
    auto scores = [&](const char* kl, int st, f32x4 (&s0)[QT], f32x4 (&s1)[QT]) {
        const f16x4 k0 = *reinterpret_cast<const f16x4*>(kl + (2 * st) * 512 + koff);
        const f16x4 k1 = *reinterpret_cast<const f16x4*>(kl + (2 * st + 1) * 512 + koff);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            s0[qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(k0, qf[qt], cin[qt], 0, 0, 0);      // keys 4g + r
            s1[qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(k1, qf[qt], cin[qt], 0, 0, 0);      // keys 16 + 4g + r
        }
    };
    auto block = [&](const int kb, const int buf, const bool tail) {
        const char* kl = lds + buf * BUF;
        const char* vl = kl + KB * 32;
        f32x4 s0[QT], s1[QT];
        scores(kl, 0, s0, s1);
#pragma unroll
        for (int st = 0; st < KB / 32; ++st) {
            const f16x8 vp = *reinterpret_cast<const f16x8*>(vl + st * 1024 + voff);
            float s[QT][8];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[qt][r] = s0[qt][r]; s[qt][4 + r] = s1[qt][r]; }
            }
            if (tail) {                                                    // padded keys (last block only; `tail` is a compile-time constant per call site)
                const int key0 = kb + 32 * st + 4 * g;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (key0 + r >= n) s[qt][r] = -INFINITY;
                        if (key0 + 16 + r >= n) s[qt][4 + r] = -INFINITY;
                    }
                }
            }
            if (st + 1 < KB / 32) scores(kl, st + 1, s0, s1);             // next step's scores: in flight under this step's exponentials
            float p[QT][8], ps[QT], psmax = 0.0f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                p[qt][0] = __builtin_amdgcn_exp2f(s[qt][0]);
                ps[qt] = p[qt][0];
#pragma unroll
                for (int e = 1; e < 8; ++e) { p[qt][e] = __builtin_amdgcn_exp2f(s[qt][e]); ps[qt] += p[qt][e]; }
                psmax = fmaxf(psmax, ps[qt]);                            // sums are >= 0; an overflowed sum is +inf (never NaN: inf - inf cannot occur)
            }
            if (__any(psmax > kTrig)) {
                // rare: some query's scores outgrew its running maximum by more than 2^kLazy - raise the maxima of the wave's tiles.  The
                // scores of step st + 1 (already computed against the old maximum) move with it.
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float* sq = s[qt];
                    const float mx = fmaxf(fmaxf(fmaxf(sq[0], sq[1]), fmaxf(sq[2], sq[3])), fmaxf(fmaxf(sq[4], sq[5]), fmaxf(sq[6], sq[7])));
                    const float d = max_over_groups(fmaxf(mx - kBias, 0.0f));
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    l[qt] *= alpha;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cin[qt][r] -= d; o[qt][r] *= alpha; ls[qt][r] *= alpha; s0[qt][r] -= d; s1[qt][r] -= d; }
                    ps[qt] = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { p[qt][e] = __builtin_amdgcn_exp2f(sq[e] - d); ps[qt] += p[qt][e]; }
                }
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f16x8 ph;
#pragma unroll
                for (int e = 0; e < 8; ++e) ph[e] = (_Float16)p[qt][e];
                if (LSUM_MFMA) ls[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, ls[qt], 0, 0, 0);   // row sums of the ROUNDED p on the matrix pipe
                else l[qt] += ps[qt];
                o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vp, ph, o[qt], 0, 0, 0);
            }
        }
    };

    int buf = 0;
    const int nfull = n / KB * KB;                                         // blocks without padded keys
    for (int kb = 0; kb < npad; kb += KB, buf ^= 1) {
        if (kb + KB < npad) stage(kb + KB, buf ^ 1);                       // lands while this block is computed
        if (kb < nfull) block(kb, buf, false);
        else if (kb < n) block(kb, buf, true);
        MVS_WAIT_VMEM();
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lt;
        if (LSUM_MFMA) {
            lt = ls[qt][0];                            // every row of the ones-product holds the query's whole sum
        } else {
            lt = l[qt];
            lt += __shfl_xor(lt, 16);
            lt += __shfl_xor(lt, 32);
        }
        const int tok = q0 + 16 * qt + j;
        if (tok < n) {
            const float inv = 1.0f / lt;
            *reinterpret_cast<float4*>(out + (((size_t)b * n + tok) * heads + hh) * 16 + 4 * g) =
                make_float4(o[qt][0] * inv, o[qt][1] * inv, o[qt][2] * inv, o[qt][3] * inv);
        }
    }
}

template <int QT, int KB, bool LM>
static int launch_one(const void* q, const void* k, const void* vp, float* out, int B, int n, int npad, int heads, hipStream_t st) {
    const dim3 grid(heads * (npad / (64 * QT)), B);
    hipLaunchKernelGGL((tr_attention_f16_kernel<QT, KB, LM>), grid, dim3(256), 0, st, static_cast<const _Float16*>(q),
                       static_cast<const _Float16*>(k), static_cast<const _Float16*>(vp), out, n, npad, heads);
    return check_launch("tr_attention_f16_kernel");
}

// variant: measurement switch (MVS_ATTN_VARIANT, scripts/prof_attn.py); 0 = the product's choice
int launch_attention_f16(const void* q, const void* k, const void* vp, float* out, int B, int n, int heads, int variant, hipStream_t st) {
    const int npad = (n + kAttnPad - 1) / kAttnPad * kAttnPad;
    switch (variant) {
        case 1: return launch_one<1, 128, false>(q, k, vp, out, B, n, npad, heads, st);
        case 2: return launch_one<2, 128, false>(q, k, vp, out, B, n, npad, heads, st);
        case 3: return launch_one<4, 128, false>(q, k, vp, out, B, n, npad, heads, st);
        case 4: return launch_one<2, 256, false>(q, k, vp, out, B, n, npad, heads, st);
        case 5: return launch_one<2, 128, true>(q, k, vp, out, B, n, npad, heads, st);
        case 6: return launch_one<4, 256, false>(q, k, vp, out, B, n, npad, heads, st);
        case 7: return launch_one<4, 128, true>(q, k, vp, out, B, n, npad, heads, st);
        default: return launch_one<2, 128, false>(q, k, vp, out, B, n, npad, heads, st);
    }
}

}  // namespace mvs
