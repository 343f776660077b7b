// Flash attention of the stage-1 transformer regulariser with ONE 16-bit term per operand (round 4): MVS_PREC_ATTN16.
//
// Reference: FlashAttnBlock -> flash_attn_qkvpacked_func / F.scaled_dot_product_attention
// (models/dino/layers/attention.py:76-101,141-170, module.py:535-583).  The reference's own GPU path runs this product with q, k, v
// and the probabilities in bf16 (flash-attn, fp32 accumulation).  Here q and k are fp16 (11 significant bits instead of 8: the score
// error enters the exponent, it is the sensitive side), the probabilities and v are bf16 like the reference's (fp32's exponent range: a
// probability can never overflow, which is what lets the inner loop run without any running-maximum check, below); scores, row sums
// and both accumulations fp32.  Measured on the oracle before the build (scripts/study_attention_precision.py): refined depth 7e-6 to
// 9e-6 relative L1 from the fp32 oracle on plain inputs, 2.5e-4 to 2.8e-4 on the x30-logits stress set (all-bf16 like the reference:
// 7e-6 to 1.3e-5 / 2.8e-4 to 4.2e-4; bar 1e-3).  The split-bf16 form of rounds 1-3 (tr_attention_kernel: 4 + 3 MFMAs and ~70 VALU
// instructions per 32 keys and query tile) stays as attention_precision "bf16x3".
//
// Shape: head_dim 16, n = 27 648 tokens at cfg2, 4 heads: per layer 3.06 G exponentials and 196 GFLOP.  Round-4 measurements that shape
// the kernel (scripts/ubench/pipes2.hip, profiles/r04_pipes2.txt; cycles per wave-instruction and SIMD at 4 waves per SIMD): v_exp_f32 7.4,
// plain VALU 2.8, v_cvt_pk 4, ds_read_b64 = ds_read_b128 = 15 (LDS pipe, overlaps everything), v_mfma 16x16x16 = 16x16x32 = 16; an MFMA
// hides about two plain VALU instructions, beyond that VALU time adds.  The first build of this kernel (fp16 probabilities, a rescale
// check per step) ran 0.73 ms per layer, VALU pipe 75 % busy, matrix pipe 17 %; ablations: no check 0.49 ms, no exponentials 0.56,
// neither 0.33, no MFMAs at all 0.70 (profiles/r04_attn_ab.txt).  So the loop is built around the VALU count:
//   * per 32 keys and 16 queries the VALU issues 8 v_exp_f32 + 4 v_cvt_pk_bf16_f32 and nothing else; row sums are taken by the matrix
//     pipe (one more MFMA against an all-ones operand: the sum of the ROUNDED probabilities, the normaliser consistent with p.v);
//   * with K = head_dim = 16 the score product of 16 keys x 16 queries is one v_mfma_f32_16x16x16_f16 whose result layout (lane (j, g):
//     keys 4g..4g+3 of query j) IS the B-operand layout of the p.v product when the 32 keys of a step are taken in the order
//         k-slot 8g + e  <->  key 16 (e >> 2) + 4g + (e & 3)            (e = 0..7)
//     so P never moves between lanes: exp2 -> cvt -> v_mfma_f32_16x16x32_bf16 with V^T pre-permuted to that order by the qkv
//     projection's epilogue (tr_gemm_kernel<EPI_QKV16>); "- m" rides in the score MFMA's accumulator input;
//   * the running maximum m is the row maximum over the first 32 keys and is raised at BLOCK boundaries only (KB keys), from the
//     growth of the row sum over the block (one subtract + compare per block): bf16 probabilities keep their 8 bits at any magnitude,
//     so a stale m costs nothing until fp32 itself overflows (a score 120 binades above the running maximum inside one block).  That
//     case - and any other non-finite result - is caught at the end of the work-group, which then redoes its queries with the
//     classic per-step online softmax (SAFE path: correct for any input, never taken on sane data).
//
// Operand buffers (written by mvs_tr_qkv_fwd with operand format MVS_PREC_ATTN16), npad = n rounded up to kAttnPad:
//   Q   [B, heads, npad, 16]                     fp16, pre-multiplied by softmax_scale * log2(e)
//   KP  [B, heads, npad/32, 4 (g), 16 (j), 2, 4]  fp16: dims 4g..4g+3 of keys 32 step + j and 32 step + 16 + j  (A operands of a step's two
//                                                 score MFMAs in ONE 16-byte read per lane)
//   VP  [B, heads, npad/32, 4 (g), 16 (d), 8]     bf16: v[key(8g + e)][d]                                      (A operand of the p.v MFMA)
// Both are lane-linear: a wave's read is one contiguous 1-KiB run (conflict-free), and a block of KB keys is KB x 32 contiguous bytes of
// each, so the LDS image is a straight copy made by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), double-buffered, one
// barrier per block.
#include "mvs_common.h"
#include "attention_f16.h"

namespace mvs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr float kRaise = 1048576.0f;         // 2^20: a block whose probabilities sum to more than this raises the running maximum

// fp32 with all exponent bits set = inf or NaN.  The file is compiled with -fno-honor-nans (the softmax never produces one on the fast
// path's sane inputs), so no floating-point comparison can be trusted to see a NaN - the compiler even rewrites the bit test below into
// `fabs(v) == inf` (round 4: the SAFE path was never entered on the GPU) unless the bits are laundered through an opaque register first.
__device__ __forceinline__ bool non_finite(float v) {
    union { float f; unsigned u; } c;
    c.f = v;
    unsigned u = c.u;
    asm volatile("" : "+" MVS_OPAQUE_REG(u));
    return (u & 0x7f800000u) == 0x7f800000u;
}

// two probabilities -> one register of two bf16 by TRUNCATION (v_perm_b32: a plain VALU instruction, v_cvt_pk_bf16_f32 costs 1.4x as much).
// Truncation biases every probability low by 2^-9 on average; the row sum is taken from the SAME truncated values (ones-MFMA), so the
// bias cancels in o / l and what is left is the rounding noise of one bf16 ulp, as with round-to-nearest.
__device__ __forceinline__ unsigned pack_bf16_trunc(float lo, float hi) {
    union { float f; unsigned u; } a, b;
    a.f = hi;
    b.f = lo;
    return __builtin_amdgcn_perm(a.u, b.u, 0x07060302u);
}

__device__ __forceinline__ float max_over_groups(float v) {      // max over the four lanes (g = lane >> 4) that share a query
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

}  // namespace

// grid = (heads * npad / (64 QT), B): block -> head = block % heads (consecutive blocks go to consecutive XCDs, so with 4 heads every
// XCD's L2 holds the K / V of one head only: 1.8 MB at cfg2), query block = block / heads; wave w owns 16 QT queries.
// ABL: measurement-only ablation mask (scripts/prof_attn.py; results are wrong): 2 no exponentials, 4 no staging after the first block,
// 8 no p.v / row-sum MFMAs, 16 no score MFMAs, 32 no barrier in the loop, 64 row sums by VALU adds instead of the ones-MFMA.
template <int QT, int KB, int ABL = 0>
__global__ __launch_bounds__(256) void tr_attention16_kernel(const _Float16* __restrict__ Q, const _Float16* __restrict__ KP,
                                                             const __bf16* __restrict__ VP, float* __restrict__ out, int n, int npad,
                                                             int heads) {
    constexpr int BUF = KB * 64;                      // bytes of one buffer: K image (KB x 32) then V image (KB x 32)
    constexpr int PIECES = KB * 32 / 1024 / 4;        // 1-KiB DMA pieces per wave, block and operand
    static_assert(KB % 128 == 0, "a block is staged as 1-KiB pieces, 4 waves");
    __shared__ float4 lds4[2 * BUF / 16];
    __shared__ int redo;
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
    const int loff = lane * 16;                        // inside a 32-key step image (K and V alike)

    if (tid == 0) redo = 0;
    stage(0, 0);
    MVS_WAIT_VMEM();
    __syncthreads();

    // running maximum from the first 32 keys (key 0 always exists).  cin = -m in all four accumulator-input registers of a tile.
    f32x4 cin[QT], o[QT], ls[QT];
    float lprev[QT];
    auto init_state = [&]() {
        const f16x8 kk = *reinterpret_cast<const f16x8*>(lds + loff);
        const f16x4 k0 = __builtin_shufflevector(kk, kk, 0, 1, 2, 3), k1 = __builtin_shufflevector(kk, kk, 4, 5, 6, 7);
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
            const float c = -max_over_groups(mx);
            cin[qt] = (f32x4){c, c, c, c};
            lprev[qt] = 0.0f;
            o[qt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            ls[qt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    init_state();
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // ---- one block of KB keys, fast path.  The score MFMAs of step st + 1 are issued BEFORE the exponentials of step st (software
    // pipeline inside the block); a step is branch-free: [2 LDS reads] [2 QT score MFMAs] [8 QT exp, 4 QT cvt] [2 QT MFMAs: p.v, row sum]
    auto scores = [&](const char* kl, int st, f32x4 (&s0)[QT], f32x4 (&s1)[QT]) {
        const f16x8 kk = *reinterpret_cast<const f16x8*>(kl + st * 1024 + loff);
        const f16x4 k0 = __builtin_shufflevector(kk, kk, 0, 1, 2, 3), k1 = __builtin_shufflevector(kk, kk, 4, 5, 6, 7);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (ABL & 16) {
                s0[qt] = cin[qt] + (f32x4){(float)k0[0], (float)k0[1], (float)k0[2], (float)k0[3]};
                s1[qt] = cin[qt] + (f32x4){(float)k1[0], (float)k1[1], (float)k1[2], (float)k1[3]};
                continue;
            }
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
            const bf16x8 vp = *reinterpret_cast<const bf16x8*>(vl + st * 1024 + loff);
            float s[QT][8];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[qt][r] = s0[qt][r]; s[qt][4 + r] = s1[qt][r]; }
            }
            if (tail) {                                                    // padded keys (last block only; a compile-time constant per call site)
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
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = (ABL & 2) ? s[qt][e] : __builtin_amdgcn_exp2f(s[qt][e]);
                bf16x8 ph;
                if (ABL & 64) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ph[e] = (__bf16)p[e];   // round to nearest: the VALU row sums are those of the unrounded p
                } else {
                    union { unsigned u[4]; bf16x8 v; } pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk.u[e] = pack_bf16_trunc(p[2 * e], p[2 * e + 1]);
                    ph = pk.v;
                }
                if (ABL & 8) { o[qt][0] += (float)ph[0] + (float)ph[2] + (float)ph[4] + (float)ph[6] + (float)vp[qt]; continue; }
                if (ABL & 64) ls[qt][0] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                else ls[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ph, ls[qt], 0, 0, 0);    // row sums of the rounded p: every row = the query's sum
                o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vp, ph, o[qt], 0, 0, 0);
            }
        }
        // block boundary: raise the running maximum of queries whose row sum grew by more than kRaise over the block (over-estimates the
        // needed raise by at most log2(KB): harmless for bf16 probabilities).  ls[.][0] is the same in the four lanes of a query.
        if (!(ABL & 64)) {
            float grow = 0.0f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) grow = fmaxf(grow, ls[qt][0] - lprev[qt]);
            if (__builtin_expect(__any(grow > kRaise), 0)) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float d = floorf(__builtin_amdgcn_logf(fmaxf(ls[qt][0] - lprev[qt], 1.0f)));      // >= 0; log2
                    const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cin[qt][r] -= d; o[qt][r] *= alpha; ls[qt][r] *= alpha; }
                }
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) lprev[qt] = ls[qt][0];
        }
    };

    // ---- SAFE path: the classic online softmax, the maximum checked at every step.  Only taken when the fast path produced a non-finite
    // result (fp32 overflow of a probability: a score > 120 binades above the running maximum inside one block).
    auto block_safe = [&](const int kb, const int buf) {
        const char* kl = lds + buf * BUF;
        const char* vl = kl + KB * 32;
        for (int st = 0; st < KB / 32; ++st) {
            const f16x8 kk = *reinterpret_cast<const f16x8*>(kl + st * 1024 + loff);
            const f16x4 k0 = __builtin_shufflevector(kk, kk, 0, 1, 2, 3), k1 = __builtin_shufflevector(kk, kk, 4, 5, 6, 7);
            const bf16x8 vp = *reinterpret_cast<const bf16x8*>(vl + st * 1024 + loff);
            const int key0 = kb + 32 * st + 4 * g;
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(k0, qf[qt], cin[qt], 0, 0, 0);
                f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(k1, qf[qt], cin[qt], 0, 0, 0);
                float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + r >= n) s[r] = -INFINITY;
                    if (key0 + 16 + r >= n) s[4 + r] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, s[e]);
                const float d = max_over_groups(fmaxf(mx, 0.0f));          // scores are relative to the running maximum: raise it by d
                const float alpha = __builtin_amdgcn_exp2f(-d);
                bf16x8 ph;
#pragma unroll
                for (int e = 0; e < 8; ++e) ph[e] = (__bf16)__builtin_amdgcn_exp2f(s[e] - d);
#pragma unroll
                for (int r = 0; r < 4; ++r) { cin[qt][r] -= d; o[qt][r] *= alpha; ls[qt][r] *= alpha; }
                ls[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ph, ls[qt], 0, 0, 0);
                o[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vp, ph, o[qt], 0, 0, 0);
            }
        }
    };

    const int nfull = n / KB * KB;                                         // blocks without padded keys
    int buf = 0;
    for (int kb = 0; kb < npad; kb += KB, buf ^= 1) {
        if (!(ABL & 4) && kb + KB < npad) stage(kb + KB, buf ^ 1);         // lands while this block is computed
        if (kb < nfull) block(kb, buf, false);
        else if (kb < n) block(kb, buf, true);
        if (!(ABL & 4)) MVS_WAIT_VMEM();
        if (!(ABL & 32)) __syncthreads();
    }
    if (ABL & 64) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { float t = ls[qt][0]; t += __shfl_xor(t, 16); t += __shfl_xor(t, 32); ls[qt][0] = t; }
    }

    // non-finite anywhere in the work-group -> redo its queries on the SAFE path (work-group uniform: the staging is cooperative)
    {
        bool bad = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float t = ls[qt][0] + ((o[qt][0] + o[qt][1]) + (o[qt][2] + o[qt][3]));
            bad = bad || non_finite(t) || non_finite(ls[qt][0]) || ls[qt][0] <= 0.0f;
        }
        if (bad) redo = 1;
        __syncthreads();
        if (__builtin_expect(redo != 0, 0)) {
            stage(0, 0);
            MVS_WAIT_VMEM();
            __syncthreads();
            init_state();
            buf = 0;
            for (int kb = 0; kb < npad; kb += KB, buf ^= 1) {
                if (kb + KB < npad) stage(kb + KB, buf ^ 1);
                if (kb < n) block_safe(kb, buf);
                MVS_WAIT_VMEM();
                __syncthreads();
            }
        }
    }

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int tok = q0 + 16 * qt + j;
        if (tok < n) {
            const float inv = 1.0f / ls[qt][0];
            *reinterpret_cast<float4*>(out + (((size_t)b * n + tok) * heads + hh) * 16 + 4 * g) =
                make_float4(o[qt][0] * inv, o[qt][1] * inv, o[qt][2] * inv, o[qt][3] * inv);
        }
    }
}

template <int QT, int KB, int ABL = 0>
static int launch_one(const void* q, const void* k, const void* vp, float* out, int B, int n, int npad, int heads, hipStream_t st) {
    const dim3 grid(heads * (npad / (64 * QT)), B);
    hipLaunchKernelGGL((tr_attention16_kernel<QT, KB, ABL>), grid, dim3(256), 0, st, static_cast<const _Float16*>(q),
                       static_cast<const _Float16*>(k), static_cast<const __bf16*>(vp), out, n, npad, heads);
    return check_launch("tr_attention16_kernel");
}

// variant: measurement switch (MVS_ATTN_VARIANT, scripts/prof_attn.py); 0 = the product's choice.  1 .. 6 are tile
// shapes of the SAME algorithm (every one numerically valid).  The ablations (exponentials / p.v MFMAs / barriers removed: timing only,
// wrong results) exist only in a library built with -DMVS_ATTN_ABLATIONS; the shipped one ignores their numbers.
int launch_attention16(const void* q, const void* k, const void* vp, float* out, int B, int n, int heads, int variant, hipStream_t st) {
    const int npad = (n + kAttnPad - 1) / kAttnPad * kAttnPad;
    switch (variant) {
        case 1: return launch_one<1, 128>(q, k, vp, out, B, n, npad, heads, st);
        case 2: return launch_one<2, 128>(q, k, vp, out, B, n, npad, heads, st);
        case 3: return launch_one<4, 128>(q, k, vp, out, B, n, npad, heads, st);
        case 4: return launch_one<2, 256>(q, k, vp, out, B, n, npad, heads, st);
        case 5: return launch_one<1, 256>(q, k, vp, out, B, n, npad, heads, st);
        case 6: return launch_one<4, 256>(q, k, vp, out, B, n, npad, heads, st);
#ifdef MVS_ATTN_ABLATIONS
        // ablations of the QT = 2, KB = 128 shape (16 + mask): timing only
        case 18: return launch_one<2, 128, 2>(q, k, vp, out, B, n, npad, heads, st);
        case 20: return launch_one<2, 128, 4>(q, k, vp, out, B, n, npad, heads, st);
        case 52: return launch_one<2, 128, 36>(q, k, vp, out, B, n, npad, heads, st);
        case 24: return launch_one<2, 128, 8>(q, k, vp, out, B, n, npad, heads, st);
        case 32: return launch_one<2, 128, 16>(q, k, vp, out, B, n, npad, heads, st);
        case 40: return launch_one<2, 128, 24>(q, k, vp, out, B, n, npad, heads, st);
        case 80: return launch_one<2, 128, 64>(q, k, vp, out, B, n, npad, heads, st);
#endif
        default: return launch_one<1, 256>(q, k, vp, out, B, n, npad, heads, st);
    }
}

}  // namespace mvs
