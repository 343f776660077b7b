// Visibility CNN (reference cost_volume.py:36,93 + ConvBnReLU module.py:168-197) as ONE
// row-streaming launch:  entropy [N,H,W] -> sigmoid(conv1x1(CBR(16->8)(CBR(16->16)(CBR(1->16)(entropy))))) [N,H,W].
//
// Round 1 ran two launches with a 16-channel fp32 intermediate in HBM (641 + 650 MB per reference view at cfg2 for an
// operator whose algorithmic traffic is 8 bytes per pixel) and re-fetched the packed weights per tile.  Here a workgroup
// owns a vertical strip of 60 output columns x SH output rows and walks down it one row per iteration; the three layers
// run skewed by two rows each, so every iteration only reads rows produced by EARLIER iterations - one barrier per row:
//     iteration i:  A  layer-1 row i      (VALU, 1 -> 16, from a 3x3 entropy window held in registers)
//                   B  layer-2 row i-2    (MFMA 16 -> 16 from layer-1 rows i-3 .. i-1)
//                   C  layer-3 row i-4    (MFMA 16 -> 8, then 1x1 + sigmoid in the epilogue, from layer-2 rows i-5 .. i-3)
// Layer-1 / layer-2 activations live in two 4-row LDS rings as split bf16 (hi | lo), never in HBM; all packed weights of
// both MFMA layers sit in registers for the whole strip (80 VGPRs) and the 1 -> 16 weights in SGPRs.  Contraction =
// the 3-term split-bf16 product of the MFMA convolutions (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, ~2^-16 relative).
//
// LDS layout per layer: [octet o of 8 channels][ring row r][column c][hi x8 | lo x8], 32 B per position, octet planes offset by
// 16 B modulo the 256-byte bank row: the 16 ds_read_b128 lanes of a group (8 columns of octet 0 + 8 of octet 1) hit 16
// different 16-byte bank slots (the 8-byte epilogue stores are 4-way, but there are 4 of them against 20 reads per row).
// The ring slot of a row is row & 3; the row loop dispatches on i & 3 so that every slot offset is a compile-time constant.  Column c of the layer-1 ring is image column x0 - 2 + c, of the layer-2 ring
// x0 - 1 + c (c = 0..63); an MFMA tile is 16 consecutive columns of one row, wave w owns tile w.
// Zero padding follows the reference exactly: every layer's OUTPUT is forced to zero outside the image (the next
// Conv2d pads its input with zeros there), which is not the same as convolving zero-padded entropy.
#include "mvs_common.h"

namespace mvs {

typedef __bf16 vs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vs_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int VS_TW = 60;                 // output columns per strip
constexpr int VS_P = 64;                  // columns per ring row
constexpr int VS_RING = 4;                // ring rows
constexpr int VS_SH_MAX = 64;             // output rows per block (segment height) at most
constexpr int VS_EP = VS_P + 2;           // entropy tile pitch: image columns x0 - 3 .. x0 + 62
constexpr int VS_ENT_BYTES = (VS_SH_MAX + 6) * VS_EP * 4;
// F16 = false: split bf16 rings ([hi x8 | lo x8] = 32 B per position and octet, three MFMA terms); F16 = true (MVS_PREC_F16X2): fp16
// rings (16 B per position and octet), weights fp16 hi + lo, two MFMA terms - 20 instead of 30 MFMAs and 10 instead of 20 operand
// reads per row and wave
// fp16 rings: the second octet plane on the SAME 16-byte slots as the first (see bf_f16_plane_shift in conv_bf16x3_kernels.hip: the
// service groups of a ds_read_b128 pair lanes {0-3, 12-15} of one operand group with {20-27} of the other; rounds 3-4 shipped 128 and
// with it a 2-way conflict on every operand read - 40 % of the kernel's LDS cycles)
#ifndef VS_F16_PLANE_SHIFT
#ifdef MVS_F16_PLANE_SHIFT
#define VS_F16_PLANE_SHIFT MVS_F16_PLANE_SHIFT
#else
#define VS_F16_PLANE_SHIFT 0
#endif
#endif
template <bool F16>
struct VsL {
    static constexpr int POSB = F16 ? 16 : 32;                 // bytes per (position, octet)
    // one octet plane; the octet-1 lanes of a ds_read_b128 service group ({20-27} beside {0-3, 12-15}) take the free 16-byte slots of the
    // 256-byte bank row: 16 B away for the 32-byte positions (which cover the even slots), 0 B for the 16-byte positions (slots 4-11)
    static constexpr int PLANE = ((VS_RING * VS_P + 2) * POSB + 255) / 256 * 256 + (F16 ? VS_F16_PLANE_SHIFT : 16);
    static constexpr int LAYER = 2 * PLANE;
    static constexpr int LDS = 2 * LAYER + VS_ENT_BYTES;
};
typedef _Float16 vs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 vs_f16x4 __attribute__((ext_vector_type(4)));

#ifndef MVS_OPAQUE_VEC
#define MVS_OPAQUE_VEC "v"
#endif

// four fp32 -> four halves as TWO v_cvt_pk_f16_f32 (vector conversion: the element-wise form compiled to 4 v_cvt + 2 v_pack here, with
// the zero-padding selects on the four scalars; on the packed pairs they are two)
// zero padding on the packed result: two selects instead of four on the fp32 values
__device__ __forceinline__ vs_f16x4 vs_mask4(vs_f16x4 h, bool in) {
    typedef unsigned vs_u32x2 __attribute__((ext_vector_type(2)));
    vs_u32x2 u = __builtin_bit_cast(vs_u32x2, h);
    u[0] = in ? u[0] : 0u;
    u[1] = in ? u[1] : 0u;
    return __builtin_bit_cast(vs_f16x4, u);
}

__device__ __forceinline__ vs_f16x4 vs_cvt4(const float* v) {
    typedef float vs_f32x4 __attribute__((ext_vector_type(4)));
    return __builtin_convertvector((vs_f32x4){v[0], v[1], v[2], v[3]}, vs_f16x4);
}

__device__ __forceinline__ void vs_split4(const float* v, vs_bf16x4& hi, vs_bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
}

// B-operand offset of contraction step t: lane group g>>1 picks the first or second tap of the step (two taps x 16 channels =
// 32 k-values).  SLOT0 = ring slot of tap row kh = 0 (compile-time: the row loop dispatches on i & 3).
template <int SLOT0, int POSB>
__device__ __forceinline__ int vs_step_off(int laneoff, int tapsel, int t) {
    const int tapA = 2 * t, tapB = 2 * t + 1 < 9 ? 2 * t + 1 : 8;              // tap 9 does not exist: zero weights, any finite data
    const int offA = (((SLOT0 + tapA / 3) & (VS_RING - 1)) * VS_P + tapA % 3) * POSB;
    const int offB = (((SLOT0 + tapB / 3) & (VS_RING - 1)) * VS_P + tapB % 3) * POSB;
    return laneoff + (tapsel ? offB : offA);
}

struct VsOperand { vs_bf16x8 h, l; };
template <bool F16>
__device__ __forceinline__ VsOperand vs_load(const char* p) {
    VsOperand o;
    o.h = *reinterpret_cast<const vs_bf16x8*>(p);
    if constexpr (!F16) o.l = *reinterpret_cast<const vs_bf16x8*>(p + 16);
    else o.l = o.h;
    return o;
}

// one contraction step of one layer: the split-bf16 three-term product or the fp16 two-term product (weights hi + lo)
template <bool F16>
__device__ __forceinline__ f32x4 vs_mfma_lo(const vs_bf16x8& wl, const VsOperand& x, f32x4 a) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vs_f16x8, wl), __builtin_bit_cast(vs_f16x8, x.h), a, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, x.h, a, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 vs_mfma_hi(const vs_bf16x8& wh, const VsOperand& x, f32x4 a) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vs_f16x8, wh), __builtin_bit_cast(vs_f16x8, x.h), a, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, x.h, a, 0, 0, 0);
}

// one MFMA layer row: 5 contraction steps, three split-bf16 terms on three accumulators, the operand reads of step t+1 issued
// before the MFMAs of step t
// ONE (round 4, MVS_PREC_F16 / _F16MIX): the weights' lo term is not used - 10 instead of 20 MFMAs per row and wave (error study:
// scripts/study_weight_precision.py, "visibility CNN one term")
// `init`: the layer's bias, which rides in the first MFMA's accumulator input instead of four v_add_f32 in the epilogue
template <int SLOT0, bool F16, bool ONE = false>
__device__ __forceinline__ f32x4 vs_layer_row(const char* lds_layer, int laneoff, int tapsel, const vs_bf16x8* wh, const vs_bf16x8* wl, f32x4 init) {
    constexpr int POSB = VsL<F16>::POSB;
    f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0, a2 = init;
    VsOperand cur = vs_load<F16>(lds_layer + vs_step_off<SLOT0, POSB>(laneoff, tapsel, 0));
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        VsOperand nxt = cur;
        if (t < 4) nxt = vs_load<F16>(lds_layer + vs_step_off<SLOT0, POSB>(laneoff, tapsel, t + 1));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!ONE) a0 = vs_mfma_lo<F16>(wl[t], cur, a0);
        if constexpr (!F16) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t], cur.l, a1, 0, 0, 0);
        a2 = vs_mfma_hi<F16>(wh[t], cur, a2);
        cur = nxt;
    }
    return a0 + a1 + a2;
}

// both MFMA layers of one iteration interleaved (layer 2 reads ring 1, layer 3 reads ring 2: independent): four operand
// reads in flight under six MFMAs; one accumulator per layer - the two chains alternate, so a chain's next link is issued two
// MFMAs (32 cycles) after the previous one
template <int SLOT2, int SLOT3, bool F16, bool ONE = false>
__device__ __forceinline__ void vs_two_layer_rows(const char* lds1, const char* lds2, int laneoff, int tapsel, const vs_bf16x8* w2h,
                                                  const vs_bf16x8* w2l, const vs_bf16x8* w3h, const vs_bf16x8* w3l, f32x4& out2, f32x4& out3) {
    constexpr int POSB = VsL<F16>::POSB;
    f32x4 a = out2, c = out3;                                   // in: the layers' biases (accumulator init), out: the rows
    VsOperand cb = vs_load<F16>(lds1 + vs_step_off<SLOT2, POSB>(laneoff, tapsel, 0));
    VsOperand cc = vs_load<F16>(lds2 + vs_step_off<SLOT3, POSB>(laneoff, tapsel, 0));
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        VsOperand nb = cb, nc = cc;
        if (t < 4) {
            nb = vs_load<F16>(lds1 + vs_step_off<SLOT2, POSB>(laneoff, tapsel, t + 1));
            nc = vs_load<F16>(lds2 + vs_step_off<SLOT3, POSB>(laneoff, tapsel, t + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!ONE) {
            a = vs_mfma_lo<F16>(w2l[t], cb, a);
            c = vs_mfma_lo<F16>(w3l[t], cc, c);
        }
        if constexpr (!F16) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[t], cb.l, a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3h[t], cc.l, c, 0, 0, 0);
        }
        a = vs_mfma_hi<F16>(w2h[t], cb, a);
        c = vs_mfma_hi<F16>(w3h[t], cc, c);
        cb = nb;
        cc = nc;
    }
    out2 = a;
    out3 = c;
}

template <int V> struct VsInt { static constexpr int value = V; };

// grid = (strips, row segments, N)
template <bool F16, bool ONE = false>
__global__ __launch_bounds__(256) void vis_cnn_kernel(const float* __restrict__ ent, const float* __restrict__ w1 /*[9][16]*/,
                                                      const float* __restrict__ b1, const void* __restrict__ wp2, const float* __restrict__ b2,
                                                      const void* __restrict__ wp3, const float* __restrict__ b3, const float* __restrict__ w4,
                                                      const float* __restrict__ b4, float* __restrict__ vis, int H, int W, int SH) {
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* lds1 = reinterpret_cast<char*>(lds4);
    constexpr int VS_POSB = VsL<F16>::POSB, VS_PLANE = VsL<F16>::PLANE, VS_LAYER = VsL<F16>::LAYER;
    char* lds2 = lds1 + VS_LAYER;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int x0 = (int)blockIdx.x * VS_TW, r0 = (int)blockIdx.y * SH;
    const int r1 = r0 + SH < H ? r0 + SH : H;
    const float* e = ent + (size_t)blockIdx.z * H * W;
    float* vo = vis + (size_t)blockIdx.z * H * W;

    // ---- register-resident parameters ----
    vs_bf16x8 w2h[5], w2l[5], w3h[5], w3l[5];
    {
        const vs_bf16x8* q2 = reinterpret_cast<const vs_bf16x8*>(wp2) + lane;
        const vs_bf16x8* q3 = reinterpret_cast<const vs_bf16x8*>(wp3) + lane;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            w2h[t] = q2[(t * 2) * 64]; w2l[t] = q2[(t * 2 + 1) * 64];
            w3h[t] = q3[(t * 2) * 64]; w3l[t] = q3[(t * 2 + 1) * 64];
            // opaque: the compiler must keep them in registers instead of re-loading the invariant memory every row
#ifndef MVS_NO_OPAQUE_VEC
            if constexpr (ONE) asm volatile("" : "+" MVS_OPAQUE_VEC(w2h[t]), "+" MVS_OPAQUE_VEC(w3h[t]));
            else asm volatile("" : "+" MVS_OPAQUE_VEC(w2h[t]), "+" MVS_OPAQUE_VEC(w2l[t]), "+" MVS_OPAQUE_VEC(w3h[t]), "+" MVS_OPAQUE_VEC(w3l[t]));
#endif
        }
    }
    // Layer 1 (1 -> 16 channels, 3x3) in the one-term fp16 formats: ONE MFMA per 16 pixels.  K = 18 of 32: lane group g < 3 carries
    // kernel row kh = g as [e_hi(kw 0..2) | e_lo(kw 0..2) | 0 0] against the weights [w(kw 0..2) | w(kw 0..2) | 0 0] (one fp16 term, like the
    // two MFMA layers) - the entropy keeps its full precision as an fp16 hi + lo pair, the result layout (lane (pixel, g): channels
    // 4g .. 4g + 3) is the B-operand quad the ring stores, exactly like a layer-2 row.  6 VALU + 3 LDS reads + 1 MFMA per row and wave
    // instead of 18 v_pk_fma_f32 + 9 reads on four channels of all 64 columns.
    constexpr bool L1MFMA = F16 && ONE;
    vs_f16x8 w1a = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 b1q = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (L1MFMA) {
        const int kh = g < 3 ? g : 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const _Float16 wv = g < 3 ? (_Float16)w1[(kh * 3 + kw) * 16 + li] : (_Float16)0.0f;       // A operand: row = output channel li
            w1a[kw] = wv;
            w1a[3 + kw] = wv;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) b1q[r] = b1[4 * g + r];
    }
    float w1s[9][4], b1s[4];                                      // layer 1 on the VALU: channels 4*wave .. 4*wave+3 (wave-uniform -> SGPRs)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) w1s[t][r] = w1[t * 16 + 4 * wave + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1s[r] = b1[4 * wave + r];
    float b2v[4], b3v[4], w4v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b2v[r] = b2[4 * g + r];
        b3v[r] = g < 2 ? b3[4 * g + r] : 0.0f;
        w4v[r] = g < 2 ? w4[4 * g + r] : 0.0f;
    }
    const float bias4 = b4[0];

    // ---- stage A input: the strip's entropy rows r0-3 .. r1+2 staged once in LDS (zero outside the image = conv1's padding);
    //      no global load is left inside the row loop ----
    const int xa = x0 - 2 + lane;                                  // image column of layer-1 column `lane`
    const int i0 = r0 - 2, i1 = r1 + 4;                            // iterations i0 .. i1-1 (layer-1 rows i0 .. r1+1 are needed)
    float* ent_s = reinterpret_cast<float*>(lds1 + 2 * VS_LAYER);
    {
        const int nrow = r1 - r0 + 6;
        for (int idx = tid; idx < nrow * VS_EP; idx += 256) {
            const int rr = idx / VS_EP, cc = idx - rr * VS_EP;
            const int yy = r0 - 3 + rr, xx = x0 - 3 + cc;
            ent_s[idx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? e[(size_t)yy * W + xx] : 0.0f;
        }
    }
    __syncthreads();
    const int colmaskA = xa >= 0 && xa < W;
    // per-lane LDS offsets
    const int wrA = (wave >> 1) * VS_PLANE + lane * VS_POSB + (wave & 1) * 8;                     // stage A store (octet wave>>1, quad wave&1: 8 bytes of hi [, 8 of lo at + 16])
    const int laneoff = (g & 1) * VS_PLANE + (16 * wave + li) * VS_POSB;                          // MFMA B-operand reads
    const int wrB = (g >> 1) * VS_PLANE + (16 * wave + li) * VS_POSB + (g & 1) * 8;               // stage B store
    const int tapsel = g >> 1;
    const int c2 = 16 * wave + li;                                 // layer-2 / layer-3 column of this lane
    const int xb = x0 - 1 + c2, xc = x0 + c2;

    // steady: every stage has a row to produce (r0 + 4 <= i <= r1 + 1) - no range logic, no branches
    auto iteration = [&](auto phase, auto steady, int i) __attribute__((always_inline)) {      // eight call sites: without the attribute the body is
                                                                                                 // outlined, its captures go through scratch and generic pointers
        constexpr int PH = decltype(phase)::value;                  // == i & 3: ring slot of row i + k is (PH + k) & 3
        constexpr bool STEADY = decltype(steady)::value != 0;
        // ---- A: layer-1 row i ----
        if constexpr (L1MFMA) {
            if (STEADY || i <= r1 + 1) {
                // entropy row i - 1 + g, columns (x0 - 2 + c2) - 1 .. + 1 of this lane's pixel; group 3 re-reads group 2's row against zero weights
                const float* ep = ent_s + (i - 1 + (g < 3 ? g : 2) - (r0 - 3)) * VS_EP + c2;
                const float e0 = ep[0], e1 = ep[1], e2 = ep[2];
                typedef float vs_f32x2 __attribute__((ext_vector_type(2)));
                typedef _Float16 vs_f16x2 __attribute__((ext_vector_type(2)));
                typedef unsigned vs_u32x4 __attribute__((ext_vector_type(4)));
                const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((vs_f32x2){e0, e1}, vs_f16x2));     // [hi0, hi1]
                const float l0 = MVS_FMA_MIX_LO(p0, -1.0f, e0), l1 = MVS_FMA_MIX_HI(p0, -1.0f, e1);                           // e - (float)hi
                const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((vs_f32x2){e2, l0}, vs_f16x2));     // [hi2, lo0]
                const float l2 = MVS_FMA_MIX_LO(p1, -1.0f, e2);
                const unsigned p2 = __builtin_bit_cast(unsigned, __builtin_convertvector((vs_f32x2){l1, l2}, vs_f16x2));     // [lo1, lo2]
                const vs_f16x8 bop = __builtin_bit_cast(vs_f16x8, (vs_u32x4){p0, p1, p2, 0u});
                const f32x4 r1v = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1a, bop, b1q, 0, 0, 0);
                const int xa1 = x0 - 2 + c2;
                const bool in = xa1 >= 0 && xa1 < W && i >= 0 && i < H;
                float a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = fmaxf(r1v[r], 0.0f);
                *reinterpret_cast<vs_f16x4*>(lds1 + wrB + PH * (VS_P * VS_POSB)) = vs_mask4(vs_cvt4(a), in);
            }
        } else if (STEADY || i <= r1 + 1) {
            float er[3][3];                                        // entropy rows i-1 .. i+1, columns xa-1 .. xa+1
            const float* ep = ent_s + (i - 1 - (r0 - 3)) * VS_EP + lane;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) er[kh][kw] = ep[kh * VS_EP + kw];
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 a01 = {b1s[0], b1s[1]}, a23 = {b1s[2], b1s[3]};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const f32x2 ev = {er[kh][kw], er[kh][kw]};
                    a01 = __builtin_elementwise_fma(ev, (f32x2){w1s[kh * 3 + kw][0], w1s[kh * 3 + kw][1]}, a01);
                    a23 = __builtin_elementwise_fma(ev, (f32x2){w1s[kh * 3 + kw][2], w1s[kh * 3 + kw][3]}, a23);
                }
            float a[4] = {a01[0], a01[1], a23[0], a23[1]};
            const bool in = colmaskA && i >= 0 && i < H;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = F16 ? fmaxf(a[r], 0.0f) : (in ? fmaxf(a[r], 0.0f) : 0.0f);   // F16: the zero padding is applied to the packed halves
            char* p = lds1 + wrA + PH * (VS_P * VS_POSB);
            if constexpr (F16) {
                *reinterpret_cast<vs_f16x4*>(p) = vs_mask4(vs_cvt4(a), in);
            } else {
                vs_bf16x4 hi, lo;
                vs_split4(a, hi, lo);
                *reinterpret_cast<vs_bf16x4*>(p) = hi;
                *reinterpret_cast<vs_bf16x4*>(p + 16) = lo;
            }
        }
        // ---- B: layer-2 row i-2 from layer-1 rows i-3 .. i-1;  C: layer-3 row i-4 from layer-2 rows i-5 .. i-3 ----
        const int yb = i - 2, yc = i - 4;
        const bool doB = STEADY || (yb >= r0 - 1 && yb <= r1), doC = STEADY || yc >= r0;
        constexpr int S2 = (PH + 1) & 3, S3 = (PH + 3) & 3;          // slots of rows i-3 and i-5
        f32x4 acc2 = {b2v[0], b2v[1], b2v[2], b2v[3]}, acc3 = {b3v[0], b3v[1], b3v[2], b3v[3]};
        if (STEADY || (doB && doC)) vs_two_layer_rows<S2, S3, F16, ONE>(lds1, lds2, laneoff, tapsel, w2h, w2l, w3h, w3l, acc2, acc3);
        else if (doB) acc2 = vs_layer_row<S2, F16, ONE>(lds1, laneoff, tapsel, w2h, w2l, acc2);
        else if (doC) acc3 = vs_layer_row<S3, F16, ONE>(lds2, laneoff, tapsel, w3h, w3l, acc3);
        if (doB) {
            const bool in = xb >= 0 && xb < W && yb >= 0 && yb < H;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = F16 ? fmaxf(acc2[r], 0.0f) : (in ? fmaxf(acc2[r], 0.0f) : 0.0f);
            char* p = lds2 + wrB + ((PH + 2) & 3) * (VS_P * VS_POSB);          // row i-2
            if constexpr (F16) {
                *reinterpret_cast<vs_f16x4*>(p) = vs_mask4(vs_cvt4(v), in);
            } else {
                vs_bf16x4 hi, lo;
                vs_split4(v, hi, lo);
                *reinterpret_cast<vs_bf16x4*>(p) = hi;
                *reinterpret_cast<vs_bf16x4*>(p + 16) = lo;
            }
        }
        if (doC) {                                                  // 1x1 + sigmoid
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) part += fmaxf(acc3[r], 0.0f) * w4v[r];                   // rows 8..15 of the tile: zero weights
            part += __shfl_xor(part, 16);
            if (g == 0 && c2 < VS_TW && xc < W && yc < r1) {
                const float z = part + bias4;
                vo[(size_t)yc * W + xc] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z * 1.4426950408889634f));
            }
        }
        __syncthreads();
    };
    auto general = [&](int i) __attribute__((always_inline)) {
        switch (i & 3) {
            case 0: iteration(VsInt<0>(), VsInt<0>(), i); break;
            case 1: iteration(VsInt<1>(), VsInt<0>(), i); break;
            case 2: iteration(VsInt<2>(), VsInt<0>(), i); break;
            default: iteration(VsInt<3>(), VsInt<0>(), i); break;
        }
    };
    int i = i0;
    for (; i < i1 && (i < r0 + 4 || (i & 3) != 0); ++i) general(i);          // warm-up rows, then up to the next multiple of four
    for (; i + 3 <= r1 + 1; i += 4) {                                         // steady state, four rows (= the ring) per trip
        iteration(VsInt<0>(), VsInt<1>(), i);
        iteration(VsInt<1>(), VsInt<1>(), i + 1);
        iteration(VsInt<2>(), VsInt<1>(), i + 2);
        iteration(VsInt<3>(), VsInt<1>(), i + 3);
    }
    for (; i < i1; ++i) general(i);                                           // drain
}

// (Round 6 measured a wave-autonomous form of the one-term fp16 CNN - a wave owns a whole 60-column strip: four MFMA tiles per row, private
// LDS rings, the entropy window straight from global memory, no workgroup barrier; bit-identical on the emulator - at 233 VGPRs / 70 KB of
// LDS it ran two waves per SIMD and 15 % SLOWER than the block form above (101 vs 88 us per stage-3/4 launch, profiles/r06_vis_wave_ab.txt):
// the kernel's cost is the SUM of its pipes' work - 40 us of MFMA issue, 53 us of LDS operand traffic (one ds_read_b128 per MFMA at 16 output
// channels), 21 us of VALU per launch - not its barriers.  The kernel is in git history, commit "visibility CNN: wave-autonomous kernel".)

template <bool F16, bool ONE = false>
static int vis_weight_stream_t(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                               const float* b3, const float* w4, const float* b4, float* vis, int N, int H, int W, hipStream_t st) {
    constexpr int VS_LDS = VsL<F16>::LDS;
    const int strips = (int)ceil_div(W, VS_TW);
    // segment height: a block's run time is ~ (SH + 6) row iterations (6 warm-up rows) and the launch takes ceil(blocks / resident
    // blocks) of those back to back - pick the segment count that minimises that product (blocks per CU: 3 with the split-bf16 rings'
    // 52 KB of LDS, 4 with the fp16 rings' 36 KB and 114 VGPRs)
    const long long per_seg = (long long)strips * N, resident = (160 * 1024 / VS_LDS < 4 ? 160 * 1024 / VS_LDS : 4) * 256;
    int segs = (int)ceil_div(H, VS_SH_MAX), SH = (int)ceil_div(H, segs);
    long long best = -1;
    for (int sg = (int)ceil_div(H, VS_SH_MAX); sg <= (H + 7) / 8; ++sg) {
        const int sh = (int)ceil_div(H, sg);
        const long long rounds = (per_seg * ceil_div(H, sh) + resident - 1) / resident;
        const long long cost = rounds * (sh + 6);
        if (best < 0 || cost < best) { best = cost; segs = (int)ceil_div(H, sh); SH = sh; }
    }
    if (VS_LDS > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vis_cnn_kernel<F16, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, VS_LDS);
    hipLaunchKernelGGL((vis_cnn_kernel<F16, ONE>), dim3(strips, segs, N), dim3(256), VS_LDS, st, entropy, w1, b1, w2, b2, w3, b3, w4, b4, vis, H, W, SH);
    return check_launch("vis_cnn_kernel");
}

// f16: 1 = the fp16 two-term form (MVS_PREC_F16X2: rings in fp16, w2 / w3 packed as fp16 hi + lo), 2 = the same with ONE weight term
// (MVS_PREC_F16 / MVS_PREC_F16MIX)
int vis_weight_stream_bf16x3(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                             const float* b3, const float* w4, const float* b4, float* vis, int N, int H, int W, hipStream_t st, int f16) {
    if (f16 == 2) return vis_weight_stream_t<true, true>(entropy, w1, b1, w2, b2, w3, b3, w4, b4, vis, N, H, W, st);
    return f16 ? vis_weight_stream_t<true>(entropy, w1, b1, w2, b2, w3, b3, w4, b4, vis, N, H, W, st)
               : vis_weight_stream_t<false>(entropy, w1, b1, w2, b2, w3, b3, w4, b4, vis, N, H, W, st);
}

}  // namespace mvs
