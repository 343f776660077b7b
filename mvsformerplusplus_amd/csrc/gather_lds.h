// LDS-staged homography warp + group-wise correlation, round-2 form of the two
// gather passes.  Reference behaviour restated (never copied): models/warping.py:84-106 (projection, bilinear
// grid_sample with zeros padding / align_corners=True), models/cost_volume.py:74-101 (group correlation, softmax
// entropy, visibility-weighted aggregation).
//
// Why a second form: the direct kernels in warp_kernels.hip issue 2*C eight-byte loads per (pixel, plane, view) and are
// bound by the CU's vector-memory return path (~30 B/clk/CU for jittered addresses).  Here a
// workgroup owns a TILE of reference pixels (4 rows x 64 / 32 / 16 columns) and, per source view and group of depth
// planes ("unit"),
//   1. computes the bilinear tap set of every (pixel, plane) once and keeps it in registers,
//   2. reduces the exact bounding box of all taps of the tile (packed u16 min / max, wave shuffles + one LDS hop),
//   3. stages that source WINDOW - 8 channels at a time - from the planar NCHW feature map into LDS with coalesced
//      row loads (16 B-aligned segments of consecutive dwords: the 16-lanes-per-clock path of the texture addresser),
//      transposed to channel-interleaved 16-byte quads  win[quad][position],
//   4. gathers the four taps with ds_read_b128 (two quads per tap: 8 channels in two reads, 256 B/clk/CU) and
//      reduces the channel groups in registers.
// A unit whose window exceeds the LDS capacity (steep surface parts seen from a far view) falls back, block-uniformly,
// to pair loads from global memory for that unit only.  Pass 2 has two forms: gather again (gl_aggregate_kernel: the
// features come back from L2 / Infinity Cache, no intermediate in HBM), or - fp16 volume formats, stages with D >= 8 -
// stream the per-view group correlations pass 1 KEPT as fp16 (gl_entropy_kernel<KEEP>: 2 x 16 B per voxel and view;
// corr_aggregate_kernel).  The second gather is bound by its window staging and LDS reads, not by HBM, so trading it
// for 0.4 GB of streamed traffic per reference view (cfg2, stages 1-3) is a net gain.
//
// Algorithmic HBM bytes per launch: pass 1 = features (1 + n_views) * C*HW*sizeof(T) +
// hypotheses D*HW*4 + entropy n_views*HW*4; pass 2 = the same inputs + visibility + G*D*HW*4 volume write.
#pragma once
#include "gather_common.h"

namespace mvs {

#ifndef MVS_GL_CAP
#define MVS_GL_CAP 1024
#endif
constexpr int GL_CAP = MVS_GL_CAP;   // window capacity in source positions: LDS = 2 quads * GL_CAP * 16 B = 32 KiB
constexpr int GL_XALIGN = 8;       // window x origin / width granularity in pixels (32 B of fp32, 16 B of bf16)
constexpr int GL_DCH = 4;          // depth planes per work-item
constexpr int GL_TH = 4;           // tile height in pixels

// One unit = one source view x the GL_DCH depth planes of every work-item of the block.
//   KEEP_GROUPS = false: out[dd]              += wscale * sum_c ref[c] * warped[c, d]                 (pass 1)
//   KEEP_GROUPS = true : out[g * GL_DCH + dd] += wscale * sum_{c in group g} ref[c] * warped[c, d]    (pass 2)
// NOCT = C / 8 channel octets; with 8 groups an octet holds 8 / NOCT whole groups of NOCT channels each.
// `unit` is the block's running unit counter (parity selects the reduction scratch).  Every thread of the block must
// call this (barriers inside); `active` = the thread has a real (pixel, chunk) to work on.
//
// Register discipline: the feature loads are loads from `const __restrict__` memory, which the compiler is free to
// hoist above barriers and out of the (unrolled) octet loop - all octets' reference features and staging values at
// once need > 256 registers.  The per-octet plane offset is therefore laundered through an empty asm statement (it
// becomes a new value the loads depend on), and a scheduling barrier separates the plane pairs of the gather so that
// at most two planes' taps (16 ds_read_b128 results) are in flight.
#ifndef MVS_OPAQUE_SREG
#define MVS_OPAQUE_SREG "s"
#endif
//   W16: the window is staged as ONE fp16 octet per position (16 B: one ds_read_b128 per tap instead of two, twice the positions in
//   the same LDS bytes); the taps enter the fp32 interpolation through v_fma_mix_f32, so the source features are rounded to fp16 once
//   (exact for fp16 / in-range bf16 features) and nothing else changes.  The fp16 volume formats use it (MVS_GATHER_F16).
__device__ __forceinline__ float gl_round_f16(float v) { return (float)(_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f); }

// DIRECT (round 5, MVS_GL_DIRECT16): with fp16 octet tiles in HBM (the producer-side emitter's fp16 hand-off, mvs_conv2d3x3_tiles_fwd) a bilinear
// tap of 8 channels IS one 16-byte run - the four taps of a plane are four buffer_load_dwordx4 with ONE 32-bit offset register (the right
// column through the instruction's immediate offset, the lower row through a wave-uniform soffset), so the unit needs no bounding box, no
// window, no barrier and no LDS at all.  Same fp16 values, same v_fma_mix order as the fp16 window path: bit-identical results.
#ifndef MVS_GL_DIRECT16
#define MVS_GL_DIRECT16 1
#endif
// Where it is used is a measured choice (profiles/r05_gather_direct_pmc.txt, r05_gather_direct_ab.txt): the direct form is bound by the vector-memory
// return path (~25-33 TD cycles per 16-byte-per-lane load), which grows with the octet count, while the window form amortises its bounding box over the
// octets - C = 8 (one octet) wins in every pass on every box measured (-12 ... -16 % pass 2, -8 ... -12 % pass 1); C = 16 loses 5-17 % in the plain passes and
// is box-dependent in the keeping pass 1 (-3.5 ... -6 % on two boxes, +9 % on two others): C >= 16 keeps the windows.  KEEPPASS stays a parameter of the
// choice for that reason.
template <typename T, bool TILED, bool W16, int NOCT, bool KEEPPASS>
constexpr bool gl_direct_v = (MVS_GL_DIRECT16 != 0) && W16 && TILED && std::is_same<T, _Float16>::value && NOCT == 1;

// Round 6 instruction diet of the unit (ISA census in DESIGN.md section 4.1, scripts/isa_census.py): MVS_GL_OPT = 0 rebuilds round 5's form for A/B runs.
//   * the bilinear blend is issued tap-outer / channel-inner: round 5 emitted each channel's four dependent v_fma_mix back to back and the
//     compiler separated them with s_nop (96 per unit: the dependent-VALU hazard of an op_sel source) - eight independent chains interleaved need none
//   * planar staging through ONE buffer descriptor: channel plane c is a wave-uniform soffset, the position a 32-bit voffset - no 64-bit
//     address arithmetic per load (2 VALU each), clamp by v_med3 without the canonicalising v_max the fminf / fmaxf pair costs
//   * the "no tap inside the image" sentinel GL_NONE (0xffff, 0xffff) is neutral for the packed minimum as it stands and, after a packed + 1
//     (which wraps it to 0), for the maximum: no compare / select per plane in the bounding box; its window position is clamped by one
//     v_min_u32 (the weights are zero: any finite window value will do) instead of an exec-masked branch per plane
#ifndef MVS_GL_OPT
#define MVS_GL_OPT 1
#endif
#ifndef MVS_GL_SB
#define MVS_GL_SB 2                // planar staging: rounds of 256 window positions whose loads are issued back to back before any is consumed (1: round-5 order)
#endif
#ifndef MVS_GL_ABL
#define MVS_GL_ABL 0               // measurement only (scripts/gather_ablate.py; results are wrong): 1 no window loads (zeros staged), 2 no staging at all,
#endif                             // 3 no window reads in the gather, 4 no blend arithmetic, 5 no bounding-box reduction (a fixed window), 6 no tap projection

// the four taps of one plane into wv[0..7]: t = 8 halves (4 registers) per tap
#define GL_BLEND8(wv, t00, t01, t10, t11, tp)                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        wv[2 * j] = MVS_FMA_MIX_LO(t00[j], tp.w00, 0.0f);                                              \
        wv[2 * j + 1] = MVS_FMA_MIX_HI(t00[j], tp.w00, 0.0f);                                          \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        wv[2 * j] = MVS_FMA_MIX_LO(t01[j], tp.w01, wv[2 * j]);                                         \
        wv[2 * j + 1] = MVS_FMA_MIX_HI(t01[j], tp.w01, wv[2 * j + 1]);                                 \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        wv[2 * j] = MVS_FMA_MIX_LO(t10[j], tp.w10, wv[2 * j]);                                         \
        wv[2 * j + 1] = MVS_FMA_MIX_HI(t10[j], tp.w10, wv[2 * j + 1]);                                 \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        wv[2 * j] = MVS_FMA_MIX_LO(t11[j], tp.w11, wv[2 * j]);                                         \
        wv[2 * j + 1] = MVS_FMA_MIX_HI(t11[j], tp.w11, wv[2 * j + 1]);                                 \
    }

template <typename T, int NOCT, bool KEEP_GROUPS, bool TILED, bool W16, bool DIRECT = false>
__device__ __forceinline__ void gl_unit(const T* __restrict__ src, const T* __restrict__ ref, const Homography& hm, float fx, float fy,
                                        const float* depth, bool active, int H, int W, unsigned HW, unsigned pc, f32x4* win,
                                        unsigned* red, int unit, float wscale, const float* rf_in, float* out) {
    typedef typename PairOf<T>::type P2;
    constexpr int GPO = 8 / NOCT;          // groups per octet
    constexpr int CPG = NOCT;              // channels per group
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];     // warping.py:90 (once per pixel and view)
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    GTap tp[GL_DCH];
    u16x2 mn = {0xffff, 0xffff}, mx = {0, 0};
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
#if MVS_GL_OPT
    // mx holds the maximum of (x + 1, y + 1): the sentinel GL_NONE wraps to (0, 0), neutral for it; mn takes the sentinel as it is
#pragma unroll
    for (int dd = 0; dd < GL_DCH; ++dd) {
#if MVS_GL_ABL == 6
        tp[dd].pk = ((unsigned)(int)fy << 16) | (unsigned)(int)fx; tp[dd].w00 = depth[dd]; tp[dd].w01 = qx; tp[dd].w10 = 0.25f; tp[dd].w11 = 0.25f;
        if ((int)fx > W - 2 || (int)fy > H - 2) tp[dd].pk = 0;
#else
        tp[dd] = make_gtap(hm, qx, qy, qz, depth[dd], H, W, cx, cy);
#endif
        const u16x2 pkv = gl_as_vec(tp[dd].pk);
        mn = __builtin_elementwise_min(mn, pkv);
        mx = __builtin_elementwise_max(mx, (u16x2)(pkv + (u16x2){1, 1}));
    }
    if (!active) { mn = (u16x2){0xffff, 0xffff}; mx = (u16x2){0, 0}; }
#else
#pragma unroll
    for (int dd = 0; dd < GL_DCH; ++dd) {
        tp[dd] = make_gtap(hm, qx, qy, qz, depth[dd], H, W, cx, cy);
        if (active && tp[dd].pk != GL_NONE) {
            mn = __builtin_elementwise_min(mn, gl_as_vec(tp[dd].pk));
            mx = __builtin_elementwise_max(mx, gl_as_vec(tp[dd].pk));
        }
    }
#endif
    if constexpr (DIRECT) {
        static_assert(W16 && TILED && std::is_same<T, _Float16>::value, "the direct form reads fp16 octet tiles");
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned tofs[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) {
            const unsigned pk = tp[dd].pk;
            tofs[dd] = pk == GL_NONE ? 0u : ((pk >> 16) * (unsigned)W + (pk & 0xffffu)) * 16u;
        }
        const int rowb = W * 16;                                 // wave-uniform: the lower tap row through soffset
        if (!active) return;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            unsigned oofs = gl_octet_offset(o, HW);
            asm volatile("" : "+" MVS_OPAQUE_SREG(oofs));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src + oofs), 0, (int)(HW * 16u), 0x00020000);
            float rf[8];
            if (NOCT > 1) {
                gl_load8<TILED, T>(ref + oofs, HW, pc, rf);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) rf[c] = rf_in[c];
            }
            u32x4 t[GL_DCH][4];
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
#if MVS_GL_DIRECT16 == 3      // ablation (scripts/prof_gather_direct.py): no loads
                (void)rs; (void)rowb;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[dd][k] = u32x4{tofs[dd], tofs[dd] + (unsigned)k, tofs[dd] ^ 0x3c003c00u, 0x3c003c00u};
#else
                t[dd][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)tofs[dd], 0, 0);
                t[dd][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tofs[dd] + 16u), 0, 0);
                t[dd][2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)tofs[dd], rowb, 0);
                t[dd][3] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tofs[dd] + 16u), rowb, 0);
#endif
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) rf[c] *= wscale;
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
                float wv[8];
#if MVS_GL_DIRECT16 == 2      // ablation: the loads are consumed by one xor each, no interpolation
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned x = t[dd][0][j] ^ t[dd][1][j] ^ t[dd][2][j] ^ t[dd][3][j];
                    wv[2 * j] = __builtin_bit_cast(float, x & 0x3fffffffu);
                    wv[2 * j + 1] = tp[dd].w00;
                }
#elif MVS_GL_OPT
                GL_BLEND8(wv, t[dd][0], t[dd][1], t[dd][2], t[dd][3], tp[dd])
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = MVS_FMA_MIX_LO(t[dd][0][j], tp[dd].w00, 0.0f);
                    a = MVS_FMA_MIX_LO(t[dd][1][j], tp[dd].w01, a);
                    a = MVS_FMA_MIX_LO(t[dd][2][j], tp[dd].w10, a);
                    wv[2 * j] = MVS_FMA_MIX_LO(t[dd][3][j], tp[dd].w11, a);
                    float b = MVS_FMA_MIX_HI(t[dd][0][j], tp[dd].w00, 0.0f);
                    b = MVS_FMA_MIX_HI(t[dd][1][j], tp[dd].w01, b);
                    b = MVS_FMA_MIX_HI(t[dd][2][j], tp[dd].w10, b);
                    wv[2 * j + 1] = MVS_FMA_MIX_HI(t[dd][3][j], tp[dd].w11, b);
                }
#endif
                if (KEEP_GROUPS) {
#pragma unroll
                    for (int j = 0; j < GPO; ++j) {
                        float s = out[(o * GPO + j) * GL_DCH + dd];
#pragma unroll
                        for (int c = 0; c < CPG; ++c) s += rf[j * CPG + c] * wv[j * CPG + c];
                        out[(o * GPO + j) * GL_DCH + dd] = s;
                    }
                } else {
                    float s = out[dd];
#pragma unroll
                    for (int c = 0; c < 8; ++c) s += rf[c] * wv[c];
                    out[dd] = s;
                }
            }
        }
        return;
    }
#if MVS_GL_ABL == 5
    {
        __syncthreads();
        const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)tp[0].pk);
        const unsigned bx = (p0 & 0xffffu) > 8u ? (p0 & 0xffffu) - 8u : 0u, by = (p0 >> 16) > 2u ? (p0 >> 16) - 2u : 0u;
        mn = gl_as_vec((by << 16) | bx);
        mx = gl_as_vec(((by + 9u) << 16) | (bx + 81u));
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) tp[dd].pk = ((by + 1u + (unsigned)(lane & 3)) << 16) | (bx + 1u + (unsigned)(lane & 63));
    }
#else
    mn = gl_wave_reduce<false>(mn);                               // lane 63 holds the wave's result
    mx = gl_wave_reduce<true>(mx);
    unsigned* rd = red + (unit & 1) * 8;
    if (lane == 63) { rd[wave] = gl_as_u32(mn); rd[4 + wave] = gl_as_u32(mx); }
    __syncthreads();      // (A) bounding box complete; every thread has left the previous unit's gather
    mn = __builtin_elementwise_min(__builtin_elementwise_min(gl_as_vec(rd[0]), gl_as_vec(rd[1])),
                                   __builtin_elementwise_min(gl_as_vec(rd[2]), gl_as_vec(rd[3])));
    mx = __builtin_elementwise_max(__builtin_elementwise_max(gl_as_vec(rd[4]), gl_as_vec(rd[5])),
                                   __builtin_elementwise_max(gl_as_vec(rd[6]), gl_as_vec(rd[7])));
#endif
#if MVS_GL_OPT
    const int xmin = mn[0], ymin = mn[1], xmax = (int)mx[0] - 1, ymax = (int)mx[1] - 1;       // mx = maximum of (x + 1, y + 1); 0 = nothing
#else
    const int xmin = mn[0], ymin = mn[1], xmax = mx[0], ymax = mx[1];
#endif
    if (xmax < xmin) return;                                    // no tap of the whole tile is inside the source image
    const int wx0 = xmin & ~(GL_XALIGN - 1);
    const int ww = (xmax + 2 - wx0 + GL_XALIGN - 1) & ~(GL_XALIGN - 1);
    const int wh = ymax + 2 - ymin;
    const int n = ww * wh;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8* win16 = reinterpret_cast<h8*>(win);
    if (n <= (W16 ? 2 * GL_CAP : GL_CAP)) {
        unsigned pos[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) {
            const unsigned pk = tp[dd].pk;
#if MVS_GL_OPT
            // a 2x2 block inside the window sits at <= n - ww - 2; GL_NONE (weights zero) lands far beyond and is clamped there
            const unsigned raw = ((pk >> 16) - (unsigned)ymin) * (unsigned)ww + ((pk & 0xffffu) - (unsigned)wx0);
            pos[dd] = raw < (unsigned)(n - ww - 2) ? raw : (unsigned)(n - ww - 2);
#else
            pos[dd] = pk == GL_NONE ? 0u : ((pk >> 16) - (unsigned)ymin) * (unsigned)ww + ((pk & 0xffffu) - (unsigned)wx0);
#endif
        }
        const float inv_ww = __builtin_amdgcn_rcpf((float)ww) * 1.000001f;   // row = floor((i + 0.5) / ww): exact for i < 2^16
        const unsigned gbase = (unsigned)ymin * (unsigned)W + (unsigned)wx0;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (o > 0) __syncthreads();                         // (C) the previous octet's taps have been read
            unsigned oofs = gl_octet_offset(o, HW);
            asm volatile("" : "+" MVS_OPAQUE_SREG(oofs));
            const T* so = src + oofs;
            // reference features of this octet: issued ahead of the staging loop so that they land while it runs (C = 8: the
            // caller loaded them once per block)
            float rf[8];
            if (NOCT > 1) {
                gl_load8<TILED, T>(ref + oofs, HW, pc, rf);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) rf[c] = rf_in[c];
            }
#if MVS_GL_OPT
            // planar maps: one descriptor over the octet's 8 channel planes; plane c = wave-uniform soffset, position = 32-bit voffset
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(so), 0, (int)(8u * HW * (unsigned)sizeof(T)), 0x00020000);
            const unsigned planeb = HW * (unsigned)sizeof(T);
#endif
            int istart = tid;
#if MVS_GL_OPT && MVS_GL_ABL == 0
            if constexpr (!TILED) {
                // Round 6 (profiles/r06_gather_ablation.txt): the rolled loop below - load 8 dwords, wait, convert, ds_write - exposed one global-memory
                // latency per round of 256 positions, two or three rounds per unit between two barriers: 19-36 % of a fine-stage pass.  Here the loads
                // of MVS_GL_SB rounds are all in flight before the first is consumed (8 VGPRs per extra round).  Measured (profiles/r06_gather_staging_rounds_ab.txt):
                // stage-4 pass 1 -4 %, pass 2 -5 %; at C = 16 the extra registers cost the keeping pass its fourth wave per SIMD (+12 %), at C >= 32 nothing
                // moves: one octet per unit only.  (The ablation's 19-36 % are mostly the HBM time of the feature maps themselves, which the
                // barrier-separated phases of four resident blocks overlap imperfectly - not a latency two rounds in flight could hide.)
                constexpr int SBR = NOCT == 1 ? MVS_GL_SB : 1;
#pragma unroll 1
                for (int i0 = tid; i0 < n; i0 += 256 * SBR) {
                    float v[SBR][8];
#pragma unroll
                    for (int k = 0; k < SBR; ++k) {
                        const int i = i0 + 256 * k;
                        if (i < n) {
                            const int row = (int)(((float)i + 0.5f) * inv_ww);
                            const unsigned g = gbase + (unsigned)row * (unsigned)(W - ww) + (unsigned)i;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                if constexpr (sizeof(T) == 4)
                                    v[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, (int)(g * 4u), (int)((unsigned)c * planeb), 0));
                                else
                                    v[k][c] = to_f32(__builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(srs, (int)(g * 2u), (int)((unsigned)c * planeb), 0)));
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < SBR; ++k) {
                        const int i = i0 + 256 * k;
                        if (i < n) {
                            if (W16) {
                                h8 hv;
#pragma unroll
                                for (int c = 0; c < 8; ++c) hv[c] = (_Float16)__builtin_amdgcn_fmed3f(v[k][c], -65504.0f, 65504.0f);
                                win16[i] = hv;
                            } else {
                                win[i] = f32x4{v[k][0], v[k][1], v[k][2], v[k][3]};
                                win[GL_CAP + i] = f32x4{v[k][4], v[k][5], v[k][6], v[k][7]};
                            }
                        }
                    }
                }
                istart = n;
            }
#endif
#pragma unroll 1
            for (int i = istart; i < (MVS_GL_ABL == 2 ? 0 : n); i += 256) {
#if MVS_GL_ABL == 1
                if (W16) { h8 hz; for (int c = 0; c < 8; ++c) hz[c] = (_Float16)(float)(i & 7); win16[i] = hz; } else { win[i] = f32x4{1, 2, 3, 4}; win[GL_CAP + i] = f32x4{1, 2, 3, 4}; }
                continue;
#endif
                const int row = (int)(((float)i + 0.5f) * inv_ww);
                const unsigned g = gbase + (unsigned)row * (unsigned)(W - ww) + (unsigned)i;    // (ymin+row)*W + wx0 + (i - row*ww)
#if MVS_GL_OPT
                if constexpr (!TILED) {
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if constexpr (sizeof(T) == 4)
                            v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, (int)(g * 4u), (int)((unsigned)c * planeb), 0));
                        else
                            v[c] = to_f32(__builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(srs, (int)(g * 2u), (int)((unsigned)c * planeb), 0)));
                    }
                    if (W16) {
                        h8 hv;
#pragma unroll
                        for (int c = 0; c < 8; ++c) hv[c] = (_Float16)__builtin_amdgcn_fmed3f(v[c], -65504.0f, 65504.0f);
                        win16[i] = hv;
                    } else {
                        win[i] = f32x4{v[0], v[1], v[2], v[3]};
                        win[GL_CAP + i] = f32x4{v[4], v[5], v[6], v[7]};
                    }
                    continue;
                }
#endif
                if constexpr (W16 && TILED && std::is_same<T, _Float16>::value) {
                    // fp16 octet tiles (what a producer-side emitter hands over, mvs_conv2d3x3_tiles_fwd with out_dtype fp16): the window
                    // position IS the 16-byte run in HBM - staging is a copy, no conversion, no clamp (round 5)
                    win16[i] = *(reinterpret_cast<const h8*>(so) + g);
                    continue;
                }
                float v[8];
                gl_load8<TILED, T>(so, HW, g, v);
                if (W16) {
                    h8 hv;
#pragma unroll
                    for (int c = 0; c < 8; ++c) hv[c] = (_Float16)fminf(fmaxf(v[c], -65504.0f), 65504.0f);
                    win16[i] = hv;
                } else {
                    win[i] = f32x4{v[0], v[1], v[2], v[3]};
                    win[GL_CAP + i] = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
            __syncthreads();                                    // (B) window of octet o is in LDS
            if (active) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rf[c] *= wscale;
#pragma unroll
                for (int dd = 0; dd < GL_DCH; ++dd) {
                    if (dd == 2) __builtin_amdgcn_sched_barrier(0);
                    float wv[8];
                    if (W16) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4* w0 = reinterpret_cast<const u32x4*>(win16) + pos[dd];
#if MVS_GL_ABL == 3
                        const u32x4 t00 = {pos[dd], 0x3c003c00u, 0x3c003c00u, pos[dd]}, t01 = t00, t10 = {0x3c003c00u, pos[dd], pos[dd], 0x3c003c00u}, t11 = t10;
#else
                        const u32x4 t00 = w0[0], t01 = w0[1], t10 = w0[ww], t11 = w0[ww + 1];
#endif
#if MVS_GL_ABL == 4
#pragma unroll
                        for (int j = 0; j < 4; ++j) { wv[2 * j] = __builtin_bit_cast(float, (t00[j] ^ t01[j] ^ t10[j] ^ t11[j]) & 0x3fffffffu); wv[2 * j + 1] = tp[dd].w00; }
#elif MVS_GL_OPT
                        GL_BLEND8(wv, t00, t01, t10, t11, tp[dd])
#else
#pragma unroll
                        for (int j = 0; j < 4; ++j) {            // 4 v_fma_mix_f32 per channel, no conversion instructions
                            float a = MVS_FMA_MIX_LO(t00[j], tp[dd].w00, 0.0f);
                            a = MVS_FMA_MIX_LO(t01[j], tp[dd].w01, a);
                            a = MVS_FMA_MIX_LO(t10[j], tp[dd].w10, a);
                            wv[2 * j] = MVS_FMA_MIX_LO(t11[j], tp[dd].w11, a);
                            float b = MVS_FMA_MIX_HI(t00[j], tp[dd].w00, 0.0f);
                            b = MVS_FMA_MIX_HI(t01[j], tp[dd].w01, b);
                            b = MVS_FMA_MIX_HI(t10[j], tp[dd].w10, b);
                            wv[2 * j + 1] = MVS_FMA_MIX_HI(t11[j], tp[dd].w11, b);
                        }
#endif
                    } else {
                        const f32x4* w0 = win + pos[dd];
                        const f32x4 a0 = w0[0], a1 = w0[1], b0 = w0[ww], b1 = w0[ww + 1];
                        const f32x4 c0 = w0[GL_CAP], c1 = w0[GL_CAP + 1], d0 = w0[GL_CAP + ww], d1 = w0[GL_CAP + ww + 1];
                        f32x4 lo = a0 * tp[dd].w00;
                        lo += a1 * tp[dd].w01;
                        lo += b0 * tp[dd].w10;
                        lo += b1 * tp[dd].w11;
                        f32x4 hi = c0 * tp[dd].w00;
                        hi += c1 * tp[dd].w01;
                        hi += d0 * tp[dd].w10;
                        hi += d1 * tp[dd].w11;
                        wv[0] = lo[0]; wv[1] = lo[1]; wv[2] = lo[2]; wv[3] = lo[3]; wv[4] = hi[0]; wv[5] = hi[1]; wv[6] = hi[2]; wv[7] = hi[3];
                    }
                    if (KEEP_GROUPS) {
#pragma unroll
                        for (int j = 0; j < GPO; ++j) {
                            float s = out[(o * GPO + j) * GL_DCH + dd];
#pragma unroll
                            for (int c = 0; c < CPG; ++c) s += rf[j * CPG + c] * wv[j * CPG + c];
                            out[(o * GPO + j) * GL_DCH + dd] = s;
                        }
                    } else {
                        float s = out[dd];
#pragma unroll
                        for (int c = 0; c < 8; ++c) s += rf[c] * wv[c];
                        out[dd] = s;
                    }
                }
            }
        }
    } else if (active) {
        // window larger than the LDS capacity: this unit gathers straight from global memory with pair loads
        unsigned top[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) top[dd] = tp[dd].pk == GL_NONE ? 0u : (tp[dd].pk >> 16) * (unsigned)W + (tp[dd].pk & 0xffffu);
#pragma unroll 1
        for (int c = 0; c < 8 * NOCT; ++c) {                    // rolled: one channel's taps in flight (rare path)
            // element offset of channel c at position 0 and the stride between horizontally adjacent positions
            const unsigned cbase = TILED ? (unsigned)(c >> 3) * 8u * HW + (unsigned)(c & 7) : (unsigned)c * HW;
            const unsigned pstep = TILED ? 8u : 1u;
            const float rfc = to_f32(ref[cbase + pc * pstep]) * wscale;
            const T* sp = src + cbase;
            const int g = c / CPG;
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
                float t0, t1, b0, b1;
                if (TILED) {
                    __builtin_amdgcn_sched_barrier(0);          // one plane's four taps in flight: the addresses are the register hog
                    t0 = to_f32(sp[top[dd] * 8u]); t1 = to_f32(sp[(top[dd] + 1u) * 8u]);
                    b0 = to_f32(sp[(top[dd] + (unsigned)W) * 8u]); b1 = to_f32(sp[(top[dd] + (unsigned)W + 1u) * 8u]);
                } else {
                    const P2 t = *reinterpret_cast<const P2*>(sp + top[dd]);
                    const P2 b = *reinterpret_cast<const P2*>(sp + top[dd] + (unsigned)W);
                    t0 = to_f32(t.x); t1 = to_f32(t.y); b0 = to_f32(b.x); b1 = to_f32(b.y);
                }
                if (W16) {                                      // the same once-rounded source features as the fp16 window
                    t0 = gl_round_f16(t0); t1 = gl_round_f16(t1); b0 = gl_round_f16(b0); b1 = gl_round_f16(b1);
                }
                float wv = tp[dd].w00 * t0;
                wv += tp[dd].w01 * t1;
                wv += tp[dd].w10 * b0;
                wv += tp[dd].w11 * b1;
                if (KEEP_GROUPS) {
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg) out[gg * GL_DCH + dd] += (g == gg) ? rfc * wv : 0.0f;
                } else {
                    out[dd] += rfc * wv;
                }
            }
        }
    }
}

// dynamic LDS: [ window: 2 * GL_CAP f32x4 ][ red: 16 u32 ][ sim: D * TP floats (pass 1 only) ]
constexpr size_t GL_WIN_BYTES = (size_t)2 * GL_CAP * 16;
constexpr size_t GL_RED_BYTES = 64;

// Work decomposition shared by both passes (compile-time: no run-time divisions in the prologue).  NS work-items share a
// pixel, one per chunk of GL_DCH planes ("slot"); a block = TP = 256 / NS pixels = a tile of TW x 4.  Chunk groups of NS
// chunks beyond the first are iterations (`it`) of a loop (pass 1) or blocks along grid.y (pass 2).
template <int NS>
struct GlTile {
    static constexpr int TP = 256 / NS, TW = TP / GL_TH;
    int slot, pi, x, y;
    bool valid;
    unsigned pc;
    float fx, fy;
    __device__ __forceinline__ GlTile(int blk, int ntx, int H, int W) {
        const int tid = (int)threadIdx.x;
        const int ty = blk / ntx, tx = blk - ty * ntx;
        slot = tid / TP;
        pi = tid % TP;
        x = tx * TW + pi % TW;
        y = ty * GL_TH + pi / TW;
        valid = x < W && y < H;
        pc = valid ? (unsigned)y * (unsigned)W + (unsigned)x : (unsigned)H * (unsigned)W - 1u;
        fx = (float)(valid ? x : W - 1);
        fy = (float)(valid ? y : H - 1);
    }
};

// ------------------------------------------------------------------------------------------------
// pass 1: entropy of the depth-softmax of the group-summed correlation          cost_volume.py:79-92
// grid = (tiles, ceil(views in launch / vpb), B); a block walks `vpb` consecutive source views of its tile
// ------------------------------------------------------------------------------------------------
// KEEP: the per-view GROUP correlations (mean over the group's channels, cost_volume.py:79-84) are also written for
// corr_aggregate_kernel, [B, V-1, D, HW, 8]; the group sum the entropy needs is taken from them.  KEEP && W16: fp16 windows and fp16
// correlations clamped to the fp16 range (MVS_CORR_F16, 16 B per voxel and view); KEEP && !W16: fp32 windows and fp32 correlations
// (MVS_CORR_F32, 32 B per voxel and view - the exact form: what the second gather would have recomputed, round 5).
template <int DT, int NOCT, int NS, bool TILED, bool KEEP, bool W16>
__global__ __launch_bounds__(256) void gl_entropy_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                         const float* __restrict__ hyp, float* __restrict__ entropy,
                                                         void* __restrict__ corr, int V, int D, int H,
                                                         int W, int view_begin, int view_end, int vpb, int ntx, int nblk) {
    typedef typename FeatT<DT>::type T;
    constexpr bool DIRECT = gl_direct_v<T, TILED, W16, NOCT, KEEP>;
    HIP_DYNAMIC_SHARED(float, smem)
    f32x4* win = reinterpret_cast<f32x4*>(smem);
    unsigned* red = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES);
    float* sim = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES + GL_RED_BYTES);
    constexpr int C = 8 * NOCT;
    constexpr int TP = GlTile<NS>::TP;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int b = (int)blockIdx.z;
    const GlTile<NS> t((int)xcd_remap(blockIdx.x, (unsigned)nblk), ntx, H, W);
    const int nch = (D + GL_DCH - 1) / GL_DCH, niter = (nch + NS - 1) / NS;
    const T* feat = reinterpret_cast<const T*>(feat_) + (size_t)(b * V) * C * HW;
    const T* ref = feat;
    const float* hp = hyp + (size_t)b * D * HW;
    const float inv_cpg = 1.0f / (float)NOCT;
    float rf0[8];                                               // C = 8: the pixel's reference features, once per block
    if (NOCT == 1) gl_load8<TILED, T>(ref, HW, t.pc, rf0);
    else {
#pragma unroll
        for (int c = 0; c < 8; ++c) rf0[c] = 0.0f;
    }
    const int v0 = view_begin + (int)blockIdx.y * vpb, v1 = v0 + vpb < view_end ? v0 + vpb : view_end;
    int unit = 0;
    float sat_amax = 0.0f;                                      // fp16 saturation counter (mvs_common.h), KEEP only
    for (int v = v0; v < v1; ++v) {
        const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
        const T* src = feat + (size_t)v * C * HW;
        float s[GL_DCH];
        for (int it = 0; it < niter; ++it, ++unit) {
            const int chunk = it * NS + t.slot;
            const bool active = t.valid && chunk < nch;
            const int d0 = (chunk < nch ? chunk : nch - 1) * GL_DCH;
            float depth[GL_DCH];
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) depth[dd] = hp[(unsigned)(d0 + dd < D ? d0 + dd : D - 1) * HW + t.pc];
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) s[dd] = 0.0f;
            if (KEEP) {
                float acc[8 * GL_DCH];
#pragma unroll
                for (int i = 0; i < 8 * GL_DCH; ++i) acc[i] = 0.0f;
                gl_unit<T, NOCT, true, TILED, W16, DIRECT>(src, ref, hm, t.fx, t.fy, depth, active, H, W, HW, t.pc, win, red, unit, inv_cpg, rf0, acc);
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                const size_t cbase = (size_t)(b * (V - 1) + (v - 1)) * D * HW + t.pc;
#pragma unroll
                for (int dd = 0; dd < GL_DCH; ++dd) {
                    float r[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) { r[g] = acc[g * GL_DCH + dd]; s[dd] += r[g]; }
                    if (active && d0 + dd < D) {
                        const size_t vox = cbase + (size_t)(unsigned)(d0 + dd) * HW;
                        if constexpr (W16) {
                            h8 hv;
#pragma unroll
                            for (int g = 0; g < 8; ++g) hv[g] = (_Float16)fminf(fmaxf(r[g], -65504.0f), 65504.0f);
                            sat::track(sat_amax, r[0], r[1], r[2], r[3]);
                            sat::track(sat_amax, r[4], r[5], r[6], r[7]);
                            reinterpret_cast<h8*>(corr)[vox] = hv;
                        } else {
                            f32x4* cv = reinterpret_cast<f32x4*>(corr) + vox * 2;
                            cv[0] = f32x4{r[0], r[1], r[2], r[3]};
                            cv[1] = f32x4{r[4], r[5], r[6], r[7]};
                        }
                    }
                }
            } else {
                gl_unit<T, NOCT, false, TILED, W16, DIRECT>(src, ref, hm, t.fx, t.fy, depth, active, H, W, HW, t.pc, win, red, unit, inv_cpg, rf0, s);   // sum_g mean_c = (1/cpg) sum_c
            }
            if (NS > 1 || niter > 1) {
                if (chunk < nch) {
#pragma unroll
                    for (int dd = 0; dd < GL_DCH; ++dd)
                        if (d0 + dd < D) sim[(d0 + dd) * TP + t.pi] = s[dd];
                }
            }
        }
        float* dst = entropy + (size_t)(b * (V - 1) + (v - 1)) * HW + t.pc;
        if (NS == 1 && niter == 1) {
            // every work-item owns all D <= 4 planes of its pixel: softmax-entropy straight from registers
            const float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            float m = s[0];
#pragma unroll
            for (int dd = 1; dd < GL_DCH; ++dd) m = dd < D ? fmaxf(m, s[dd]) : m;
            float e[GL_DCH], den = 0.0f;
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) { e[dd] = dd < D ? __builtin_amdgcn_exp2f((s[dd] - m) * LOG2E) : 0.0f; den += e[dd]; }
            const float rden = 1.0f / den;
            float ent = 0.0f;
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
                const float pr = e[dd] * rden;
                if (dd < D) ent -= pr * (__builtin_amdgcn_logf(pr + 1e-7f) * LN2);                  // cost_volume.py:92
            }
            if (t.valid) *dst = ent;
        } else {
            __syncthreads();
            // the next view's first sim store comes after its unit's barrier (A): no second barrier needed here
            if (t.slot == 0 && t.valid) gl_softmax_entropy_store(sim + t.pi, TP, D, dst);
            if constexpr (DIRECT) __syncthreads();   // no barrier (A) in the direct unit: the next view's sim stores wait here
        }
    }
    if (KEEP && W16) sat::commit(sat_amax);
}

// ------------------------------------------------------------------------------------------------
// pass 2: visibility-weighted aggregation over the source views of the launch    cost_volume.py:97-101
// grid = (tiles, chunk groups, B); output channel-last [D,HW,8].
// ------------------------------------------------------------------------------------------------
// W16 (fp16 windows) is what the launcher picks for the fp16 volume format
template <int DT, int NOCT, int NS, bool TILED, bool W16>
__global__ __launch_bounds__(256) void gl_aggregate_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                           const float* __restrict__ hyp, const float* __restrict__ vis,
                                                           float* __restrict__ vol, float* __restrict__ vis_sum, int normalise, int V,
                                                           int D, int H, int W, int view_begin, int view_end, int ntx, int nblk) {
    typedef typename FeatT<DT>::type T;
    HIP_DYNAMIC_SHARED(float, smem)
    f32x4* win = reinterpret_cast<f32x4*>(smem);
    unsigned* red = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES);
    constexpr int C = 8 * NOCT;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int b = (int)blockIdx.z, it = (int)blockIdx.y;
    const GlTile<NS> t((int)xcd_remap(blockIdx.x, (unsigned)nblk), ntx, H, W);
    const int nch = (D + GL_DCH - 1) / GL_DCH;
    const T* feat = reinterpret_cast<const T*>(feat_) + (size_t)(b * V) * C * HW;
    const T* ref = feat;
    const float* hp = hyp + (size_t)b * D * HW;
    const float* vp = vis + (size_t)(b * (V - 1)) * HW + t.pc;
    float vsum = 0.0f;
    for (int v = view_begin; v < view_end; ++v) vsum += vp[(unsigned)(v - 1) * HW];               // cost_volume.py:98
    if (vis_sum != nullptr && it == 0 && t.slot == 0 && t.valid) vis_sum[(size_t)b * HW + t.pc] = vsum;
    const float rdenom = (normalise & 1) ? 1.0f / (vsum + 1e-6f) : 1.0f;                          // cost_volume.py:101
    const bool split_out = (normalise & 2) != 0;                 // volume in the split activation format of the bf16x3 U-Net
    const bool f16_out = (normalise & 4) != 0;                   // volume as fp16 [D,HW,8] (MVS_PREC_F16X2 U-Net), clamped to the fp16 range
    float sat_amax = 0.0f;                                       // fp16 saturation counter (mvs_common.h)
    const float inv_cpg = 1.0f / (float)NOCT;
    float rf0[8];                                               // C = 8: the pixel's reference features, once per block
    if (NOCT == 1) gl_load8<TILED, T>(ref, HW, t.pc, rf0);
    else {
#pragma unroll
        for (int c = 0; c < 8; ++c) rf0[c] = 0.0f;
    }
    const int chunk = it * NS + t.slot;
    const bool active = t.valid && chunk < nch;
    const int d0 = (chunk < nch ? chunk : nch - 1) * GL_DCH;
    float depth[GL_DCH];
#pragma unroll
    for (int dd = 0; dd < GL_DCH; ++dd) depth[dd] = hp[(unsigned)(d0 + dd < D ? d0 + dd : D - 1) * HW + t.pc];
    float acc[8 * GL_DCH];
#pragma unroll
    for (int i = 0; i < 8 * GL_DCH; ++i) acc[i] = 0.0f;
    int unit = 0;
    for (int v = view_begin; v < view_end; ++v, ++unit) {
        const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
        const float w = vp[(unsigned)(v - 1) * HW];                                               // cost_volume.py:97
        gl_unit<T, NOCT, true, TILED, W16, gl_direct_v<T, TILED, W16, NOCT, false>>(feat + (size_t)v * C * HW, ref, hm, t.fx, t.fy, depth, active, H, W, HW, t.pc, win, red, unit, inv_cpg * w, rf0, acc);
    }
    if (active) {
        float* vb = vol + (size_t)b * D * HW * 8;
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) {
            if (d0 + dd >= D) continue;
            float r[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) r[g] = acc[g * GL_DCH + dd] * rdenom;
            if (f16_out) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 hv;
#pragma unroll
                for (int g = 0; g < 8; ++g) hv[g] = (_Float16)fminf(fmaxf(r[g], -65504.0f), 65504.0f);
                sat::track(sat_amax, r[0], r[1], r[2], r[3]);
                sat::track(sat_amax, r[4], r[5], r[6], r[7]);
                *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(vol) + ((size_t)b * D * HW + (size_t)(unsigned)(d0 + dd) * HW + t.pc) * 8) = hv;
                continue;
            }
            f32x4* o = reinterpret_cast<f32x4*>(vb + ((size_t)(unsigned)(d0 + dd) * HW + t.pc) * 8);
            if (split_out) {                                    // [hi x8 | lo x8] bf16: the same 32 bytes (conv_bf16x3_kernels.hip)
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
                continue;
            }
            o[0] = f32x4{r[0], r[1], r[2], r[3]};
            o[1] = f32x4{r[4], r[5], r[6], r[7]};
        }
    }
    if (f16_out) sat::commit(sat_amax);
}

// ------------------------------------------------------------------------------------------------
// host-side launchers shared by the translation units of the LDS-staged gather.  The instantiations are spread over five files
// (gather_lds_kernels / _entropy_w16 / _keep / _aggregate / _aggregate_w16 .hip) only to compile them in parallel.
// ------------------------------------------------------------------------------------------------
static int gl_slots(int D) {
    const int nch = (D + GL_DCH - 1) / GL_DCH;
    return nch >= 8 ? 8 : (nch >= 4 ? 4 : (nch >= 2 ? 2 : 1));
}

bool gl_window_f16_enabled();
bool gl_supported(int C, int G, int D, int H, int W);

template <int DT, int NOCT, int NS, bool TILED, bool KEEP = false, bool W16 = false>
static int gl_launch_entropy_t(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int D, int H, int W, int vb,
                               int ve, hipStream_t st, void* corr = nullptr) {
    constexpr int TP = 256 / NS, TW = TP / GL_TH;
    const int ntx = (int)ceil_div(W, TW), nty = (int)ceil_div(H, GL_TH);
    const int nblk = ntx * nty;
    // plenty of tiles: one block walks all views of its tile (prologue, hypotheses and reference features amortised);
    // few tiles (coarse stages): one block per (tile, view) so that the chip fills
    const int vpb = (long long)nblk * B >= 4096 ? ve - vb : 1;
    const size_t lds = GL_WIN_BYTES + GL_RED_BYTES + (size_t)D * TP * sizeof(float);
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gl_entropy_kernel<DT, NOCT, NS, TILED, KEEP, W16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gl_entropy_kernel<DT, NOCT, NS, TILED, KEEP, W16>), dim3(nblk, ceil_div(ve - vb, vpb), B), dim3(256), lds, st, feat, hom, hyp, ent,
                       corr, V, D, H, W, vb, ve, vpb, ntx, nblk);
    return check_launch("gl_entropy_kernel");
}

template <int DT, int NOCT, int NS, bool TILED, bool W16 = false>
static int gl_launch_aggregate_t(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                                 int normalise, int B, int V, int D, int H, int W, int vb, int ve, hipStream_t st) {
    constexpr int TP = 256 / NS, TW = TP / GL_TH;
    const int ntx = (int)ceil_div(W, TW), nty = (int)ceil_div(H, GL_TH);
    const int nblk = ntx * nty;
    const int nch = (D + GL_DCH - 1) / GL_DCH, niter = (nch + NS - 1) / NS;
    const size_t lds = GL_WIN_BYTES + GL_RED_BYTES;
    hipLaunchKernelGGL((gl_aggregate_kernel<DT, NOCT, NS, TILED, W16>), dim3(nblk, niter, B), dim3(256), lds, st, feat, hom, hyp, vis, vol, vis_sum,
                       normalise, V, D, H, W, vb, ve, ntx, nblk);
    return check_launch("gl_aggregate_kernel");
}

#define GL_DISPATCH_NS(FN, DTV, NOCTV, ...)                                                    \
    switch (gl_slots(D) * 2 + (layout == MVS_LAYOUT_OCTET_TILED ? 1 : 0)) {                    \
        case 2: return FN<DTV, NOCTV, 1, false>(__VA_ARGS__);                                  \
        case 3: return FN<DTV, NOCTV, 1, true>(__VA_ARGS__);                                   \
        case 4: return FN<DTV, NOCTV, 2, false>(__VA_ARGS__);                                  \
        case 5: return FN<DTV, NOCTV, 2, true>(__VA_ARGS__);                                   \
        case 8: return FN<DTV, NOCTV, 4, false>(__VA_ARGS__);                                  \
        case 9: return FN<DTV, NOCTV, 4, true>(__VA_ARGS__);                                   \
        case 16: return FN<DTV, NOCTV, 8, false>(__VA_ARGS__);                                 \
        default: return FN<DTV, NOCTV, 8, true>(__VA_ARGS__);                                  \
    }
#define GL_DISPATCH_C(FN, DTV, ...)                                                            \
    switch (C) {                                                                               \
        case 8: GL_DISPATCH_NS(FN, DTV, 1, __VA_ARGS__)                                        \
        case 16: GL_DISPATCH_NS(FN, DTV, 2, __VA_ARGS__)                                       \
        case 32: GL_DISPATCH_NS(FN, DTV, 4, __VA_ARGS__)                                       \
        default: GL_DISPATCH_NS(FN, DTV, 8, __VA_ARGS__)                                       \
    }
#define GL_DISPATCH(FN, ...)                                                                   \
    do {                                                                                       \
        switch (dtype) {                                                                       \
            case MVS_DTYPE_F32: GL_DISPATCH_C(FN, MVS_DTYPE_F32, __VA_ARGS__)                  \
            case MVS_DTYPE_BF16: GL_DISPATCH_C(FN, MVS_DTYPE_BF16, __VA_ARGS__)                \
            default: GL_DISPATCH_C(FN, MVS_DTYPE_F16, __VA_ARGS__)                             \
        }                                                                                      \
    } while (0)


}  // namespace mvs
