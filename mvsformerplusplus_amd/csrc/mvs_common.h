// Shared declarations for the gfx950 kernels of the MVSFormer++ depth hot path.
//
// Conventions
//   * one wavefront = 64 lanes; every workgroup is 1-D with 256 threads (4 waves) unless stated
//   * feature maps arrive in the caller's layout [B, V, C, H, W] (fp32 / bf16 / fp16, upcast in-kernel,
//     reference: cost_volume.py:67,81,84)
//   * everything the library produces for itself (cost volume, U-Net activations) is channel-last
//     fp32 [B, D, H, W, C] so that one voxel's channels are one contiguous 16-byte-aligned run
//   * hypotheses, entropy / visibility maps, logits, depth, confidence are planar fp32
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/mvs_hip.h"

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from a per-lane global address to (wave-uniform LDS base) + 16 * lane; no VGPRs,
// no ds_write.  Completion is tracked by vmcnt (MVS_WAIT_VMEM) and needs a barrier before other waves read.  The host emulator of the
// tests defines its own (synchronous) forms first.
#ifndef MVS_GLOBAL_LOAD_LDS16
#define MVS_GLOBAL_LOAD_LDS16(gsrc, ldst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
#define MVS_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// v_fma_mix_f32: fp32 fma whose first operand is the low / high fp16 half of a 32-bit register (converted on the fly - no v_cvt, and
// the compiler cannot turn it into v_cvt + v_pk_fma_f32, which costs 1.5x the instructions).  The host emulator defines C forms first.
#ifndef MVS_FMA_MIX_LO
namespace mvs {
__device__ __forceinline__ float fma_mix_lo(unsigned h2, float w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(unsigned h2, float w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(acc));
    return r;
}
}  // namespace mvs
#define MVS_FMA_MIX_LO(h2, w, acc) mvs::fma_mix_lo(h2, w, acc)
#define MVS_FMA_MIX_HI(h2, w, acc) mvs::fma_mix_hi(h2, w, acc)
#endif

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- fp16 saturation counter (round 4) ------------------------------------------------------------------------
// Every kernel that stores fp16 ACTIVATIONS (MVS_PREC_F16X2) clamps to +-65504; a clamp that really fires means the default format
// degraded a value that the fp32-equivalent format (bf16x3) would have kept.  Each such kernel keeps the running maximum |value| it
// stored per work-item (sat_track: two v_max3 per four values, no branch in the epilogue loops) and adds one count per work-item that
// saw a value beyond the range when it ends (sat_commit).  The counter is one device global PER TRANSLATION UNIT (no relocatable device
// code); mvs_f16_saturation_count() (capi.hip) sums the units' readers.
namespace sat {
static __device__ unsigned int counter;
__device__ __forceinline__ void track(float& amax, float a, float b, float c, float d) {
    amax = fmaxf(fmaxf(amax, fabsf(a)), fabsf(b));
    amax = fmaxf(fmaxf(amax, fabsf(c)), fabsf(d));
}
__device__ __forceinline__ void commit(float amax) {
    if (amax > 65504.0f) atomicAdd(&counter, 1u);
}
}  // namespace sat
// host: read (and optionally reset) this translation unit's counter on the current device
#define MVS_DEFINE_SAT_READER(fn)                                                        \
    unsigned int fn(int reset) {                                                         \
        unsigned int v = 0;                                                              \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(mvs::sat::counter), sizeof(v)) != hipSuccess) return 0; \
        if (reset && v) { const unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(mvs::sat::counter), &z, sizeof(z)); } \
        return v;                                                                        \
    }

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kMaxSrcViews = 16;   // source views handled by one aggregate launch

// ---- error plumbing (host) ----
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- feature element access (device) ----
template <int DT> struct FeatT;
template <> struct FeatT<MVS_DTYPE_F32> { typedef float type; };
template <> struct FeatT<MVS_DTYPE_BF16> { typedef uint16_t type; };
template <> struct FeatT<MVS_DTYPE_F16> { typedef _Float16 type; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(uint16_t bf16_bits) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)bf16_bits) << 16;
    return c.f;
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }
template <> __device__ __forceinline__ uint16_t from_f32<uint16_t>(float v) {      // bf16 bits, round to nearest even (NaN kept quiet)
    union { float f; uint32_t u; } c;
    c.f = v;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((c.u >> 16) | 0x40u);
    return (uint16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}

// 3x4 homography of one source view: p = R * [x, y, 1] * depth + t   (warping.py:80-92)
struct Homography {
    float r[9];
    float t[3];
};

// Bilinear tap set of one projected point, ATen grid_sampler_2d semantics (bilinear, zeros padding,
// align_corners=True), reached through the same normalise / un-normalise round trip the reference
// performs (warping.py:93-95 then grid_sampler_unnormalize).
struct Taps {
    int off[4];      // y*W + x of nw, ne, sw, se (clamped to a valid address)
    float w[4];      // bilinear weight, 0 for an out-of-bounds tap
};

__device__ __forceinline__ Taps make_taps(const Homography& hm, float qx, float qy, float qz, float depth, int H, int W,
                                          float half_w, float half_h, bool* out_of_frame) {
    const float px = qx * depth + hm.t[0];
    const float py = qy * depth + hm.t[1];
    const float pz = qz * depth + hm.t[2];
    const float zz = pz + 1e-6f;
    const float u = px / zz;
    const float v = py / zz;
    const float xn = u / half_w - 1.0f;                  // warping.py:94
    const float yn = v / half_h - 1.0f;                  // warping.py:95
    if (out_of_frame) *out_of_frame = (xn > 1.0f) || (xn < -1.0f) || (yn > 1.0f) || (yn < -1.0f) || (pz <= 0.0f);
    const float ix = ((xn + 1.0f) / 2.0f) * (float)(W - 1);   // grid_sampler_unnormalize, align_corners
    const float iy = ((yn + 1.0f) / 2.0f) * (float)(H - 1);
    Taps tp;
    // Non-finite or far-away coordinates contribute nothing (every tap fails the bounds test in ATen).
    const bool sane = (ix > -2.0f) && (ix < (float)(W + 1)) && (iy > -2.0f) && (iy < (float)(H + 1));
    if (!sane) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { tp.off[k] = 0; tp.w[k] = 0.0f; }
        return tp;
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wy1 = iy - fy0;           // weight of the +1 neighbour
    const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x1 >= 0) && (x1 < W);
    const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y1 >= 0) && (y1 < H);
    const int cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : 0, cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : 0;
    tp.off[0] = cy0 * W + cx0; tp.w[0] = (vx0 && vy0) ? wx0 * wy0 : 0.0f;   // nw
    tp.off[1] = cy0 * W + cx1; tp.w[1] = (vx1 && vy0) ? wx1 * wy0 : 0.0f;   // ne
    tp.off[2] = cy1 * W + cx0; tp.w[2] = (vx0 && vy1) ? wx0 * wy1 : 0.0f;   // sw
    tp.off[3] = cy1 * W + cx1; tp.w[3] = (vx1 && vy1) ? wx1 * wy1 : 0.0f;   // se
    return tp;
}

// The same four taps expressed as TWO horizontally adjacent pairs (top row, bottom row): one unaligned
// 8-byte (fp32) / 4-byte (bf16, fp16) load per row instead of two 4-/2-byte loads.  `top`/`bot` address the
// LEFT element of an in-bounds pair (x clamped to [0, W-2]); the bilinear weights are routed to whichever pair
// slot the valid tap landed in, so border handling (per-tap zero padding) is bit-identical to make_taps.
// Needs W >= 2.
struct PairTaps {
    int top, bot;
    float w00, w01, w10, w11;    // weights of top.x, top.y, bot.x, bot.y
};

// register class for "opaque value" asm constraints (the host emulator of the tests overrides it)
#ifndef MVS_OPAQUE_REG
#define MVS_OPAQUE_REG "v"
#endif

template <typename T> struct PairOf;
template <> struct PairOf<float> { struct __attribute__((packed, aligned(4))) type { float x, y; }; };
template <> struct PairOf<uint16_t> { struct __attribute__((packed, aligned(2))) type { uint16_t x, y; }; };
template <> struct PairOf<_Float16> { struct __attribute__((packed, aligned(2))) type { _Float16 x, y; }; };

__device__ __forceinline__ PairTaps make_pair_taps(const Homography& hm, float qx, float qy, float qz, float depth, int H, int W,
                                                   float half_w, float half_h) {
    const float px = qx * depth + hm.t[0];
    const float py = qy * depth + hm.t[1];
    const float pz = qz * depth + hm.t[2];
    const float zz = pz + 1e-6f;
    const float u = px / zz;
    const float v = py / zz;
    const float xn = u / half_w - 1.0f;
    const float yn = v / half_h - 1.0f;
    const float ix = ((xn + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((yn + 1.0f) / 2.0f) * (float)(H - 1);
    PairTaps tp;
    const bool sane = (ix > -2.0f) && (ix < (float)(W + 1)) && (iy > -2.0f) && (iy < (float)(H + 1));
    if (!sane) {
        tp.top = 0; tp.bot = 0; tp.w00 = 0.0f; tp.w01 = 0.0f; tp.w10 = 0.0f; tp.w11 = 0.0f;
        return tp;
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx1 = ix - fx0, wy1 = iy - fy0;
    const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x0 >= -1) && (x0 < W - 1);
    const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y0 >= -1) && (y0 < H - 1);
    const int xb = x0 < 0 ? 0 : (x0 > W - 2 ? W - 2 : x0);
    // slot 0 holds column xb, slot 1 column xb + 1
    const float wa = (vx0 && x0 <= W - 2 ? wx0 : 0.0f) + (vx1 && x0 < 0 ? wx1 : 0.0f);
    const float wb = (vx0 && x0 > W - 2 ? wx0 : 0.0f) + (vx1 && x0 >= 0 ? wx1 : 0.0f);
    const float wyt = vy0 ? wy0 : 0.0f, wyb = vy1 ? wy1 : 0.0f;
    const int yt = vy0 ? y0 : 0, yb = vy1 ? y0 + 1 : 0;
    tp.top = yt * W + xb;
    tp.bot = yb * W + xb;
    tp.w00 = wa * wyt; tp.w01 = wb * wyt; tp.w10 = wa * wyb; tp.w11 = wb * wyb;
    return tp;
}

constexpr unsigned GL_NONE = 0xffffffffu;

// Bilinear tap set as ONE 2x2 block of in-bounds source pixels: (xb, yb) = top-left corner clamped to
// [0, W-2] x [0, H-2], with the four bilinear weights routed to whichever block slot each valid tap landed in and zero
// for taps outside the image (ATen grid_sampler_2d, zeros padding).  pk = (yb << 16) | xb, GL_NONE when no tap is
// inside the image.  The sample position is the projected pixel itself: the reference's normalise (warping.py:94-95)
// and grid_sample's un-normalise cancel up to fp32 rounding (validated against the reference).
struct GTap {
    unsigned pk;
    float w00, w01, w10, w11;
};

__device__ __forceinline__ GTap make_gtap(const Homography& hm, float qx, float qy, float qz, float depth, int H, int W, float cx,
                                          float cy) {
    const float px = qx * depth + hm.t[0];                     // warping.py:90-92
    const float py = qy * depth + hm.t[1];
    const float pz = qz * depth + hm.t[2];
    const float zz = pz + 1e-6f;                               // warping.py:93
    float r = __builtin_amdgcn_rcpf(zz);
    r = fmaf(fmaf(-zz, r, 1.0f), r, r);                        // one Newton step: <= 1 ulp
    const float ix = px * r, iy = py * r;
    GTap tp;
    // -1 < ix < W and -1 < iy < H (at least one tap column and row inside the image); false for NaN / inf
    const bool sane = (fabsf(ix - cx) < cx + 1.0f) && (fabsf(iy - cy) < cy + 1.0f);
    if (!sane) {
        tp.pk = GL_NONE; tp.w00 = 0.0f; tp.w01 = 0.0f; tp.w10 = 0.0f; tp.w11 = 0.0f;
        return tp;
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;                    // in [-1, W-1] x [-1, H-1]
    const float wx1 = ix - fx0, wy1 = iy - fy0;
    const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    int xb = x0, yb = y0;
    float wa = wx0, wb = wx1, wt = wy0, wd = wy1;             // weights of column xb, xb + 1, row yb, yb + 1
    if ((unsigned)x0 > (unsigned)(W - 2) || (unsigned)y0 > (unsigned)(H - 2)) {   // rare: the 2x2 block touches the image border
        xb = x0 < 0 ? 0 : (x0 > W - 2 ? W - 2 : x0);
        yb = y0 < 0 ? 0 : (y0 > H - 2 ? H - 2 : y0);
        wa = x0 < 0 ? wx1 : (x0 > W - 2 ? 0.0f : wx0);
        wb = x0 < 0 ? 0.0f : (x0 > W - 2 ? wx0 : wx1);
        wt = y0 < 0 ? wy1 : (y0 > H - 2 ? 0.0f : wy0);
        wd = y0 < 0 ? 0.0f : (y0 > H - 2 ? wy0 : wy1);
    }
    tp.pk = ((unsigned)yb << 16) | (unsigned)xb;
    tp.w00 = wa * wt; tp.w01 = wb * wt; tp.w10 = wa * wd; tp.w11 = wb * wd;
    return tp;
}

// XCD-aware work-group remap: the dispatcher places block b on XCD b % 8, so consecutive logical tiles
// (which share feature rows / halo voxels) are handed to the same XCD and hit its private L2.
// Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, xcd = b & 7u, idx = b >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Start stagger of the first generation of workgroups (experiment switch MVS_STAGGER = units of ~0.43 us per wave slot, 0 = off).
// All workgroups of a launch start within a microsecond of each other and the co-resident ones of a CU then run their memory
// phase and their contraction phase IN STEP - the phases add up instead of overlapping - and each
// successor inherits its slot's timing.  Delaying the first workgroup of wave slot s of a SIMD by s units spreads the slots' phases.
#ifndef MVS_STAGGER
#define MVS_STAGGER 0
#endif
__device__ __forceinline__ void start_stagger(unsigned first_generation_blocks) {
#if MVS_STAGGER > 0
    if (blockIdx.x < first_generation_blocks) {
        const unsigned slot = (unsigned)__builtin_amdgcn_s_getreg(6148) & 15u;      // HW_REG_HW_ID, WAVE_ID: this wave's slot in its SIMD
        for (unsigned i = 0; i < slot * MVS_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);          // 16 x 64 cycles
    }
#endif
}

// Wave priority around the contraction phase of the one-tile MFMA kernels (experiment switch MVS_PRIO: 0 off, 1 contraction
// waves high, 2 staging / epilogue waves high).  Co-resident workgroups of a CU are at different phases; the arbiter picks by priority.
#ifndef MVS_PRIO
#define MVS_PRIO 0
#endif
__device__ __forceinline__ void prio_kernel_begin() {
#if MVS_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#endif
}
__device__ __forceinline__ void prio_contract_begin() {
#if MVS_PRIO == 1
    __builtin_amdgcn_s_setprio(3);
#elif MVS_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void prio_contract_end() {
#if MVS_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif MVS_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#endif
}

static inline unsigned ceil_div(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// a13 / init_range: hypothesis d of D between the first and the last of the [B,N] depth values           module.py:674-704
__device__ __forceinline__ float init_range_value(float first, float last, int inverse, int d, int D) {
    if (inverse) {
        const float inv_min = 1.0f / first, inv_max = 1.0f / last;
        const float itv = (float)d / (float)(D - 1);
        return 1.0f / (inv_max + (inv_min - inv_max) * itv);                        // module.py:697-704
    }
    const float interval = (last - first) / (float)(D - 1);
    return first + (float)d * interval;                                             // module.py:676-682
}

}  // namespace mvs
