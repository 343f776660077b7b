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
// to pair loads from global memory for that unit only.  Pass 2 gathers again instead of streaming correlation volumes
// kept by pass 1: the kept volumes cost 2 x 32 B per voxel and view of HBM traffic (1.85 GB per reference view at
// cfg2), the second gather re-reads the features from L2 / Infinity Cache.
//
// Algorithmic HBM bytes per launch: pass 1 = features (1 + n_views) * C*HW*sizeof(T) +
// hypotheses D*HW*4 + entropy n_views*HW*4; pass 2 = the same inputs + visibility + G*D*HW*4 volume write.
#include "mvs_common.h"

namespace mvs {

#ifndef MVS_GL_CAP
#define MVS_GL_CAP 1024
#endif
constexpr int GL_CAP = MVS_GL_CAP;   // window capacity in source positions: LDS = 2 quads * GL_CAP * 16 B = 32 KiB
constexpr int GL_XALIGN = 8;       // window x origin / width granularity in pixels (32 B of fp32, 16 B of bf16)
constexpr int GL_DCH = 4;          // depth planes per work-item
constexpr int GL_TH = 4;           // tile height in pixels
constexpr unsigned GL_NONE = 0xffffffffu;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// tile geometry for D hypotheses: `ns` work-items share a pixel (one per chunk of GL_DCH planes, looping when D > 4*ns)
struct GlGeo { int nch, ns, tp, tw, niter; };
__host__ __device__ inline GlGeo gl_geometry(int D) {
    GlGeo g;
    g.nch = (D + GL_DCH - 1) / GL_DCH;
    g.ns = g.nch >= 4 ? 4 : (g.nch >= 2 ? 2 : 1);
    g.tp = 256 / g.ns;
    g.tw = g.tp / GL_TH;
    g.niter = (g.nch + g.ns - 1) / g.ns;
    return g;
}

// Bilinear tap set as ONE 2x2 block of in-bounds source pixels: (xb, yb) = top-left corner clamped to
// [0, W-2] x [0, H-2], with the four bilinear weights routed to whichever block slot each valid tap landed in and zero
// for taps outside the image (ATen grid_sampler_2d, zeros padding).  pk = (yb << 16) | xb, GL_NONE when no tap is
// inside the image.  The sample position is the projected pixel itself: the reference's normalise (warping.py:94-95)
// and grid_sample's un-normalise cancel up to fp32 rounding (validated against the reference).
struct GTap {
    unsigned pk;
    float w00, w01, w10, w11;
};

__device__ __forceinline__ GTap make_gtap(const Homography& hm, float qx, float qy, float qz, float depth, int H, int W) {
    const float px = qx * depth + hm.t[0];                     // warping.py:90-92
    const float py = qy * depth + hm.t[1];
    const float pz = qz * depth + hm.t[2];
    const float zz = pz + 1e-6f;                               // warping.py:93
    float r = __builtin_amdgcn_rcpf(zz);
    r = fmaf(fmaf(-zz, r, 1.0f), r, r);                        // one Newton step: <= 1 ulp
    const float ix = px * r, iy = py * r;
    GTap tp;
    const bool sane = (ix > -1.0f) && (ix < (float)W) && (iy > -1.0f) && (iy < (float)H);    // false for NaN / inf
    if (!sane) {
        tp.pk = GL_NONE; tp.w00 = 0.0f; tp.w01 = 0.0f; tp.w10 = 0.0f; tp.w11 = 0.0f;
        return tp;
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;                    // in [-1, W-1] x [-1, H-1]
    const float wx1 = ix - fx0, wy1 = iy - fy0;
    const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    const int xb = x0 < 0 ? 0 : (x0 > W - 2 ? W - 2 : x0);
    const int yb = y0 < 0 ? 0 : (y0 > H - 2 ? H - 2 : y0);
    const float wa = x0 < 0 ? wx1 : (x0 > W - 2 ? 0.0f : wx0);       // weight of column xb
    const float wb = x0 < 0 ? 0.0f : (x0 > W - 2 ? wx0 : wx1);       // weight of column xb + 1
    const float wt = y0 < 0 ? wy1 : (y0 > H - 2 ? 0.0f : wy0);       // weight of row yb
    const float wd = y0 < 0 ? 0.0f : (y0 > H - 2 ? wy0 : wy1);       // weight of row yb + 1
    tp.pk = ((unsigned)yb << 16) | (unsigned)xb;
    tp.w00 = wa * wt; tp.w01 = wb * wt; tp.w10 = wa * wd; tp.w11 = wb * wd;
    return tp;
}

__device__ __forceinline__ u16x2 gl_as_vec(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned gl_as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

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
template <typename T, int NOCT, bool KEEP_GROUPS>
__device__ __forceinline__ void gl_unit(const T* __restrict__ src, const T* __restrict__ ref, const Homography& hm, float fx, float fy,
                                        const float* depth, bool active, int H, int W, unsigned HW, unsigned pc, f32x4* win,
                                        unsigned* red, int unit, float wscale, float* out) {
    typedef typename PairOf<T>::type P2;
    constexpr int GPO = 8 / NOCT;          // groups per octet
    constexpr int CPG = NOCT;              // channels per group
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];     // warping.py:90 (once per pixel and view)
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    GTap tp[GL_DCH];
    u16x2 mn = {0xffff, 0xffff}, mx = {0, 0};
#pragma unroll
    for (int dd = 0; dd < GL_DCH; ++dd) {
        tp[dd] = make_gtap(hm, qx, qy, qz, depth[dd], H, W);
        if (active && tp[dd].pk != GL_NONE) {
            mn = __builtin_elementwise_min(mn, gl_as_vec(tp[dd].pk));
            mx = __builtin_elementwise_max(mx, gl_as_vec(tp[dd].pk));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = __builtin_elementwise_min(mn, gl_as_vec(__shfl_xor(gl_as_u32(mn), m)));
        mx = __builtin_elementwise_max(mx, gl_as_vec(__shfl_xor(gl_as_u32(mx), m)));
    }
    unsigned* rd = red + (unit & 1) * 8;
    if (lane == 0) { rd[wave] = gl_as_u32(mn); rd[4 + wave] = gl_as_u32(mx); }
    __syncthreads();      // (A) bounding box complete; every thread has left the previous unit's gather
    mn = __builtin_elementwise_min(__builtin_elementwise_min(gl_as_vec(rd[0]), gl_as_vec(rd[1])),
                                   __builtin_elementwise_min(gl_as_vec(rd[2]), gl_as_vec(rd[3])));
    mx = __builtin_elementwise_max(__builtin_elementwise_max(gl_as_vec(rd[4]), gl_as_vec(rd[5])),
                                   __builtin_elementwise_max(gl_as_vec(rd[6]), gl_as_vec(rd[7])));
    const int xmin = mn[0], ymin = mn[1], xmax = mx[0], ymax = mx[1];
    if (xmax < xmin) return;                                    // no tap of the whole tile is inside the source image
    const int wx0 = xmin & ~(GL_XALIGN - 1);
    const int ww = (xmax + 2 - wx0 + GL_XALIGN - 1) & ~(GL_XALIGN - 1);
    const int wh = ymax + 2 - ymin;
    const int n = ww * wh;
    if (n <= GL_CAP) {
        unsigned pos[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) {
            const unsigned pk = tp[dd].pk;
            pos[dd] = pk == GL_NONE ? 0u : ((pk >> 16) - (unsigned)ymin) * (unsigned)ww + ((pk & 0xffffu) - (unsigned)wx0);
        }
        const float inv_ww = 1.0f / (float)ww;
        const unsigned gbase = (unsigned)ymin * (unsigned)W + (unsigned)wx0;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (o > 0) __syncthreads();                         // (C) the previous octet's taps have been read
            unsigned oofs = (unsigned)o * 8u * HW;
            asm volatile("" : "+" MVS_OPAQUE_SREG(oofs));
            const T* so = src + oofs;
            for (int i = tid; i < n; i += 256) {
                const int row = (int)(((float)i + 0.5f) * inv_ww);
                const unsigned g = gbase + (unsigned)row * (unsigned)(W - ww) + (unsigned)i;    // (ymin+row)*W + wx0 + (i - row*ww)
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = to_f32(so[(size_t)c * HW + g]);
                win[i] = f32x4{v[0], v[1], v[2], v[3]};
                win[GL_CAP + i] = f32x4{v[4], v[5], v[6], v[7]};
            }
            __syncthreads();                                    // (B) window of octet o is in LDS
            if (active) {
                unsigned rofs = (unsigned)o * 8u * HW + pc;
                asm volatile("" : "+" MVS_OPAQUE_REG(rofs));
                float rf[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) rf[c] = to_f32(ref[(size_t)c * HW + rofs]) * wscale;
#pragma unroll
                for (int dd = 0; dd < GL_DCH; ++dd) {
                    if (dd == 2) __builtin_amdgcn_sched_barrier(0);
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
                    const float wv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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
        for (int c = 0; c < 8 * NOCT; ++c) {                    // rolled: one channel's pair loads in flight (rare path)
            const unsigned plane = (unsigned)c * HW;
            const float rfc = to_f32(ref[plane + pc]) * wscale;
            const T* sp = src + plane;
            const int g = c / CPG;
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
                const P2 t = *reinterpret_cast<const P2*>(sp + top[dd]);
                const P2 b = *reinterpret_cast<const P2*>(sp + top[dd] + (unsigned)W);
                float wv = tp[dd].w00 * to_f32(t.x);
                wv += tp[dd].w01 * to_f32(t.y);
                wv += tp[dd].w10 * to_f32(b.x);
                wv += tp[dd].w11 * to_f32(b.y);
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

__device__ __forceinline__ void gl_softmax_entropy_store(const float* sim, int stride, int D, float* dst) {
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, sim[d * stride]);
    float den = 0.0f;
    for (int d = 0; d < D; ++d) den += expf(sim[d * stride] - m);
    float ent = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = expf(sim[d * stride] - m) / den;
        ent += -pr * logf(pr + 1e-7f);                          // cost_volume.py:92
    }
    *dst = ent;
}

__device__ __forceinline__ Homography gl_load_homography(const float* p) {
    Homography hm;
#pragma unroll
    for (int i = 0; i < 9; ++i) hm.r[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) hm.t[i] = p[9 + i];
    return hm;
}

// dynamic LDS: [ window: 2 * GL_CAP f32x4 ][ red: 16 u32 ][ sim: D * tp floats (pass 1 only) ]
constexpr size_t GL_WIN_BYTES = (size_t)2 * GL_CAP * 16;
constexpr size_t GL_RED_BYTES = 64;

// ------------------------------------------------------------------------------------------------
// pass 1: entropy of the depth-softmax of the group-summed correlation          cost_volume.py:79-92
// grid = (tiles, views in launch, B)
// ------------------------------------------------------------------------------------------------
template <int DT, int NOCT>
__global__ __launch_bounds__(256) void gl_entropy_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                         const float* __restrict__ hyp, float* __restrict__ entropy, int V, int D, int H,
                                                         int W, int view_begin, int ntx, int nblk) {
    typedef typename FeatT<DT>::type T;
    HIP_DYNAMIC_SHARED(float, smem)
    f32x4* win = reinterpret_cast<f32x4*>(smem);
    unsigned* red = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES);
    float* sim = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES + GL_RED_BYTES);
    constexpr int C = 8 * NOCT;
    const GlGeo geo = gl_geometry(D);
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int tid = (int)threadIdx.x;
    const int v = view_begin + (int)blockIdx.y, b = (int)blockIdx.z;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int ty = blk / ntx, tx = blk - ty * ntx;
    const int slot = tid / geo.tp, pi = tid - slot * geo.tp;
    const int py = pi / geo.tw, px = pi - py * geo.tw;
    const int x = tx * geo.tw + px, y = ty * GL_TH + py;
    const bool valid = x < W && y < H;
    const unsigned pc = valid ? (unsigned)y * (unsigned)W + (unsigned)x : HW - 1;
    const float fx = (float)(valid ? x : W - 1), fy = (float)(valid ? y : H - 1);
    const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
    const T* feat = reinterpret_cast<const T*>(feat_);
    const T* ref = feat + (size_t)(b * V) * C * HW;
    const T* src = feat + (size_t)(b * V + v) * C * HW;
    const float* hp = hyp + (size_t)b * D * HW + pc;
    const float inv_cpg = 1.0f / (float)NOCT;
    for (int it = 0; it < geo.niter; ++it) {
        const int chunk = it * geo.ns + slot;
        const bool active = valid && chunk < geo.nch;
        const int d0 = (chunk < geo.nch ? chunk : geo.nch - 1) * GL_DCH;
        float depth[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) depth[dd] = hp[(size_t)(d0 + dd < D ? d0 + dd : D - 1) * HW];
        float s[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) s[dd] = 0.0f;
        gl_unit<T, NOCT, false>(src, ref, hm, fx, fy, depth, active, H, W, HW, pc, win, red, it, inv_cpg, s);   // sum_g mean_c = (1/cpg) sum_c
        if (chunk < geo.nch) {
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd)
                if (d0 + dd < D) sim[(d0 + dd) * geo.tp + pi] = s[dd];
        }
    }
    __syncthreads();
    if (slot == 0 && valid) gl_softmax_entropy_store(sim + pi, geo.tp, D, entropy + (size_t)(b * (V - 1) + (v - 1)) * HW + pc);
}

// ------------------------------------------------------------------------------------------------
// pass 2: visibility-weighted aggregation over the source views of the launch    cost_volume.py:97-101
// grid = (tiles, 1, B); output channel-last [D,HW,8].
// ------------------------------------------------------------------------------------------------
template <int DT, int NOCT>
__global__ __launch_bounds__(256) void gl_aggregate_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                           const float* __restrict__ hyp, const float* __restrict__ vis,
                                                           float* __restrict__ vol, float* __restrict__ vis_sum, int normalise, int V,
                                                           int D, int H, int W, int view_begin, int view_end, int ntx, int nblk) {
    typedef typename FeatT<DT>::type T;
    HIP_DYNAMIC_SHARED(float, smem)
    f32x4* win = reinterpret_cast<f32x4*>(smem);
    unsigned* red = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + GL_WIN_BYTES);
    constexpr int C = 8 * NOCT;
    const GlGeo geo = gl_geometry(D);
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int tid = (int)threadIdx.x;
    const int b = (int)blockIdx.z;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int ty = blk / ntx, tx = blk - ty * ntx;
    const int slot = tid / geo.tp, pi = tid - slot * geo.tp;
    const int py = pi / geo.tw, px = pi - py * geo.tw;
    const int x = tx * geo.tw + px, y = ty * GL_TH + py;
    const bool valid = x < W && y < H;
    const unsigned pc = valid ? (unsigned)y * (unsigned)W + (unsigned)x : HW - 1;
    const float fx = (float)(valid ? x : W - 1), fy = (float)(valid ? y : H - 1);
    const T* feat = reinterpret_cast<const T*>(feat_);
    const T* ref = feat + (size_t)(b * V) * C * HW;
    const float* hp = hyp + (size_t)b * D * HW + pc;
    const float* vp = vis + (size_t)(b * (V - 1)) * HW + pc;
    float vsum = 0.0f;
    for (int v = view_begin; v < view_end; ++v) vsum += vp[(size_t)(v - 1) * HW];                 // cost_volume.py:98
    if (vis_sum != nullptr && slot == 0 && valid) vis_sum[(size_t)b * HW + pc] = vsum;
    const float denom = vsum + 1e-6f;                                                             // cost_volume.py:101
    const float inv_cpg = 1.0f / (float)NOCT;
    int unit = 0;
    for (int it = 0; it < geo.niter; ++it) {
        const int chunk = it * geo.ns + slot;
        const bool active = valid && chunk < geo.nch;
        const int d0 = (chunk < geo.nch ? chunk : geo.nch - 1) * GL_DCH;
        float depth[GL_DCH];
#pragma unroll
        for (int dd = 0; dd < GL_DCH; ++dd) depth[dd] = hp[(size_t)(d0 + dd < D ? d0 + dd : D - 1) * HW];
        float acc[8 * GL_DCH];
#pragma unroll
        for (int i = 0; i < 8 * GL_DCH; ++i) acc[i] = 0.0f;
        for (int v = view_begin; v < view_end; ++v, ++unit) {
            const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
            const float w = vp[(size_t)(v - 1) * HW];                                             // cost_volume.py:97
            gl_unit<T, NOCT, true>(feat + (size_t)(b * V + v) * C * HW, ref, hm, fx, fy, depth, active, H, W, HW, pc, win, red, unit,
                                   inv_cpg * w, acc);
        }
        if (active) {
#pragma unroll
            for (int dd = 0; dd < GL_DCH; ++dd) {
                if (d0 + dd >= D) continue;
                float r[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) r[g] = normalise ? acc[g * GL_DCH + dd] / denom : acc[g * GL_DCH + dd];
                f32x4* o = reinterpret_cast<f32x4*>(vol + ((size_t)(b * D + d0 + dd) * HW + pc) * 8);
                o[0] = f32x4{r[0], r[1], r[2], r[3]};
                o[1] = f32x4{r[4], r[5], r[6], r[7]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch (called from the C entry points in warp_kernels.hip)
// ------------------------------------------------------------------------------------------------
bool gl_supported(int C, int G, int D, int H, int W) {
    if (G != 8 || !(C == 8 || C == 16 || C == 32 || C == 64)) return false;
    if (W % GL_XALIGN != 0 || W < GL_XALIGN || H < 2 || W > 65535 || H > 65535) return false;
    const GlGeo geo = gl_geometry(D);
    return (size_t)D * geo.tp * sizeof(float) <= 64 * 1024;
}

template <int DT, int NOCT>
static int gl_launch_entropy_t(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int D, int H, int W, int vb,
                               int ve, hipStream_t st) {
    const GlGeo geo = gl_geometry(D);
    const int ntx = (int)ceil_div(W, geo.tw), nty = (int)ceil_div(H, GL_TH);
    const int nblk = ntx * nty;
    const size_t lds = GL_WIN_BYTES + GL_RED_BYTES + (size_t)D * geo.tp * sizeof(float);
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gl_entropy_kernel<DT, NOCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gl_entropy_kernel<DT, NOCT>), dim3(nblk, ve - vb, B), dim3(256), lds, st, feat, hom, hyp, ent, V, D, H, W, vb, ntx, nblk);
    return check_launch("gl_entropy_kernel");
}

template <int DT, int NOCT>
static int gl_launch_aggregate_t(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                                 int normalise, int B, int V, int D, int H, int W, int vb, int ve, hipStream_t st) {
    const GlGeo geo = gl_geometry(D);
    const int ntx = (int)ceil_div(W, geo.tw), nty = (int)ceil_div(H, GL_TH);
    const int nblk = ntx * nty;
    const size_t lds = GL_WIN_BYTES + GL_RED_BYTES;
    hipLaunchKernelGGL((gl_aggregate_kernel<DT, NOCT>), dim3(nblk, 1, B), dim3(256), lds, st, feat, hom, hyp, vis, vol, vis_sum, normalise, V,
                       D, H, W, vb, ve, ntx, nblk);
    return check_launch("gl_aggregate_kernel");
}

#define GL_DISPATCH(FN, ...)                                                                   \
    do {                                                                                       \
        switch (dtype * 4 + (C == 8 ? 0 : C == 16 ? 1 : C == 32 ? 2 : 3)) {                    \
            case 0: return FN<MVS_DTYPE_F32, 1>(__VA_ARGS__);                                  \
            case 1: return FN<MVS_DTYPE_F32, 2>(__VA_ARGS__);                                  \
            case 2: return FN<MVS_DTYPE_F32, 4>(__VA_ARGS__);                                  \
            case 3: return FN<MVS_DTYPE_F32, 8>(__VA_ARGS__);                                  \
            case 4: return FN<MVS_DTYPE_BF16, 1>(__VA_ARGS__);                                 \
            case 5: return FN<MVS_DTYPE_BF16, 2>(__VA_ARGS__);                                 \
            case 6: return FN<MVS_DTYPE_BF16, 4>(__VA_ARGS__);                                 \
            case 7: return FN<MVS_DTYPE_BF16, 8>(__VA_ARGS__);                                 \
            case 8: return FN<MVS_DTYPE_F16, 1>(__VA_ARGS__);                                  \
            case 9: return FN<MVS_DTYPE_F16, 2>(__VA_ARGS__);                                  \
            case 10: return FN<MVS_DTYPE_F16, 4>(__VA_ARGS__);                                 \
            default: return FN<MVS_DTYPE_F16, 8>(__VA_ARGS__);                                 \
        }                                                                                      \
    } while (0)

int gl_launch_entropy(const void* feat, int dtype, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H, int W,
                      int vb, int ve, hipStream_t st) {
    GL_DISPATCH(gl_launch_entropy_t, feat, hom, hyp, ent, B, V, D, H, W, vb, ve, st);
}

int gl_launch_aggregate(const void* feat, int dtype, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                        int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st) {
    GL_DISPATCH(gl_launch_aggregate_t, feat, hom, hyp, vis, vol, vis_sum, normalise, B, V, D, H, W, vb, ve, st);
}

}  // namespace mvs
