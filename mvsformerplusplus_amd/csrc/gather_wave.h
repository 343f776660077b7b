// Wave-autonomous form of the two gather passes (homography warp + group-wise correlation),
// round 3.  Reference behaviour restated (never copied): models/warping.py:84-106 (projection, bilinear grid_sample with zeros
// padding / align_corners=True), models/cost_volume.py:74-101 (group correlation, softmax entropy, visibility-weighted
// aggregation).
//
// Round 2 (gather_lds_kernels.hip) gave a whole 256-thread workgroup one source window per unit: two workgroup barriers per
// unit (bounding box, window ready) plus one per channel octet, every wave waiting on the slowest, and the window's load latency
// exposed once per unit and octet - PMC: 48-57 % of the wave cycles in s_waitcnt / s_barrier, and the coarse stages (C = 32 / 64:
// eight serial octet rounds per unit) ran at 5-18 % of the HBM roofline for 39-160 MB of traffic.  Here every WAVE is its own
// master:
//   * a wave owns a small tile of reference pixels (16x4 / 8x4 / 4x4 / 4x2 for 1 / 2 / 4 / 8 work-items per pixel; one
//     work-item = one pixel x 4 consecutive depth planes) and a private 12 KiB LDS window - there is NO workgroup barrier in
//     the kernels (LDS operations of one wave execute in order, so a window written by the wave is visible to its own reads);
//   * the bounding box of a unit's taps is reduced with DPP inside the wave and read back with v_readlane: the window
//     geometry lives in scalar registers;
//   * the stream of (view, chunk group, channel octet) items is software-pipelined: the window loads of item k + 1 (sixteen-byte
//     loads of 4 consecutive positions per channel, straight from the planar NCHW map, or whole 32-byte runs of the octet-tiled
//     hand-off layout) and its reference features are in flight, in registers, while item k is gathered from LDS; the tap set
//     of the next unit is computed before the current unit's last gather (two tap sets alive);
//   * the window is committed as channel-interleaved quads win[quad][position] (ds_write_b128) and gathered with
//     ds_read_b128 exactly as in round 2.
// Price: a wave-sized tile has a larger halo than a workgroup-sized one (stage 4: ~0.75 staged positions per (pixel, plane)
// instead of 0.45) - L2 / texture-addresser traffic, not HBM.  A unit whose window exceeds the capacity falls back,
// wave-uniformly, to pair loads from global memory.
//
// Algorithmic HBM bytes per launch: pass 1 = features (1 + n_views) * C*HW*sizeof(T) + hypotheses
// D*HW*4 + entropy n_views*HW*4; pass 2 = the same inputs + visibility + G*D*HW*4 volume write.
#pragma once
#include "gather_common.h"

namespace mvs {

#ifndef MVS_OPAQUE_SREG
#define MVS_OPAQUE_SREG "s"
#endif
#ifndef MVS_GW_PLANES_IN_FLIGHT
#define MVS_GW_PLANES_IN_FLIGHT 1     // planes whose taps (8 ds_read_b128 results each) are in flight in the gather
#endif
#ifndef MVS_GW_XUNIT
#define MVS_GW_XUNIT 0                // 1: the next unit's tap set and window loads are requested before the current unit's last gather
#endif                                // (two tap sets + the staged item alive across a gather: ~200 VGPRs); 0: only the octets of a unit are pipelined
#ifndef MVS_GW_DBG
#define MVS_GW_DBG 0
#endif
#ifndef MVS_GW_CAP
#define MVS_GW_CAP 384
#endif
constexpr int GW_CAP = MVS_GW_CAP;   // window capacity in source positions per wave: 2 quads * GW_CAP * 16 B = 12 KiB
constexpr int GW_DCH = 4;            // depth planes per work-item
static_assert(GW_CAP % 4 == 0 && GW_CAP >= 256 && GW_CAP <= 512, "a window is staged in at most two rounds of 64 lane tasks of 4 positions");

// NSW work-items ("slots") share a pixel, one per chunk of GW_DCH planes; a wave = NP pixels = a PW x PH tile; a workgroup = four
// waves side by side (they share nothing but the launch).
template <int NSW>
struct GwTile {
    static constexpr int NP = 64 / NSW;
    static constexpr int PW = NP >= 64 ? 16 : (NP >= 32 ? 8 : 4), PH = NP / PW;
    static constexpr int BW = 4 * PW;
};

// wave-level LDS hand-over: the hardware executes one wave's LDS operations in order; the compiler must not move them across
// this point, and the host emulator (lanes are fibers) needs the rendezvous
__device__ __forceinline__ void gw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tap set + window geometry of one unit (one source view x the GW_DCH planes of every work-item of the wave)
struct GwUnit {
    unsigned pk[GW_DCH];        // (yb << 16) | xb of the 2x2 block, GL_NONE when no tap is inside the image
    unsigned pos[GW_DCH];       // window position of the block's top-left corner
    float w[GW_DCH][4];
    // wave-uniform:
    int ww;                     // window width in positions (multiple of 8)
    int n4;                     // lane tasks of 4 positions (n / 4); 0: no tap of the wave is inside the image; -1: oversize (fallback)
    unsigned gbase;             // ymin * W + wx0
    float inv_ww4;
};

template <typename T> struct GwVec { typedef T type __attribute__((ext_vector_type(4))); };

// Feature reads go through a buffer descriptor per view (cdna_hip_programming.md T8): ONE 32-bit per-lane offset register serves
// all 8 channel loads of an item (the channel plane is the scalar offset), where flat loads need a 64-bit address pair each; reads
// beyond the view (idle lanes of a staging round) return zero instead of faulting, so the common path has no exec-mask branches.
typedef __amdgpu_buffer_rsrc_t gw_rsrc;
__device__ __forceinline__ gw_rsrc gw_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ typename GwVec<T>::type gw_buf_load4(gw_rsrc rs, unsigned voff, unsigned soff) {
    typedef typename GwVec<T>::type V4;
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
    else return __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
}
template <typename T>
__device__ __forceinline__ T gw_buf_load1(gw_rsrc rs, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, (int)soff, 0));
}

// registers of one staged item while its loads are in flight
template <typename T>
struct GwStage {
    typename GwVec<T>::type r[8];
    T f[8];                              // the pixel's reference features of the octet (raw)
};

__device__ __forceinline__ void gw_prepare(GwUnit& u, const Homography& hm, float fx, float fy, const float* depth, bool active, int H, int W) {
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];     // warping.py:90 (once per pixel and view)
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    u16x2 mn = {0xffff, 0xffff}, mx = {0, 0};
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
#pragma unroll
    for (int dd = 0; dd < GW_DCH; ++dd) {
        const GTap tp = make_gtap(hm, qx, qy, qz, depth[dd], H, W, cx, cy);
        u.pk[dd] = active ? tp.pk : GL_NONE;
        u.w[dd][0] = tp.w00; u.w[dd][1] = tp.w01; u.w[dd][2] = tp.w10; u.w[dd][3] = tp.w11;
        if (u.pk[dd] != GL_NONE) {
            mn = __builtin_elementwise_min(mn, gl_as_vec(tp.pk));
            mx = __builtin_elementwise_max(mx, gl_as_vec(tp.pk));
        }
    }
    mn = gl_wave_reduce<false>(mn);                               // lane 63 holds the wave's result
    mx = gl_wave_reduce<true>(mx);
    const unsigned smn = (unsigned)__builtin_amdgcn_readlane((int)gl_as_u32(mn), 63);
    const unsigned smx = (unsigned)__builtin_amdgcn_readlane((int)gl_as_u32(mx), 63);
    const int xmin = (int)(smn & 0xffffu), ymin = (int)(smn >> 16), xmax = (int)(smx & 0xffffu), ymax = (int)(smx >> 16);
    if (xmax < xmin) {                                          // no tap of the whole wave is inside the source image
        u.ww = 8; u.n4 = 0; u.gbase = 0; u.inv_ww4 = 0.5f;
#pragma unroll
        for (int dd = 0; dd < GW_DCH; ++dd) u.pos[dd] = 0;
        return;
    }
    const int wx0 = xmin & ~7;
    const int ww = (xmax + 2 - wx0 + 7) & ~7;
    const int wh = ymax + 2 - ymin;
    const int n = ww * wh;
    u.ww = ww;
    u.n4 = n <= GW_CAP ? n >> 2 : -1;
    u.gbase = (unsigned)ymin * (unsigned)W + (unsigned)wx0;
    u.inv_ww4 = __builtin_amdgcn_rcpf((float)(ww >> 2)) * 1.000001f;      // row = floor((j + 0.5) / ww4): exact for j < 2^16
#pragma unroll
    for (int dd = 0; dd < GW_DCH; ++dd) {
        const unsigned pk = u.pk[dd];
        u.pos[dd] = pk == GL_NONE ? 0u : ((pk >> 16) - (unsigned)ymin) * (unsigned)ww + ((pk & 0xffffu) - (unsigned)wx0);
    }
}

// window loads of lane task j (positions 4j .. 4j+3 of the row-major window) of the octet at element offset `oofs` of the view
template <typename T, bool TILED>
__device__ __forceinline__ void gw_load_task(const GwUnit& u, gw_rsrc rs, unsigned oofs, unsigned HW, int W, int j, typename GwVec<T>::type* r) {
    const int row = (int)(((float)j + 0.5f) * u.inv_ww4);
    const unsigned g = u.gbase + (unsigned)row * (unsigned)(W - u.ww) + 4u * (unsigned)j;      // (ymin + row) * W + wx0 + 4 * (j - row * ww4)
    if (!TILED) {
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = gw_buf_load4<T>(rs, g * (unsigned)sizeof(T), (oofs + (unsigned)c * HW) * (unsigned)sizeof(T));
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = gw_buf_load4<T>(rs, (g * 8u + 4u * k) * (unsigned)sizeof(T), oofs * (unsigned)sizeof(T));
    }
}

// ... and their commit to the wave's window as channel-interleaved quads win[quad][position]
template <typename T, bool TILED>
__device__ __forceinline__ void gw_store_task(f32x4* win, int j, const typename GwVec<T>::type* r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 lo, hi;
        if (!TILED) {
            lo = f32x4{to_f32((T)r[0][k]), to_f32((T)r[1][k]), to_f32((T)r[2][k]), to_f32((T)r[3][k])};
            hi = f32x4{to_f32((T)r[4][k]), to_f32((T)r[5][k]), to_f32((T)r[6][k]), to_f32((T)r[7][k])};
        } else {
            lo = f32x4{to_f32((T)r[2 * k][0]), to_f32((T)r[2 * k][1]), to_f32((T)r[2 * k][2]), to_f32((T)r[2 * k][3])};
            hi = f32x4{to_f32((T)r[2 * k + 1][0]), to_f32((T)r[2 * k + 1][1]), to_f32((T)r[2 * k + 1][2]), to_f32((T)r[2 * k + 1][3])};
        }
        win[4 * j + k] = lo;
        win[GW_CAP + 4 * j + k] = hi;
    }
}

// the pixel's reference features of the octet at element offset `oofs`, raw
template <typename T, bool TILED>
__device__ __forceinline__ void gw_load_ref(gw_rsrc rr, unsigned oofs, unsigned HW, unsigned pc, T* f) {
    if (!TILED) {
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = gw_buf_load1<T>(rr, pc * (unsigned)sizeof(T), (oofs + (unsigned)c * HW) * (unsigned)sizeof(T));
    } else {
        const typename GwVec<T>::type a = gw_buf_load4<T>(rr, pc * 8u * (unsigned)sizeof(T), oofs * (unsigned)sizeof(T));
        const typename GwVec<T>::type b = gw_buf_load4<T>(rr, (pc * 8u + 4u) * (unsigned)sizeof(T), oofs * (unsigned)sizeof(T));
#pragma unroll
        for (int c = 0; c < 4; ++c) { f[c] = a[c]; f[4 + c] = b[c]; }
    }
}

// request one item: the pixel's reference features and the first 64 lane tasks of the window (tasks beyond the window read
// in-range garbage or, beyond the view, zeros; their window slots are never gathered)
template <typename T, bool TILED>
__device__ __forceinline__ void gw_issue(const GwUnit& u, gw_rsrc rs, gw_rsrc rr, unsigned oofs, unsigned HW, unsigned pc, int W, int lane, GwStage<T>& s) {
    gw_load_ref<T, TILED>(rr, oofs, HW, pc, s.f);
    gw_load_task<T, TILED>(u, rs, oofs, HW, W, lane, s.r);
}

// commit the item to the wave's window (every earlier gather of the wave has been issued: in-order LDS)
template <typename T, bool TILED>
__device__ __forceinline__ void gw_commit(const GwUnit& u, gw_rsrc rs, unsigned oofs, unsigned HW, int W, int lane, f32x4* win, GwStage<T>& s) {
    gw_wave_sync();
    gw_store_task<T, TILED>(win, lane, s.r);                    // positions 0 .. 255 < GW_CAP: no guard
    if (u.n4 > 64) {                                            // wave-uniform, rare: windows of more than 256 positions
        typename GwVec<T>::type r2[8];
        if (lane + 64 < u.n4) {
            gw_load_task<T, TILED>(u, rs, oofs, HW, W, lane + 64, r2);
            gw_store_task<T, TILED>(win, lane + 64, r2);
        }
    }
    gw_wave_sync();
}

// One item = one channel octet of one unit, gathered from the wave's window:
//   KEEP_GROUPS = false: out[dd]                    += sum_c rf[c] * warped[c, d]                 (pass 1)
//   KEEP_GROUPS = true : out[(O*GPO + j)*DCH + dd]  += sum_{c in group} rf[c] * warped[c, d]      (pass 2)
// rf = the reference features of the octet, already scaled.
template <int NOCT, bool KEEP_GROUPS, int O>
__device__ __forceinline__ void gw_gather(const GwUnit& u, const f32x4* win, const float* rf, float* out) {
    constexpr int GPO = 8 / NOCT, CPG = NOCT;
    const int ww = u.ww;
#pragma unroll
    for (int dd = 0; dd < GW_DCH; ++dd) {
        if (dd % MVS_GW_PLANES_IN_FLIGHT == 0) __builtin_amdgcn_sched_barrier(0);
        const f32x4* w0 = win + u.pos[dd];
        const f32x4 a0 = w0[0], a1 = w0[1], b0 = w0[ww], b1 = w0[ww + 1];
        const f32x4 c0 = w0[GW_CAP], c1 = w0[GW_CAP + 1], d0 = w0[GW_CAP + ww], d1 = w0[GW_CAP + ww + 1];
        f32x4 lo = a0 * u.w[dd][0];
        lo += a1 * u.w[dd][1];
        lo += b0 * u.w[dd][2];
        lo += b1 * u.w[dd][3];
        f32x4 hi = c0 * u.w[dd][0];
        hi += c1 * u.w[dd][1];
        hi += d0 * u.w[dd][2];
        hi += d1 * u.w[dd][3];
        const float wv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (KEEP_GROUPS) {
#pragma unroll
            for (int j = 0; j < GPO; ++j) {
                float s = out[(O * GPO + j) * GW_DCH + dd];
#pragma unroll
                for (int c = 0; c < CPG; ++c) s += rf[j * CPG + c] * wv[j * CPG + c];
                out[(O * GPO + j) * GW_DCH + dd] = s;
            }
        } else {
            float s = out[dd];
#pragma unroll
            for (int c = 0; c < 8; ++c) s += rf[c] * wv[c];
            out[dd] = s;
        }
    }
}

// oversize window: the whole unit (all octets) straight from global memory with pair loads, one channel at a time (rare path)
template <typename T, int NOCT, bool KEEP_GROUPS, bool TILED>
__device__ __forceinline__ void gw_fallback_unit(const GwUnit& u, const T* __restrict__ src, const T* __restrict__ ref, unsigned HW, unsigned pc, int W,
                                                 float wscale, float* out) {
    typedef typename PairOf<T>::type P2;
    constexpr int CPG = NOCT;
    unsigned top[GW_DCH];
#pragma unroll
    for (int dd = 0; dd < GW_DCH; ++dd) top[dd] = u.pk[dd] == GL_NONE ? 0u : (u.pk[dd] >> 16) * (unsigned)W + (u.pk[dd] & 0xffffu);
#pragma unroll 1
    for (int c = 0; c < 8 * NOCT; ++c) {
        const unsigned cbase = TILED ? (unsigned)(c >> 3) * 8u * HW + (unsigned)(c & 7) : (unsigned)c * HW;
        const unsigned pstep = TILED ? 8u : 1u;
        const float rfc = to_f32(ref[cbase + pc * pstep]) * wscale;
        const T* sp = src + cbase;
        const int g = c / CPG;
#pragma unroll
        for (int dd = 0; dd < GW_DCH; ++dd) {
            float t0, t1, b0, b1;
            if (TILED) {
                __builtin_amdgcn_sched_barrier(0);              // one plane's four taps in flight: the addresses are the register hog
                t0 = to_f32(sp[top[dd] * 8u]); t1 = to_f32(sp[(top[dd] + 1u) * 8u]);
                b0 = to_f32(sp[(top[dd] + (unsigned)W) * 8u]); b1 = to_f32(sp[(top[dd] + (unsigned)W + 1u) * 8u]);
            } else {
                const P2 t = *reinterpret_cast<const P2*>(sp + top[dd]);
                const P2 b = *reinterpret_cast<const P2*>(sp + top[dd] + (unsigned)W);
                t0 = to_f32(t.x); t1 = to_f32(t.y); b0 = to_f32(b.x); b1 = to_f32(b.y);
            }
            float wv = u.w[dd][0] * t0;
            wv += u.w[dd][1] * t1;
            wv += u.w[dd][2] * b0;
            wv += u.w[dd][3] * b1;
            if (KEEP_GROUPS) {
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) out[gg * GW_DCH + dd] += (g == gg) ? rfc * wv : 0.0f;
            } else {
                out[dd] += rfc * wv;
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ void gw_scaled_ref(const GwStage<T>& s, float scale, float* rf) {
#pragma unroll
    for (int c = 0; c < 8; ++c) rf[c] = to_f32(s.f[c]) * scale;
}

// the octet loop of one unit, pipelined: on entry the loads of (cur, octet 0) are in flight in `st`; on exit - if `have_next` -
// those of (nxt, octet 0) are.  `prep_next` computes the next unit's tap set and descriptor (called once, before the last gather).
template <typename T, int NOCT, bool KEEP_GROUPS, bool TILED, int O, class PrepNext>
__device__ __forceinline__ void gw_unit_octets(const GwUnit& cur, gw_rsrc rs, gw_rsrc rr, GwUnit& nxt, const T*& src_next, unsigned vbytes, bool have_next,
                                               PrepNext&& prep_next, unsigned HW, unsigned pc, int W, int lane, f32x4* win, GwStage<T>& st, float scale, float* out) {
    if constexpr (O < NOCT) {
        float rf[8];
        // The feature loads read `const` memory: left alone, the compiler hoists the loads of every later octet of the (unrolled)
        // loop to the top.  The octet offsets are laundered through empty asm statements - new values the loads depend on - and
        // scheduling barriers fence the phases.
        unsigned ofs_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)gl_octet_offset(O, HW));
        unsigned ofs_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)gl_octet_offset(O + 1 < NOCT ? O + 1 : O, HW));
        asm volatile("" : "+" MVS_OPAQUE_SREG(ofs_cur), "+" MVS_OPAQUE_SREG(ofs_nxt));
        if (cur.n4 > 0) gw_commit<T, TILED>(cur, rs, ofs_cur, HW, W, lane, win, st);
        gw_scaled_ref<T>(st, scale, rf);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (O + 1 < NOCT) {
            if (cur.n4 > 0) gw_issue<T, TILED>(cur, rs, rr, ofs_nxt, HW, pc, W, lane, st);
        } else {
            if (have_next) {
                prep_next();                                    // tap set, window geometry, descriptor and weight of the next unit
                if (nxt.n4 > 0) gw_issue<T, TILED>(nxt, gw_make_rsrc(src_next, vbytes), rr, 0u, HW, pc, W, lane, st);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cur.n4 > 0) gw_gather<NOCT, KEEP_GROUPS, O>(cur, win, rf, out);
        __builtin_amdgcn_sched_barrier(0);
        gw_unit_octets<T, NOCT, KEEP_GROUPS, TILED, O + 1>(cur, rs, rr, nxt, src_next, vbytes, have_next, prep_next, HW, pc, W, lane, win, st, scale, out);
    }
}

constexpr size_t GW_WIN_BYTES = (size_t)2 * GW_CAP * 16;

// ------------------------------------------------------------------------------------------------
// pass 1: entropy of the depth-softmax of the group-summed correlation          cost_volume.py:79-92
// grid = (tiles, ceil(views in launch / vpb), B); a wave walks `vpb` consecutive source views of its tile
// dynamic LDS per wave: [ window ][ sim: D * NP floats (only when the planes of a pixel are spread over lanes / iterations) ]
// ------------------------------------------------------------------------------------------------
template <int DT, int NOCT, int NSW, bool TILED>
__global__ __launch_bounds__(256) void gw_entropy_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                         const float* __restrict__ hyp, float* __restrict__ entropy, int V, int D, int H,
                                                         int W, int view_begin, int view_end, int vpb, int ntx, int nblk, int wave_lds) {
    typedef typename FeatT<DT>::type T;
    typedef GwTile<NSW> Tile;
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int C = 8 * NOCT, NP = Tile::NP;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4* win = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + (size_t)wave * wave_lds);
    float* sim = reinterpret_cast<float*>(reinterpret_cast<char*>(win) + GW_WIN_BYTES);
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int b = (int)blockIdx.z;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int ty = blk / ntx, tx = blk - ty * ntx;
    const int pix = lane % NP, slot = lane / NP;
    const int x = tx * Tile::BW + wave * Tile::PW + pix % Tile::PW, y = ty * Tile::PH + pix / Tile::PW;
    const bool valid = x < W && y < H;
    const unsigned pc = valid ? (unsigned)y * (unsigned)W + (unsigned)x : HW - 1u;
    const float fx = (float)(valid ? x : W - 1), fy = (float)(valid ? y : H - 1);
    const int nch = (D + GW_DCH - 1) / GW_DCH, niter = (nch + NSW - 1) / NSW;
    const T* feat = reinterpret_cast<const T*>(feat_) + (size_t)(b * V) * C * HW;
    const T* ref = feat;
    const unsigned vbytes = (unsigned)C * HW * (unsigned)sizeof(T);          // one view (gw_supported: < 4 GiB)
    const gw_rsrc rr = gw_make_rsrc(ref, vbytes);
    const float* hp = hyp + (size_t)b * D * HW;
    const float inv_cpg = 1.0f / (float)NOCT;                   // sum_g mean_c = (1 / cpg) sum_c
    const int v0 = view_begin + (int)blockIdx.y * vpb, v1 = v0 + vpb < view_end ? v0 + vpb : view_end;
    const int nu = (v1 - v0) * niter;                           // units of this wave: (view, chunk group), view-major
    if (nu <= 0) return;
    const bool direct = NSW == 1 && niter == 1;                 // every work-item owns all D <= 4 planes of its pixel

    GwUnit ua, ub;
    GwStage<T> st;
    float depth[GW_DCH];
    int cur_it = -1;
    // tap set of unit k into `u`; returns its source view pointer
    auto prepare = [&](GwUnit& u, int k) __attribute__((always_inline)) -> const T* {
        const int v = v0 + k / niter, it = k - (k / niter) * niter;
        const int chunk = it * NSW + slot;
        if (it != cur_it) {                                     // the hypotheses change with the chunk group only
            const int d0 = (chunk < nch ? chunk : nch - 1) * GW_DCH;
#pragma unroll
            for (int dd = 0; dd < GW_DCH; ++dd) depth[dd] = hp[(unsigned)(d0 + dd < D ? d0 + dd : D - 1) * HW + pc];
            cur_it = it;
        }
        const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
        gw_prepare(u, hm, fx, fy, depth, valid && chunk < nch, H, W);
        return feat + (size_t)v * C * HW;
    };
    const T* src_a = prepare(ua, 0);
    const T* src_b = src_a;
    if (ua.n4 > 0) gw_issue<T, TILED>(ua, gw_make_rsrc(src_a, vbytes), rr, 0u, HW, pc, W, lane, st);

    auto finish = [&](const GwUnit& u, const T* src, int k, float* s) __attribute__((always_inline)) {
        const int v = v0 + k / niter, it = k - (k / niter) * niter;
        const int chunk = it * NSW + slot;
        float* dst = entropy + (size_t)(b * (V - 1) + (v - 1)) * HW + pc;
        if (direct) {
            const float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            float m = s[0];
#pragma unroll
            for (int dd = 1; dd < GW_DCH; ++dd) m = dd < D ? fmaxf(m, s[dd]) : m;
            float e[GW_DCH], den = 0.0f;
#pragma unroll
            for (int dd = 0; dd < GW_DCH; ++dd) { e[dd] = dd < D ? __builtin_amdgcn_exp2f((s[dd] - m) * LOG2E) : 0.0f; den += e[dd]; }
            const float rden = 1.0f / den;
            float ent = 0.0f;
#pragma unroll
            for (int dd = 0; dd < GW_DCH; ++dd) {
                const float pr = e[dd] * rden;
                if (dd < D) ent -= pr * (__builtin_amdgcn_logf(pr + 1e-7f) * LN2);                  // cost_volume.py:92
            }
            if (valid) *dst = ent;
            return;
        }
        if (chunk < nch) {
            const int d0 = chunk * GW_DCH;
#pragma unroll
            for (int dd = 0; dd < GW_DCH; ++dd)
                if (d0 + dd < D) sim[(d0 + dd) * NP + pix] = s[dd];
        }
        if (it == niter - 1) {
            gw_wave_sync();
            if (slot == 0 && valid) gl_softmax_entropy_store(sim + pix, NP, D, dst);
            gw_wave_sync();                                     // the next view's sims overwrite these
        }
    };
    for (int k = 0; k < nu; ++k) {
        float s[GW_DCH];
#pragma unroll
        for (int dd = 0; dd < GW_DCH; ++dd) s[dd] = 0.0f;
        const bool have_next = MVS_GW_XUNIT && k + 1 < nu;
        auto prep_next = [&]() __attribute__((always_inline)) { src_b = prepare(ub, k + 1); };
        if (!MVS_GW_XUNIT && k > 0) {
            src_a = prepare(ua, k);
            if (ua.n4 > 0) gw_issue<T, TILED>(ua, gw_make_rsrc(src_a, vbytes), rr, 0u, HW, pc, W, lane, st);
        }
        if (ua.n4 >= 0) {
            gw_unit_octets<T, NOCT, false, TILED, 0>(ua, gw_make_rsrc(src_a, vbytes), rr, ub, src_b, vbytes, have_next, prep_next, HW, pc, W, lane, win, st, inv_cpg, s);
        } else {
            if (MVS_GW_DBG != 1) gw_fallback_unit<T, NOCT, false, TILED>(ua, src_a, ref, HW, pc, W, inv_cpg, s);
            if (have_next) {
                prep_next();
                if (ub.n4 > 0) gw_issue<T, TILED>(ub, gw_make_rsrc(src_b, vbytes), rr, 0u, HW, pc, W, lane, st);
            }
        }
        finish(ua, src_a, k, s);
        if (MVS_GW_XUNIT) {
            ua = ub;                                            // register moves: two tap sets are alive only across the last gather
            src_a = src_b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: visibility-weighted aggregation over the source views of the launch    cost_volume.py:97-101
// grid = (tiles, chunk groups, B); output channel-last [D,HW,8].
// ------------------------------------------------------------------------------------------------
template <int DT, int NOCT, int NSW, bool TILED>
__global__ __launch_bounds__(256) void gw_aggregate_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                           const float* __restrict__ hyp, const float* __restrict__ vis,
                                                           float* __restrict__ vol, float* __restrict__ vis_sum, int normalise, int V,
                                                           int D, int H, int W, int view_begin, int view_end, int ntx, int nblk) {
    typedef typename FeatT<DT>::type T;
    typedef GwTile<NSW> Tile;
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int C = 8 * NOCT, NP = Tile::NP;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4* win = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + (size_t)wave * GW_WIN_BYTES);
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int b = (int)blockIdx.z, it = (int)blockIdx.y;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int ty = blk / ntx, tx = blk - ty * ntx;
    const int pix = lane % NP, slot = lane / NP;
    const int x = tx * Tile::BW + wave * Tile::PW + pix % Tile::PW, y = ty * Tile::PH + pix / Tile::PW;
    const bool valid = x < W && y < H;
    const unsigned pc = valid ? (unsigned)y * (unsigned)W + (unsigned)x : HW - 1u;
    const float fx = (float)(valid ? x : W - 1), fy = (float)(valid ? y : H - 1);
    const int nch = (D + GW_DCH - 1) / GW_DCH;
    const T* feat = reinterpret_cast<const T*>(feat_) + (size_t)(b * V) * C * HW;
    const T* ref = feat;
    const unsigned vbytes = (unsigned)C * HW * (unsigned)sizeof(T);          // one view (gw_supported: < 4 GiB)
    const gw_rsrc rr = gw_make_rsrc(ref, vbytes);
    const float* hp = hyp + (size_t)b * D * HW;
    const float* vp = vis + (size_t)(b * (V - 1)) * HW + pc;
    const int nu = view_end - view_begin;
    if (nu <= 0) return;
    float vsum = 0.0f;
    for (int v = view_begin; v < view_end; ++v) vsum += vp[(unsigned)(v - 1) * HW];               // cost_volume.py:98
    if (vis_sum != nullptr && it == 0 && slot == 0 && valid) vis_sum[(size_t)b * HW + pc] = vsum;
    const float rdenom = normalise ? 1.0f / (vsum + 1e-6f) : 1.0f;                                // cost_volume.py:101
    const float inv_cpg = 1.0f / (float)NOCT;
    const int chunk = it * NSW + slot;
    const bool active = valid && chunk < nch;
    const int d0 = (chunk < nch ? chunk : nch - 1) * GW_DCH;
    float depth[GW_DCH];
#pragma unroll
    for (int dd = 0; dd < GW_DCH; ++dd) depth[dd] = hp[(unsigned)(d0 + dd < D ? d0 + dd : D - 1) * HW + pc];
    float acc[8 * GW_DCH];
#pragma unroll
    for (int i = 0; i < 8 * GW_DCH; ++i) acc[i] = 0.0f;

    GwUnit ua, ub;
    GwStage<T> st;
    float w_a, w_b;
    auto prepare = [&](GwUnit& u, int k, float& wk) __attribute__((always_inline)) -> const T* {
        const int v = view_begin + k;
        const Homography hm = gl_load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
        wk = vp[(unsigned)(v - 1) * HW];                                                          // cost_volume.py:97
        gw_prepare(u, hm, fx, fy, depth, active, H, W);
        return feat + (size_t)v * C * HW;
    };
    const T* src_a = prepare(ua, 0, w_a);
    const T* src_b = src_a;
    w_b = 0.0f;
    if (ua.n4 > 0) gw_issue<T, TILED>(ua, gw_make_rsrc(src_a, vbytes), rr, 0u, HW, pc, W, lane, st);

    for (int k = 0; k < nu; ++k) {
        const bool have_next = MVS_GW_XUNIT && k + 1 < nu;
        auto prep_next = [&]() __attribute__((always_inline)) { src_b = prepare(ub, k + 1, w_b); };
        if (!MVS_GW_XUNIT && k > 0) {
            src_a = prepare(ua, k, w_a);
            if (ua.n4 > 0) gw_issue<T, TILED>(ua, gw_make_rsrc(src_a, vbytes), rr, 0u, HW, pc, W, lane, st);
        }
        if (ua.n4 >= 0) {
            gw_unit_octets<T, NOCT, true, TILED, 0>(ua, gw_make_rsrc(src_a, vbytes), rr, ub, src_b, vbytes, have_next, prep_next, HW, pc, W, lane, win, st, inv_cpg * w_a, acc);
        } else {
            if (MVS_GW_DBG != 1) gw_fallback_unit<T, NOCT, true, TILED>(ua, src_a, ref, HW, pc, W, inv_cpg * w_a, acc);
            if (have_next) {
                prep_next();
                if (ub.n4 > 0) gw_issue<T, TILED>(ub, gw_make_rsrc(src_b, vbytes), rr, 0u, HW, pc, W, lane, st);
            }
        }
        if (MVS_GW_XUNIT) {
            ua = ub;
            src_a = src_b;
            w_a = w_b;
        }
    }
    if (active) {
        float* vb = vol + (size_t)b * D * HW * 8;
#pragma unroll
        for (int dd = 0; dd < GW_DCH; ++dd) {
            if (d0 + dd >= D) continue;
            float r[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) r[g] = acc[g * GW_DCH + dd] * rdenom;
            f32x4* o = reinterpret_cast<f32x4*>(vb + ((size_t)(unsigned)(d0 + dd) * HW + pc) * 8);
            o[0] = f32x4{r[0], r[1], r[2], r[3]};
            o[1] = f32x4{r[4], r[5], r[6], r[7]};
        }
    }
}

}  // namespace mvs
