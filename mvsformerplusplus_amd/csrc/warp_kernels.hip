// Homography warp + group-wise correlation kernels.
//
// Reference behaviour restated (never copied): models/warping.py:69-109 builds a [B,C,D,H,W] warped
// volume with F.grid_sample, models/cost_volume.py:74-101 multiplies it with a D-times repeated
// reference volume, reduces channel groups, derives a per-view visibility weight from the softmax
// entropy over depth and accumulates the weighted volumes.  Here the warped volume is never
// materialised: every (pixel, depth) work-item gathers its 4 bilinear taps per channel straight from
// the NCHW source feature map (lanes = 64 consecutive pixels, so each tap load is a near-contiguous
// span), reduces the channel groups in registers and
//   pass 1 (warp_corr_entropy)    keeps only sum_g -> softmax_D -> entropy per pixel        [HW floats out]
//   pass 2 (warp_corr_aggregate)  recomputes the correlation for all source views of the launch,
//                                 weights by vis_v, sums over views, normalises and writes the
//                                 cost volume ONCE, channel-last [D,H,W,G].
// Re-gathering C*HW features is cheaper than writing + re-reading G*D*HW products per view.
//
// HBM algorithmic bytes per launch: features (1 + n_views) * C*HW*sizeof(T) + hypotheses D*HW*4
// (+ G*D*HW*4 volume write in pass 2); both kernels are gather/L1-bound, not MFMA work.
#include "mvs_common.h"

#include <stdlib.h>

namespace mvs {

// ------------------------------------------------------------------------------------------------
// a1 + warping.py:80: per-view homography  M = P_v @ inverse(P_0)
// ------------------------------------------------------------------------------------------------
__device__ void compose_p(const float* pm /*[2,4,4]*/, double* P /*16*/) {
    const float* E = pm;
    const float* K = pm + 16;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;                                    // fp32 like torch.matmul at cost_volume.py:69
            for (int k = 0; k < 3; ++k) s += K[i * 4 + k] * E[k * 4 + j];
            P[i * 4 + j] = (double)s;
        }
    for (int j = 0; j < 4; ++j) P[12 + j] = (double)E[12 + j];
}

// Gauss-Jordan with partial pivoting in fp64 (the reference inverts in fp32, warping.py:80; fp64 here is
// the exact value both a CPU and a GPU fp32 inverse approximate - they differ from it by <= 1e-4 px).
__device__ void invert4(const double* A, double* Ai) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { m[i][j] = A[i * 4 + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(m[c][c]);
        for (int r = c + 1; r < 4; ++r) if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); piv = r; }
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
        const double inv = 1.0 / m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = m[r][c];
            for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ai[i * 4 + j] = m[i][4 + j];
}

__device__ void homography_out(const double* Ps, const double* Pr, float* out) {
    double inv[16];
    invert4(Pr, inv);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += Ps[i * 4 + k] * inv[k * 4 + j];
            if (j < 3) out[i * 3 + j] = (float)s; else out[9 + i] = (float)s;
        }
    }
}

__global__ void compose_homography_kernel(const float* proj, int B, int V, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * (V - 1)) return;
    const int b = idx / (V - 1), v = 1 + idx % (V - 1);
    double Pr[16], Ps[16];
    compose_p(proj + (size_t)(b * V) * 32, Pr);
    compose_p(proj + (size_t)(b * V + v) * 32, Ps);
    homography_out(Ps, Pr, out + (size_t)idx * 12);
}

// Round 5: the cascade's prologue as ONE launch - the homographies of EVERY stage (the per-stage proj tensors differ only in the
// intrinsics' scale, general_eval.py:229-242) and stage 1's hypotheses (init_range / init_inverse_range, module.py:674-704): five launches of
// ~5 us each before (4 x compose_homography + init_range), all latency.  Blocks [0, init_blocks) fill the hypotheses, the last block
// composes: thread idx -> (stage, b, v).
struct ProloguePtrs {
    const float* proj[8];
    int n;
};

__global__ __launch_bounds__(256) void cascade_prologue_kernel(ProloguePtrs pp, int B, int V, float* __restrict__ hom, const float* __restrict__ dv,
                                                               int N, int inverse, float* __restrict__ hyp, int D, int HW, int init_blocks) {
    // grid = (init_blocks + 1, D or 1, B or 1): blocks x < init_blocks fill plane (b, d) of the hypotheses (grid-stride over HW, no divisions);
    // block (init_blocks, 0, 0) composes the homographies; the other x == init_blocks blocks have nothing to do
    if ((int)blockIdx.x < init_blocks) {
        const int b = (int)blockIdx.z, d = (int)blockIdx.y;
        const float v = init_range_value(dv[(size_t)b * N], dv[(size_t)b * N + N - 1], inverse, d, D);
        float* o = hyp + ((size_t)b * D + d) * HW;
        for (int p = (int)blockIdx.x * 256 + (int)threadIdx.x; p < HW; p += init_blocks * 256) o[p] = v;
        return;
    }
    if (blockIdx.y != 0 || blockIdx.z != 0) return;
    const int per = B * (V - 1);
    for (int idx = (int)threadIdx.x; idx < pp.n * per; idx += 256) {
        const int s = idx / per, r = idx - s * per;
        const int b = r / (V - 1), v = 1 + r % (V - 1);
        double Pr[16], Ps[16];
        compose_p(pp.proj[s] + (size_t)(b * V) * 32, Pr);
        compose_p(pp.proj[s] + (size_t)(b * V + v) * 32, Ps);
        homography_out(Ps, Pr, hom + (size_t)idx * 12);
    }
}

__global__ void homography_from_proj_kernel(const float* src_proj, const float* ref_proj, int B, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double Pr[16], Ps[16];
    for (int i = 0; i < 16; ++i) { Pr[i] = (double)ref_proj[b * 16 + i]; Ps[i] = (double)src_proj[b * 16 + i]; }
    homography_out(Ps, Pr, out + (size_t)b * 12);
}

__device__ __forceinline__ Homography load_homography(const float* p) {
    Homography hm;
#pragma unroll
    for (int i = 0; i < 9; ++i) hm.r[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) hm.t[i] = p[9 + i];
    return hm;
}

// ------------------------------------------------------------------------------------------------
// Work decomposition shared by both passes.  One work-item owns one pixel and a chunk of DCH = 4 consecutive
// depth hypotheses: the DCH tap sets are computed once and every channel is then gathered for all DCH planes
// back to back - neighbouring hypotheses project within a pixel or two of each other, so the 2*DCH pair loads of
// a channel fall into the same few cache lines (L1-resident re-use instead of one L2 trip per plane) and the
// reference feature is loaded once per DCH planes.  A 256-thread block = 4 waves = (64*SP pixels) x (SD depth
// chunk slots), SD = min(4, #chunks); lanes are 64 consecutive pixels so every load is a near-contiguous span.
// ------------------------------------------------------------------------------------------------
constexpr int DCH = 4;

struct ChunkMap {
    int nch, sd, sp, ppb;     // #depth chunks, chunk slots per block, pixel sub-blocks per block, pixels per block
};
__host__ __device__ inline ChunkMap chunk_map(int D) {
    ChunkMap m;
    m.nch = (D + DCH - 1) / DCH;
    m.sd = m.nch >= 4 ? 4 : (m.nch >= 2 ? 2 : 1);
    m.sp = 4 / m.sd;
    m.ppb = 64 * m.sp;
    return m;
}

// Correlation of the DCH planes of one work-item with one source view, GT channel groups of `cpg` channels each:
//   SUM_GROUPS  out[dd]       += sum_g mean_{c in g} ref[c] * bilinear(src[c])      (sim_vol, cost_volume.py:90)
//   otherwise   out[dd*GT+g]   = mean_{c in g} ref[c] * bilinear(src[c])            (in_prod_vol, cost_volume.py:79-87)
// The group loop is unrolled (accumulators are statically indexed registers) but the channel loop inside a group is
// a REAL loop with a run-time trip count: the feature loads are invariant loads, which neither a memory clobber nor
// sched_barrier pins, so a fully unrolled body issues every load of every channel up front and needs > 512
// registers.  One iteration = one channel = 1 + 2*DCH independent loads in flight per work-item.
template <typename T, int GT, bool SUM_GROUPS>
__device__ __forceinline__ void correlate_chunk(const T* __restrict__ src, const T* __restrict__ ref, const PairTaps* tp, unsigned HW,
                                                unsigned pc, int cpg, float* out) {
    typedef typename PairOf<T>::type P2;
    const float inv_cpg = 1.0f / (float)cpg;
    unsigned plane = 0;                                        // c * HW, uniform
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        float acc[DCH];
#pragma unroll
        for (int dd = 0; dd < DCH; ++dd) acc[dd] = 0.0f;
#pragma unroll 1
        for (int cc = 0; cc < cpg; ++cc) {
            const T* sp = src + plane;
            const float r = to_f32(ref[plane + pc]);
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd) {
                const P2 t = *reinterpret_cast<const P2*>(sp + (unsigned)tp[dd].top);
                const P2 b = *reinterpret_cast<const P2*>(sp + (unsigned)tp[dd].bot);
                float wv = tp[dd].w00 * to_f32(t.x);
                wv += tp[dd].w01 * to_f32(t.y);
                wv += tp[dd].w10 * to_f32(b.x);
                wv += tp[dd].w11 * to_f32(b.y);
                acc[dd] += r * wv;
            }
            plane += HW;
        }
#pragma unroll
        for (int dd = 0; dd < DCH; ++dd) {
            if (SUM_GROUPS) out[dd] += acc[dd] * inv_cpg;
            else out[dd * GT + g] = acc[dd] * inv_cpg;
        }
    }
}

// generic (run-time C, G) single-voxel correlation used by the fallback kernels
template <typename T>
__device__ __forceinline__ void correlate_generic(const T* __restrict__ src, const T* __restrict__ ref, const Taps& tp, int HW, int pc, int C,
                                                  int G, float* ip) {
    const int cpg = C / G;
    const float inv_cpg = 1.0f / (float)cpg;
    for (int g = 0; g < G; ++g) {
        float acc = 0.0f;
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = g * cpg + cc;
            const T* sp = src + (size_t)c * HW;
            float wv = tp.w[0] * to_f32(sp[tp.off[0]]);
            wv += tp.w[1] * to_f32(sp[tp.off[1]]);
            wv += tp.w[2] * to_f32(sp[tp.off[2]]);
            wv += tp.w[3] * to_f32(sp[tp.off[3]]);
            acc += to_f32(ref[(size_t)c * HW + pc]) * wv;
        }
        ip[g] = acc * inv_cpg;
    }
}

__device__ __forceinline__ void softmax_entropy_store(const float* sim, int stride, int D, float* dst) {
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, sim[d * stride]);
    float den = 0.0f;
    for (int d = 0; d < D; ++d) den += expf(sim[d * stride] - m);
    float ent = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = expf(sim[d * stride] - m) / den;
        ent += -pr * logf(pr + 1e-7f);                      // cost_volume.py:92
    }
    *dst = ent;
}

// ------------------------------------------------------------------------------------------------
// pass 1: entropy of the depth-softmax of the group-summed correlation        cost_volume.py:79-92
// grid = (pixel blocks, views in launch, B).  dynamic LDS: sim[D][pixels per block] floats.
// ------------------------------------------------------------------------------------------------
template <int DT, int CT, int GT>
__global__ __launch_bounds__(256) void warp_corr_entropy_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                                const float* __restrict__ hyp, float* __restrict__ entropy,
                                                                int V, int C_, int G_, int D, int H, int W, int view_begin, int nblk) {
    typedef typename FeatT<DT>::type T;
    HIP_DYNAMIC_SHARED(float, sim)
    const int HW = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = view_begin + (int)blockIdx.y;
    const int b = (int)blockIdx.z;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const Homography hm = load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const T* feat = reinterpret_cast<const T*>(feat_);
    if (CT > 0) {
        const ChunkMap cm = chunk_map(D);
        const int psub = wave / cm.sd, cslot = wave % cm.sd;
        const int pl = psub * 64 + lane;                     // pixel inside the block
        const int p = blk * cm.ppb + pl;
        const bool valid = p < HW;
        const int pc = valid ? p : HW - 1;
        const int y = pc / W, x = pc - y * W;
        const float fx = (float)x, fy = (float)y;
        const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
        const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
        const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
        const T* ref = feat + (size_t)(b * V) * C_ * HW;
        const T* src = feat + (size_t)(b * V + v) * C_ * HW;
        const float* hp = hyp + (size_t)b * D * HW + pc;
        const int cpg = C_ / GT;
        for (int ch = cslot; ch < cm.nch; ch += cm.sd) {
            const int d0 = ch * DCH;
            PairTaps tp[DCH];
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd) {
                const int d = d0 + dd < D ? d0 + dd : D - 1;
                tp[dd] = make_pair_taps(hm, qx, qy, qz, hp[(size_t)d * HW], H, W, half_w, half_h);
            }
            float s[DCH];
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd) s[dd] = 0.0f;
            correlate_chunk<T, (GT > 0 ? GT : 8), true>(src, ref, tp, (unsigned)HW, (unsigned)pc, cpg, s);
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd)
                if (d0 + dd < D) sim[(d0 + dd) * cm.ppb + pl] = s[dd];
        }
        __syncthreads();
        if (cslot == 0 && valid) softmax_entropy_store(sim + pl, cm.ppb, D, entropy + (size_t)(b * (V - 1) + (v - 1)) * HW + p);
    } else {
        // run-time C / G fallback: 64 pixels x 4 depth slots, one voxel at a time
        const int C = C_, G = G_;
        const int p = blk * 64 + lane;
        const bool valid = p < HW;
        const int pc = valid ? p : HW - 1;
        const int y = pc / W, x = pc - y * W;
        const float fx = (float)x, fy = (float)y;
        const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
        const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
        const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
        const T* ref = feat + (size_t)(b * V) * C * HW;
        const T* src = feat + (size_t)(b * V + v) * C * HW;
        const float* hp = hyp + (size_t)b * D * HW + pc;
        for (int d = wave; d < D; d += 4) {
            const Taps tp = make_taps(hm, qx, qy, qz, hp[(size_t)d * HW], H, W, half_w, half_h, nullptr);
            float ip[64];
            correlate_generic<T>(src, ref, tp, HW, pc, C, G, ip);
            float s = 0.0f;
            for (int g = 0; g < G; ++g) s += ip[g];
            sim[d * 64 + lane] = s;
        }
        __syncthreads();
        if (wave == 0 && valid) softmax_entropy_store(sim + lane, 64, D, entropy + (size_t)(b * (V - 1) + (v - 1)) * HW + p);
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: visibility-weighted aggregation over the source views of the launch    cost_volume.py:97-101
// grid = (pixel blocks, 1, B); output channel-last [D,HW,G].
// ------------------------------------------------------------------------------------------------
template <int DT, int CT, int GT>
__global__ __launch_bounds__(256) void warp_corr_aggregate_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                                  const float* __restrict__ hyp, const float* __restrict__ vis,
                                                                  float* __restrict__ vol, float* __restrict__ vis_sum, int normalise,
                                                                  int V, int C_, int G_, int D, int H, int W, int view_begin,
                                                                  int view_end, int nblk) {
    typedef typename FeatT<DT>::type T;
    const int HW = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = (int)blockIdx.z;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const T* feat = reinterpret_cast<const T*>(feat_);
    ChunkMap cm;
    if (CT > 0) cm = chunk_map(D); else { cm.nch = D; cm.sd = 4; cm.sp = 1; cm.ppb = 64; }
    const int psub = wave / cm.sd, cslot = wave % cm.sd;
    const int p = blk * cm.ppb + psub * 64 + lane;
    const bool valid = p < HW;
    const int pc = valid ? p : HW - 1;
    const int y = pc / W, x = pc - y * W;
    const float fx = (float)x, fy = (float)y;
    float vsum = 0.0f;
    for (int v = view_begin; v < view_end; ++v) vsum += vis[(size_t)(b * (V - 1) + (v - 1)) * HW + pc];   // cost_volume.py:98
    if (vis_sum != nullptr && cslot == 0 && valid) vis_sum[(size_t)b * HW + p] = vsum;
    const float denom = vsum + 1e-6f;                                                                     // cost_volume.py:101
    const float* hp = hyp + (size_t)b * D * HW + pc;
    if (CT > 0) {
        constexpr int GG = GT > 0 ? GT : 8;
        const int CC = C_, cpg = C_ / GG;
        const T* ref = feat + (size_t)(b * V) * CC * HW;
        for (int ch = cslot; ch < cm.nch; ch += cm.sd) {
            const int d0 = ch * DCH;
            float depth[DCH];
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd) depth[dd] = hp[(size_t)(d0 + dd < D ? d0 + dd : D - 1) * HW];
            float acc[DCH * GG];
#pragma unroll
            for (int i = 0; i < DCH * GG; ++i) acc[i] = 0.0f;
            for (int v = view_begin; v < view_end; ++v) {
                const Homography hm = load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
                const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
                const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
                const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
                PairTaps tp[DCH];
#pragma unroll
                for (int dd = 0; dd < DCH; ++dd) tp[dd] = make_pair_taps(hm, qx, qy, qz, depth[dd], H, W, half_w, half_h);
                float ip[DCH * GG];
                correlate_chunk<T, GG, false>(feat + (size_t)(b * V + v) * CC * HW, ref, tp, (unsigned)HW, (unsigned)pc, cpg, ip);
                const float w = vis[(size_t)(b * (V - 1) + (v - 1)) * HW + pc];
#pragma unroll
                for (int i = 0; i < DCH * GG; ++i) acc[i] += ip[i] * w;                                   // cost_volume.py:97
            }
            if (valid) {
#pragma unroll
                for (int dd = 0; dd < DCH; ++dd) {
                    if (d0 + dd >= D) continue;
                    float* o = vol + ((size_t)(b * D + d0 + dd) * HW + p) * GG;
                    float r[GG];
#pragma unroll
                    for (int g = 0; g < GG; ++g) r[g] = normalise ? acc[dd * GG + g] / denom : acc[dd * GG + g];
                    if (GG == 8) {
                        reinterpret_cast<float4*>(o)[0] = make_float4(r[0], r[1], r[2], r[3]);
                        reinterpret_cast<float4*>(o)[1] = make_float4(r[4 % GG], r[5 % GG], r[6 % GG], r[7 % GG]);
                    } else {
                        for (int g = 0; g < GG; ++g) o[g] = r[g];
                    }
                }
            }
        }
    } else {
        const int C = C_, G = G_;
        const T* ref = feat + (size_t)(b * V) * C * HW;
        for (int d = cslot; d < D; d += 4) {
            const float depth = hp[(size_t)d * HW];
            float acc[64];
            for (int g = 0; g < G; ++g) acc[g] = 0.0f;
            for (int v = view_begin; v < view_end; ++v) {
                const Homography hm = load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
                const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
                const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
                const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
                const Taps tp = make_taps(hm, qx, qy, qz, depth, H, W, half_w, half_h, nullptr);
                float ip[64];
                correlate_generic<T>(feat + (size_t)(b * V + v) * C * HW, ref, tp, HW, pc, C, G, ip);
                const float w = vis[(size_t)(b * (V - 1) + (v - 1)) * HW + pc];
                for (int g = 0; g < G; ++g) acc[g] += ip[g] * w;
            }
            if (valid) {
                float* o = vol + ((size_t)(b * D + d) * HW + p) * G;
                for (int g = 0; g < G; ++g) o[g] = normalise ? acc[g] / denom : acc[g];
            }
        }
    }
}

__global__ void volume_normalise_kernel(float* __restrict__ vol, const float* __restrict__ vis_sum, int D, int HW, int G, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t vox = i / G;
        const size_t p = vox % HW;
        const size_t b = vox / ((size_t)D * HW);
        vol[i] = vol[i] / (vis_sum[b * HW + p] + 1e-6f);
    }
}

// fp32 channel-last volume [.., 8] -> fp16 [.., 8] in a SEPARATE buffer (MVS_PREC_F16X2 U-Net), optionally normalising by the summed
// visibility first; clamped to the fp16 range.  Out of place: a compacting in-place conversion would race between work-items.
__global__ void volume_to_f16_kernel(const float* __restrict__ vol, const float* __restrict__ vis_sum, _Float16* __restrict__ out, int D, int HW, size_t nvox) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    float sat_amax = 0.0f;                                       // fp16 saturation counter (mvs_common.h)
    for (size_t vox = (size_t)blockIdx.x * blockDim.x + threadIdx.x; vox < nvox; vox += (size_t)gridDim.x * blockDim.x) {
        const float4* q = reinterpret_cast<const float4*>(vol + vox * 8);
        const float4 a = q[0], c = q[1];
        float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        const float den = vis_sum != nullptr ? vis_sum[(vox / ((size_t)D * HW)) * HW + vox % HW] + 1e-6f : 1.0f;
        h8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (vis_sum != nullptr) x[j] = x[j] / den;
            hv[j] = (_Float16)fminf(fmaxf(x[j], -65504.0f), 65504.0f);
        }
        sat::track(sat_amax, x[0], x[1], x[2], x[3]);
        sat::track(sat_amax, x[4], x[5], x[6], x[7]);
        *reinterpret_cast<h8*>(out + vox * 8) = hv;
    }
    sat::commit(sat_amax);
}

// fp32 channel-last volume [.., 8] -> the split activation format of MVS_PREC_BF16X3_SPLIT (per voxel [hi x8 | lo x8] bf16), in
// place (same 32 bytes per voxel), optionally normalising by the summed visibility first (the view-sharded multi-GPU path)
__global__ void volume_to_split_kernel(float* __restrict__ vol, const float* __restrict__ vis_sum, int D, int HW, size_t nvox) {
    for (size_t vox = (size_t)blockIdx.x * blockDim.x + threadIdx.x; vox < nvox; vox += (size_t)gridDim.x * blockDim.x) {
        float4* q = reinterpret_cast<float4*>(vol + vox * 8);
        const float4 a = q[0], c = q[1];
        float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        if (vis_sum != nullptr) {
            const float den = vis_sum[(vox / ((size_t)D * HW)) * HW + vox % HW] + 1e-6f;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] / den;
        }
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = from_f32<uint16_t>(x[j]);
            lo[j] = from_f32<uint16_t>(x[j] - to_f32(hi[j]));
        }
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hw[j] = (unsigned)hi[2 * j] | ((unsigned)hi[2 * j + 1] << 16);
            lw[j] = (unsigned)lo[2 * j] | ((unsigned)lo[2 * j + 1] << 16);
        }
        q[0] = make_float4(__builtin_bit_cast(float, hw[0]), __builtin_bit_cast(float, hw[1]), __builtin_bit_cast(float, hw[2]), __builtin_bit_cast(float, hw[3]));
        q[1] = make_float4(__builtin_bit_cast(float, lw[0]), __builtin_bit_cast(float, lw[1]), __builtin_bit_cast(float, lw[2]), __builtin_bit_cast(float, lw[3]));
    }
}

// ------------------------------------------------------------------------------------------------
// Slab exchange of the view-sharded latency mode (no reference counterpart: the reference is single-GPU).
// A message to / from rank j = the rows [r0_j, r1_j) of a partial volume [B,D,H,W,G] followed by the same rows of the partial
// visibility sum [B,H,W].  Round 2 built every message with two strided torch copies and summed the received partials with R - 1
// separate add_ launches; here ONE launch packs the messages of all destinations and ONE launch sums the own slice and every
// received message (fixed rank order) into the slab the regulariser reads.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxShardRanks = 16;
struct SlabMsgs {
    float* ptr[kMaxShardRanks];
    int r0[kMaxShardRanks], r1[kMaxShardRanks];
    int n;
};

// element e of a message over rows [r0, r1): e < nvol -> volume element (b, d, row, x, g), else visibility-sum element (b, row, x)
__device__ __forceinline__ float slab_src(const float* __restrict__ vol, const float* __restrict__ vsum, size_t e, int r0, int rows, int B, int D, int H, int W,
                                          int G) {
    const size_t rowq = (size_t)W * G, nvol = (size_t)B * D * rows * rowq;
    if (e < nvol) {
        const size_t x = e % rowq, t = e / rowq;
        const size_t r = t % rows, bd = t / rows;
        return vol[(bd * H + (size_t)r0 + r) * rowq + x];
    }
    const size_t k = e - nvol, x = k % W, t = k / W;
    const size_t r = t % rows, b = t / rows;
    return vsum[(b * H + (size_t)r0 + r) * W + x];
}

// grid = (blocks, destinations)
__global__ void slab_pack_kernel(const float* __restrict__ vol, const float* __restrict__ vsum, SlabMsgs m, int B, int D, int H, int W, int G) {
    const int j = (int)blockIdx.y;
    const int r0 = m.r0[j], rows = m.r1[j] - r0;
    if (m.ptr[j] == nullptr || rows <= 0) return;
    const size_t n = (size_t)B * D * rows * W * G + (size_t)B * rows * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) m.ptr[j][e] = slab_src(vol, vsum, e, r0, rows, B, D, H, W, G);
}

// out[e] = sum over ranks in rank order of their partial of my slab: the own slice straight from (vol, vsum), the others from their messages
__global__ void slab_reduce_kernel(const float* __restrict__ vol, const float* __restrict__ vsum, SlabMsgs m, int my_rank, float* __restrict__ out,
                                   int r0, int r1, int B, int D, int H, int W, int G) {
    const int rows = r1 - r0;
    const size_t n = (size_t)B * D * rows * W * G + (size_t)B * rows * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int j = 0; j < m.n; ++j) acc += (j == my_rank) ? slab_src(vol, vsum, e, r0, rows, B, D, H, W, G) : (m.ptr[j] != nullptr ? m.ptr[j][e] : 0.0f);
        out[e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// a2/a3 standalone: warped [B,C,D,H,W] + proj_mask [B,D,H,W]                  warping.py:69-109
// grid = (pixel blocks of 64, D-chunks, B); block = 64 px x 4 depth slots.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void homo_warp_kernel(const void* __restrict__ src_, const float* __restrict__ hom,
                                                        const float* __restrict__ depth_, int depth_is_volume,
                                                        float* __restrict__ warped, uint8_t* __restrict__ mask, int C, int D, int H,
                                                        int W) {
    typedef typename FeatT<DT>::type T;
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int p = (int)blockIdx.x * 64 + lane;
    const int b = (int)blockIdx.z;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const Homography hm = load_homography(hom + (size_t)b * 12);
    const float fx = (float)x, fy = (float)y;
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const T* src = reinterpret_cast<const T*>(src_) + (size_t)b * C * HW;
    for (int d = (int)blockIdx.y * 4 + slot; d < D; d += (int)gridDim.y * 4) {
        const float depth = depth_is_volume ? depth_[((size_t)b * D + d) * HW + p] : depth_[(size_t)b * D + d];
        bool oof;
        const Taps tp = make_taps(hm, qx, qy, qz, depth, H, W, half_w, half_h, &oof);
        if (mask != nullptr) mask[((size_t)b * D + d) * HW + p] = oof ? 1 : 0;
        if (warped != nullptr) {
            for (int c = 0; c < C; ++c) {
                const T* sp = src + (size_t)c * HW;
                float wv = tp.w[0] * to_f32(sp[tp.off[0]]);
                wv += tp.w[1] * to_f32(sp[tp.off[1]]);
                wv += tp.w[2] * to_f32(sp[tp.off[2]]);
                wv += tp.w[3] * to_f32(sp[tp.off[3]]);
                warped[(((size_t)b * C + c) * D + d) * HW + p] = wv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
template <int DT, int CT, int GT>
static int launch_entropy(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int C, int G,
                          int D, int H, int W, int vb, int ve, hipStream_t st) {
    const int HW = H * W;
    const int ppb = CT > 0 ? chunk_map(D).ppb : 64;
    const int nblk = (int)ceil_div(HW, ppb);
    const size_t lds = (size_t)D * ppb * sizeof(float);
    if (lds > 160 * 1024) { set_error("warp_corr_entropy: D=%d needs %zu B of LDS (> 160 KiB)", D, lds); return MVS_ERR_UNSUPPORTED; }
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_corr_entropy_kernel<DT, CT, GT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((warp_corr_entropy_kernel<DT, CT, GT>), dim3(nblk, ve - vb, B), dim3(256), lds, st, feat, hom, hyp, ent, V, C, G, D,
                       H, W, vb, nblk);
    return check_launch("warp_corr_entropy_kernel");
}

template <int DT, int CT, int GT>
static int launch_aggregate(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                            int normalise, int B, int V, int C, int G, int D, int H, int W, int vb, int ve, hipStream_t st) {
    const int nblk = (int)ceil_div((long long)H * W, CT > 0 ? chunk_map(D).ppb : 64);
    hipLaunchKernelGGL((warp_corr_aggregate_kernel<DT, CT, GT>), dim3(nblk, 1, B), dim3(256), 0, st, feat, hom, hyp, vis, vol, vis_sum,
                       normalise, V, C, G, D, H, W, vb, ve, nblk);
    return check_launch("warp_corr_aggregate_kernel");
}

// fast path (pair loads, DCH planes per work-item): G == 8 (every shipped config), any C divisible by 8, W >= 2;
// the template's CT is only the fast/generic switch now (1 = fast), C itself is a run-time value
#define MVS_DISPATCH_CG(FN, DT, ...)                                              \
    do {                                                                          \
        if (G == 8 && W >= 2) return FN<DT, 1, 8>(__VA_ARGS__);                   \
        return FN<DT, 0, 0>(__VA_ARGS__);                                         \
    } while (0)

static int check_corr_args(const char* who, const void* feat, const float* hom, const float* hyp, int dtype, int B, int V, int C, int G,
                           int D, int H, int W, int vb, int ve) {
    if (!feat || !hom || !hyp) { set_error("%s: null pointer", who); return MVS_ERR_ARG; }
    if (B < 1 || V < 2 || C < 1 || G < 1 || D < 1 || H < 1 || W < 1) { set_error("%s: bad shape", who); return MVS_ERR_ARG; }
    if (G > C || C % G != 0) { set_error("%s: G must divide C and G <= C (got C=%d G=%d)", who, C, G); return MVS_ERR_ARG; }   // cost_volume.py:87
    if (G > 64) { set_error("%s: G=%d > 64 unsupported", who, G); return MVS_ERR_UNSUPPORTED; }
    if (vb < 1 || ve > V || vb >= ve) { set_error("%s: bad source-view range [%d,%d) for V=%d", who, vb, ve, V); return MVS_ERR_ARG; }
    if (dtype < 0 || dtype > 2) { set_error("%s: bad dtype %d", who, dtype); return MVS_ERR_ARG; }
    if ((long long)B * V * C * H * W > 0x7fffffffLL * 2 || (long long)C * H * W > 0x7fffffffLL) { set_error("%s: feature tensor too large for 32-bit plane offsets", who); return MVS_ERR_UNSUPPORTED; }
    return MVS_OK;
}

// LDS-staged form of the two gather passes (gather_lds_kernels.hip)
bool gl_supported(int C, int G, int D, int H, int W);
int gl_launch_entropy(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H,
                      int W, int vb, int ve, int w16, hipStream_t st);
int gl_launch_aggregate(const void* feat, int dtype, int layout, const float* hom, const float* hyp, const float* vis, float* vol,
                        float* vis_sum, int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st);
bool gl_keep_supported(int C, int G, int D, int H, int W);
bool gl_keep32_supported(int C, int G, int D, int H, int W);
int gl_launch_entropy_keep(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, void* corr, int corr_format, int B,
                           int V, int C, int D, int H, int W, hipStream_t st);
int launch_corr_aggregate(const void* corr, int corr_format, const float* vis, void* vol, int volume_format, int B, int V, int D, int H, int W,
                          hipStream_t st);
int pack_features_dispatch(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W, hipStream_t st);

// MVS_GATHER_IMPL=direct selects the round-1 direct-gather kernels (A/B measurements); anything else = LDS-staged where supported.
// (Round 3 measured a third, wave-autonomous form - one private window per wave, no workgroup barriers, register prefetch of the
// next window: parity green, 1.8-2.3x SLOWER than the workgroup-window kernels on the MI355X, profiles/r03_gather_wave_vs_lds.txt;
// it lives in git history, commit b186996.)
static int gather_impl(int C, int G, int D, int H, int W) {
    const char* e = getenv("MVS_GATHER_IMPL");
    if (e && e[0] == 'd') return 0;
    return gl_supported(C, G, D, H, W) ? 1 : 0;
}

}  // namespace mvs

using namespace mvs;

static int volume_to_split(float* volume_cl, const float* vis_sum, int B, int D, int H, int W, int G, hipStream_t st, const char* who);

extern "C" int mvs_compose_homography(const float* proj, int B, int V, float* homography, void* stream) {
    if (!proj || !homography || B < 1 || V < 2) { set_error("mvs_compose_homography: bad arguments"); return MVS_ERR_ARG; }
    const int n = B * (V - 1);
    hipLaunchKernelGGL(compose_homography_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, proj, B, V, homography);
    return check_launch("compose_homography_kernel");
}

extern "C" int mvs_cascade_prologue_fwd(const float* const* proj_host_ptrs, int n_stages, int B, int V, float* homography, const float* depth_values,
                                       int N, int inverse, float* hyp, int D, int H, int W, void* stream) {
    if (!proj_host_ptrs || !homography || n_stages < 1 || n_stages > 8 || B < 1 || V < 2) { set_error("mvs_cascade_prologue_fwd: bad arguments (1..8 stages)"); return MVS_ERR_ARG; }
    if (hyp && (!depth_values || N < 1 || D < 2 || H < 1 || W < 1)) { set_error("mvs_cascade_prologue_fwd: bad hypothesis arguments"); return MVS_ERR_ARG; }
    ProloguePtrs pp;
    pp.n = n_stages;
    for (int i = 0; i < 8; ++i) {
        pp.proj[i] = i < n_stages ? proj_host_ptrs[i] : nullptr;
        if (i < n_stages && !pp.proj[i]) { set_error("mvs_cascade_prologue_fwd: null projection tensor of stage %d", i); return MVS_ERR_ARG; }
    }
    int init_blocks = 0;
    dim3 grid(1, 1, 1);
    if (hyp) {
        const int HW = H * W;
        init_blocks = (int)(ceil_div(HW, 1024) > 64 ? 64 : ceil_div(HW, 1024));      // ~4 elements per thread and plane, at most 64 blocks per plane
        grid = dim3(init_blocks + 1, D, B);
    }
    hipLaunchKernelGGL(cascade_prologue_kernel, grid, dim3(256), 0, (hipStream_t)stream, pp, B, V, homography, depth_values, N, inverse,
                       hyp, D, H * W, init_blocks);
    return check_launch("cascade_prologue_kernel");
}

extern "C" int mvs_homography_from_proj(const float* src_proj, const float* ref_proj, int B, float* homography, void* stream) {
    if (!src_proj || !ref_proj || !homography || B < 1) { set_error("mvs_homography_from_proj: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(homography_from_proj_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, src_proj, ref_proj, B, homography);
    return check_launch("homography_from_proj_kernel");
}

extern "C" int mvs_homo_warp_fwd(const void* src_fea, int dtype, const float* homography, const float* depth, int depth_is_volume,
                                 float* warped, uint8_t* proj_mask, int B, int C, int D, int H, int W, void* stream) {
    if (!src_fea || !homography || !depth || B < 1 || C < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_homo_warp_fwd: bad arguments"); return MVS_ERR_ARG; }
    const dim3 grid(ceil_div((long long)H * W, 64), ceil_div(D, 4) > 64 ? 64 : ceil_div(D, 4), B);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case MVS_DTYPE_F32: hipLaunchKernelGGL((homo_warp_kernel<MVS_DTYPE_F32>), grid, dim3(256), 0, st, src_fea, homography, depth, depth_is_volume, warped, proj_mask, C, D, H, W); break;
        case MVS_DTYPE_BF16: hipLaunchKernelGGL((homo_warp_kernel<MVS_DTYPE_BF16>), grid, dim3(256), 0, st, src_fea, homography, depth, depth_is_volume, warped, proj_mask, C, D, H, W); break;
        case MVS_DTYPE_F16: hipLaunchKernelGGL((homo_warp_kernel<MVS_DTYPE_F16>), grid, dim3(256), 0, st, src_fea, homography, depth, depth_is_volume, warped, proj_mask, C, D, H, W); break;
        default: set_error("mvs_homo_warp_fwd: bad dtype %d", dtype); return MVS_ERR_ARG;
    }
    return check_launch("homo_warp_kernel");
}

extern "C" int mvs_pack_features(const void* features, int dtype, void* tiled, int out_dtype, int N, int C, int H, int W, void* stream) {
    if (!features || !tiled || N < 1 || C < 8 || (C % 8) || H < 1 || W < 1 || dtype < 0 || dtype > 2 || out_dtype < 0 || out_dtype > 2) {
        set_error("mvs_pack_features: bad arguments (C must be a multiple of 8)");
        return MVS_ERR_ARG;
    }
    return pack_features_dispatch(features, dtype, tiled, out_dtype, N, C, H, W, (hipStream_t)stream);
}

static int check_layout(const char* who, int layout, int C, int G, int D, int H, int W) {
    if (layout == MVS_LAYOUT_PLANAR) return MVS_OK;
    if (layout != MVS_LAYOUT_OCTET_TILED) { set_error("%s: unknown feature layout %d", who, layout); return MVS_ERR_ARG; }
    if (!gl_supported(C, G, D, H, W)) {
        set_error("%s: the octet-tiled feature layout needs G == 8, C in {8,16,32,64} and W %% 8 == 0", who);
        return MVS_ERR_UNSUPPORTED;
    }
    return MVS_OK;
}

extern "C" int mvs_warp_corr_entropy_fwd(const void* features, int dtype, int layout, const float* homography, const float* hyp, float* entropy,
                                         int B, int V, int C, int G, int D, int H, int W, int view_begin, int view_end, int gather_format,
                                         void* stream) {
    int rc = check_corr_args("mvs_warp_corr_entropy_fwd", features, homography, hyp, dtype, B, V, C, G, D, H, W, view_begin, view_end);
    if (rc != MVS_OK) return rc;
    if (!entropy) { set_error("mvs_warp_corr_entropy_fwd: null output"); return MVS_ERR_ARG; }
    if (gather_format != MVS_GATHER_F32 && gather_format != MVS_GATHER_F16) { set_error("mvs_warp_corr_entropy_fwd: unknown gather format %d", gather_format); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    rc = check_layout("mvs_warp_corr_entropy_fwd", layout, C, G, D, H, W);
    if (rc != MVS_OK) return rc;
    const int impl = gather_impl(C, G, D, H, W);
    if (layout == MVS_LAYOUT_OCTET_TILED || impl == 1)
        return gl_launch_entropy(features, dtype, layout, homography, hyp, entropy, B, V, C, D, H, W, view_begin, view_end, gather_format == MVS_GATHER_F16, st);
    switch (dtype) {
        case MVS_DTYPE_F32: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_F32, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
        case MVS_DTYPE_BF16: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_BF16, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
        default: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_F16, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
    }
}

extern "C" int mvs_gather_keeps_correlations(int layout, int C, int G, int D, int H, int W) {
    if (layout != MVS_LAYOUT_OCTET_TILED && gather_impl(C, G, D, H, W) != 1) return 0;
    return gl_keep_supported(C, G, D, H, W) ? 1 : 0;
}

extern "C" int mvs_warp_corr_entropy_keep_fwd(const void* features, int dtype, int layout, const float* homography, const float* hyp, float* entropy,
                                              void* corr, int corr_format, int B, int V, int C, int G, int D, int H, int W, void* stream) {
    int rc = check_corr_args("mvs_warp_corr_entropy_keep_fwd", features, homography, hyp, dtype, B, V, C, G, D, H, W, 1, V);
    if (rc != MVS_OK) return rc;
    if (!entropy || !corr) { set_error("mvs_warp_corr_entropy_keep_fwd: null output"); return MVS_ERR_ARG; }
    if (corr_format != MVS_CORR_F16 && corr_format != MVS_CORR_F32) { set_error("mvs_warp_corr_entropy_keep_fwd: unknown correlation format %d", corr_format); return MVS_ERR_ARG; }
    rc = check_layout("mvs_warp_corr_entropy_keep_fwd", layout, C, G, D, H, W);
    if (rc != MVS_OK) return rc;
    if (!gl_keep_supported(C, G, D, H, W) || (corr_format == MVS_CORR_F32 && !gl_keep32_supported(C, G, D, H, W))) {
        set_error("mvs_warp_corr_entropy_keep_fwd: built for the LDS-staged gather (G == 8, C in {8,16,32,64}, W %% 8 == 0; MVS_CORR_F32: D > 4 as well); "
                  "mvs_gather_keeps_correlations() says which shapes qualify");
        return MVS_ERR_UNSUPPORTED;
    }
    return gl_launch_entropy_keep(features, dtype, layout, homography, hyp, entropy, corr, corr_format, B, V, C, D, H, W, (hipStream_t)stream);
}

extern "C" int mvs_corr_aggregate_fwd(const void* corr, int corr_format, const float* vis, void* volume_cl, int volume_format, int B, int V, int D,
                                      int H, int W, void* stream) {
    if (!corr || !vis || !volume_cl) { set_error("mvs_corr_aggregate_fwd: null pointer"); return MVS_ERR_ARG; }
    if (corr_format != MVS_CORR_F16 && corr_format != MVS_CORR_F32) { set_error("mvs_corr_aggregate_fwd: unknown correlation format %d", corr_format); return MVS_ERR_ARG; }
    if (B < 1 || V < 2 || D < 1 || H < 1 || W < 1 || (long long)D * H * W > 0x7fffffffLL) { set_error("mvs_corr_aggregate_fwd: bad shape"); return MVS_ERR_ARG; }
    if (volume_format != MVS_VOLUME_F32 && volume_format != MVS_VOLUME_SPLIT && volume_format != MVS_VOLUME_F16) { set_error("mvs_corr_aggregate_fwd: unknown volume format %d", volume_format); return MVS_ERR_ARG; }
    return launch_corr_aggregate(corr, corr_format, vis, volume_cl, volume_format, B, V, D, H, W, (hipStream_t)stream);
}

extern "C" int mvs_warp_corr_aggregate_fwd(const void* features, int dtype, int layout, const float* homography, const float* hyp, const float* vis,
                                           float* volume_cl, float* vis_sum, int normalise, int volume_format, int B, int V, int C, int G,
                                           int D, int H, int W, int view_begin, int view_end, void* stream) {
    int rc = check_corr_args("mvs_warp_corr_aggregate_fwd", features, homography, hyp, dtype, B, V, C, G, D, H, W, view_begin, view_end);
    if (rc != MVS_OK) return rc;
    if (!vis || !volume_cl) { set_error("mvs_warp_corr_aggregate_fwd: null pointer"); return MVS_ERR_ARG; }
    if (!normalise && !vis_sum) { set_error("mvs_warp_corr_aggregate_fwd: partial mode needs vis_sum"); return MVS_ERR_ARG; }
    if (volume_format != MVS_VOLUME_F32 && volume_format != MVS_VOLUME_SPLIT && volume_format != MVS_VOLUME_F16) { set_error("mvs_warp_corr_aggregate_fwd: unknown volume format %d", volume_format); return MVS_ERR_ARG; }
    if (volume_format != MVS_VOLUME_F32 && (!normalise || G != 8)) {
        set_error("mvs_warp_corr_aggregate_fwd: the split / fp16 volume formats hold the NORMALISED volume of 8 groups (partial sums stay fp32)");
        return MVS_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    rc = check_layout("mvs_warp_corr_aggregate_fwd", layout, C, G, D, H, W);
    if (rc != MVS_OK) return rc;
    const int impl = gather_impl(C, G, D, H, W);
    if (layout == MVS_LAYOUT_OCTET_TILED || impl == 1)
        return gl_launch_aggregate(features, dtype, layout, homography, hyp, vis, volume_cl, vis_sum, normalise | (volume_format << 1), B, V, C, D, H, W, view_begin, view_end, st);
    if (volume_format == MVS_VOLUME_F16) {
        set_error("mvs_warp_corr_aggregate_fwd: the fp16 volume is written by the LDS-staged gather only (C in {8,16,32,64}, W %% 8 == 0); "
                  "for other shapes aggregate in MVS_VOLUME_F32 and convert with mvs_volume_to_f16");
        return MVS_ERR_UNSUPPORTED;
    }
    if (volume_format == MVS_VOLUME_SPLIT) {                // shapes outside the LDS-staged fast path: fp32 volume, converted in place
        rc = mvs_warp_corr_aggregate_fwd(features, dtype, layout, homography, hyp, vis, volume_cl, vis_sum, normalise, MVS_VOLUME_F32, B, V, C, G, D, H, W,
                                         view_begin, view_end, stream);
        return rc != MVS_OK ? rc : volume_to_split(volume_cl, nullptr, B, D, H, W, G, st, "mvs_warp_corr_aggregate_fwd");
    }
    switch (dtype) {
        case MVS_DTYPE_F32: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_F32, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
        case MVS_DTYPE_BF16: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_BF16, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
        default: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_F16, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
    }
}

static int volume_to_split(float* volume_cl, const float* vis_sum, int B, int D, int H, int W, int G, hipStream_t st, const char* who) {
    if (G != 8) { set_error("%s: the split volume format needs 8 groups", who); return MVS_ERR_UNSUPPORTED; }
    const size_t nvox = (size_t)B * D * H * W;
    const unsigned grid = (unsigned)((nvox + 255) / 256 > 16384 ? 16384 : (nvox + 255) / 256);
    hipLaunchKernelGGL(volume_to_split_kernel, dim3(grid), dim3(256), 0, st, volume_cl, vis_sum, D, H * W, nvox);
    return check_launch("volume_to_split_kernel");
}

static bool slab_args(const char* who, const float* volume_cl, const float* vis_sum, float* const* bufs, const int* r0, const int* r1, int n, int B, int D,
                      int H, int W, int G, SlabMsgs* m) {
    if (!volume_cl || !vis_sum || !bufs || !r0 || !r1 || n < 1 || n > kMaxShardRanks || B < 1 || D < 1 || H < 1 || W < 1 || G < 1) {
        set_error("%s: bad arguments (1..%d ranks)", who, kMaxShardRanks);
        return false;
    }
    m->n = n;
    for (int j = 0; j < kMaxShardRanks; ++j) {
        m->ptr[j] = j < n ? bufs[j] : nullptr;
        m->r0[j] = j < n ? r0[j] : 0;
        m->r1[j] = j < n ? r1[j] : 0;
        if (j < n && bufs[j] != nullptr && (r0[j] < 0 || r1[j] > H || r1[j] < r0[j])) { set_error("%s: rows [%d, %d) outside the volume", who, r0[j], r1[j]); return false; }
    }
    return true;
}

extern "C" int mvs_slab_pack(const float* volume_cl, const float* vis_sum, float* const* send_host_ptrs, const int* row_begin, const int* row_end,
                             int n_ranks, int B, int D, int H, int W, int G, void* stream) {
    SlabMsgs m;
    if (!slab_args("mvs_slab_pack", volume_cl, vis_sum, send_host_ptrs, row_begin, row_end, n_ranks, B, D, H, W, G, &m)) return MVS_ERR_ARG;
    hipLaunchKernelGGL(slab_pack_kernel, dim3(1024, n_ranks), dim3(256), 0, (hipStream_t)stream, volume_cl, vis_sum, m, B, D, H, W, G);
    return check_launch("slab_pack_kernel");
}

extern "C" int mvs_slab_reduce(const float* volume_cl, const float* vis_sum, float* const* recv_host_ptrs, int n_ranks, int my_rank, float* slab_out,
                               int row_begin, int row_end, int B, int D, int H, int W, int G, void* stream) {
    int r0[kMaxShardRanks], r1[kMaxShardRanks];
    for (int j = 0; j < kMaxShardRanks; ++j) { r0[j] = row_begin; r1[j] = row_end; }
    SlabMsgs m;
    if (!slab_out || my_rank < 0 || my_rank >= n_ranks || row_begin < 0 || row_end > H || row_end <= row_begin ||
        !slab_args("mvs_slab_reduce", volume_cl, vis_sum, recv_host_ptrs, r0, r1, n_ranks, B, D, H, W, G, &m)) {
        if (!slab_out || my_rank < 0 || my_rank >= n_ranks || row_begin < 0 || row_end > H || row_end <= row_begin) set_error("mvs_slab_reduce: bad arguments");
        return MVS_ERR_ARG;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, volume_cl, vis_sum, m, my_rank, slab_out, row_begin, row_end, B, D, H,
                       W, G);
    return check_launch("slab_reduce_kernel");
}

extern "C" int mvs_gather_is_lds_staged(int layout, int C, int G, int D, int H, int W) {
    return (layout == MVS_LAYOUT_OCTET_TILED || gather_impl(C, G, D, H, W) == 1) ? 1 : 0;
}

extern "C" int mvs_volume_to_f16(const float* volume_cl, const float* vis_sum, void* out_f16, int B, int D, int H, int W, int G, void* stream) {
    if (!volume_cl || !out_f16 || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_volume_to_f16: bad arguments"); return MVS_ERR_ARG; }
    if (G != 8) { set_error("mvs_volume_to_f16: the fp16 volume format needs 8 groups"); return MVS_ERR_UNSUPPORTED; }
    const size_t nvox = (size_t)B * D * H * W;
    const unsigned grid = (unsigned)((nvox + 255) / 256 > 16384 ? 16384 : (nvox + 255) / 256);
    hipLaunchKernelGGL(volume_to_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, volume_cl, vis_sum, static_cast<_Float16*>(out_f16), D, H * W, nvox);
    return check_launch("volume_to_f16_kernel");
}

extern "C" int mvs_volume_normalise(float* volume_cl, const float* vis_sum, int B, int D, int H, int W, int G, int volume_format, void* stream) {
    if (!volume_cl || !vis_sum || B < 1 || D < 1 || H < 1 || W < 1 || G < 1) { set_error("mvs_volume_normalise: bad arguments"); return MVS_ERR_ARG; }
    if (volume_format == MVS_VOLUME_SPLIT) return volume_to_split(volume_cl, vis_sum, B, D, H, W, G, (hipStream_t)stream, "mvs_volume_normalise");
    if (volume_format != MVS_VOLUME_F32) { set_error("mvs_volume_normalise: unknown volume format %d", volume_format); return MVS_ERR_ARG; }
    const size_t total = (size_t)B * D * H * W * G;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(volume_normalise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, volume_cl, vis_sum, D, H * W, G, total);
    return check_launch("volume_normalise_kernel");
}

namespace mvs { MVS_DEFINE_SAT_READER(sat_read_warp) }
