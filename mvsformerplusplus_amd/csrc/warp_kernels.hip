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

// group-wise correlation of one voxel with one source view: ip[g] = mean_{c in g} ref[c] * bilinear(src[c])
template <typename T, int CT, int GT>
__device__ __forceinline__ void correlate(const T* __restrict__ src, const T* __restrict__ ref_g, const float* r_cached,
                                          const Taps& tp, int HW, int pc, int C, int G, float* ip) {
    const int cpg = C / G;
    const float inv_cpg = 1.0f / (float)cpg;
    if (CT > 0) {
#pragma unroll
        for (int g = 0; g < (GT > 0 ? GT : 1); ++g) {
            float acc = 0.0f;
#pragma unroll
            for (int cc = 0; cc < (CT > 0 && GT > 0 ? CT / GT : 1); ++cc) {
                const int c = g * (CT / (GT > 0 ? GT : 1)) + cc;
                const T* sp = src + (size_t)c * HW;
                float wv = tp.w[0] * to_f32(sp[tp.off[0]]);
                wv += tp.w[1] * to_f32(sp[tp.off[1]]);
                wv += tp.w[2] * to_f32(sp[tp.off[2]]);
                wv += tp.w[3] * to_f32(sp[tp.off[3]]);
                acc += r_cached[c] * wv;
            }
            ip[g] = acc * inv_cpg;
        }
    } else {
        for (int g = 0; g < G; ++g) {
            float acc = 0.0f;
            for (int cc = 0; cc < cpg; ++cc) {
                const int c = g * cpg + cc;
                const T* sp = src + (size_t)c * HW;
                float wv = tp.w[0] * to_f32(sp[tp.off[0]]);
                wv += tp.w[1] * to_f32(sp[tp.off[1]]);
                wv += tp.w[2] * to_f32(sp[tp.off[2]]);
                wv += tp.w[3] * to_f32(sp[tp.off[3]]);
                acc += to_f32(ref_g[(size_t)c * HW + pc]) * wv;
            }
            ip[g] = acc * inv_cpg;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 1: entropy of the depth-softmax of the group-summed correlation        cost_volume.py:79-92
// grid = (pixel blocks of 64, views in launch, B); block = 64 pixels x 4 depth slots.
// dynamic LDS: sim[D][64] floats.
// ------------------------------------------------------------------------------------------------
template <int DT, int CT, int GT>
__global__ __launch_bounds__(256) void warp_corr_entropy_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                                const float* __restrict__ hyp, float* __restrict__ entropy,
                                                                int V, int C_, int G_, int D, int H, int W, int view_begin, int nblk) {
    typedef typename FeatT<DT>::type T;
    const int C = CT > 0 ? CT : C_;
    const int G = GT > 0 ? GT : G_;
    HIP_DYNAMIC_SHARED(float, sim)
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int p = blk * 64 + lane;
    const int v = view_begin + (int)blockIdx.y;
    const int b = (int)blockIdx.z;
    const bool valid = p < HW;
    const int pc = valid ? p : HW - 1;
    const int y = pc / W, x = pc - y * W;
    const Homography hm = load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
    const float fx = (float)x, fy = (float)y;
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const T* feat = reinterpret_cast<const T*>(feat_);
    const T* ref = feat + (size_t)(b * V) * C * HW;
    const T* src = feat + (size_t)(b * V + v) * C * HW;
    float r[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < (CT > 0 ? CT : 1); ++c) r[c] = to_f32(ref[(size_t)c * HW + pc]);
    }
    const float* hp = hyp + (size_t)b * D * HW + pc;
    for (int d = slot; d < D; d += 4) {
        const float depth = hp[(size_t)d * HW];
        const Taps tp = make_taps(hm, qx, qy, qz, depth, H, W, half_w, half_h, nullptr);
        float ip[GT > 0 ? GT : 64];
        correlate<T, CT, GT>(src, ref, r, tp, HW, pc, C, G, ip);
        float s = 0.0f;
        for (int g = 0; g < G; ++g) s += ip[g];                 // sim_vol = in_prod_vol.sum(dim=1)
        sim[d * 64 + lane] = s;
    }
    __syncthreads();
    if (slot == 0 && valid) {
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) m = fmaxf(m, sim[d * 64 + lane]);
        float den = 0.0f;
        for (int d = 0; d < D; ++d) den += expf(sim[d * 64 + lane] - m);
        float ent = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pr = expf(sim[d * 64 + lane] - m) / den;
            ent += -pr * logf(pr + 1e-7f);                      // cost_volume.py:92
        }
        entropy[(size_t)(b * (V - 1) + (v - 1)) * HW + p] = ent;
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: visibility-weighted aggregation over the source views of the launch    cost_volume.py:97-101
// grid = (pixel blocks of 64, 1, B); block = 64 pixels x 4 depth slots; output channel-last [D,HW,G].
// ------------------------------------------------------------------------------------------------
template <int DT, int CT, int GT>
__global__ __launch_bounds__(256) void warp_corr_aggregate_kernel(const void* __restrict__ feat_, const float* __restrict__ hom,
                                                                  const float* __restrict__ hyp, const float* __restrict__ vis,
                                                                  float* __restrict__ vol, float* __restrict__ vis_sum, int normalise,
                                                                  int V, int C_, int G_, int D, int H, int W, int view_begin,
                                                                  int view_end, int nblk) {
    typedef typename FeatT<DT>::type T;
    const int C = CT > 0 ? CT : C_;
    const int G = GT > 0 ? GT : G_;
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int blk = (int)xcd_remap(blockIdx.x, (unsigned)nblk);
    const int p = blk * 64 + lane;
    const int b = (int)blockIdx.z;
    const bool valid = p < HW;
    const int pc = valid ? p : HW - 1;
    const int y = pc / W, x = pc - y * W;
    const float fx = (float)x, fy = (float)y;
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const T* feat = reinterpret_cast<const T*>(feat_);
    const T* ref = feat + (size_t)(b * V) * C * HW;
    float r[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < (CT > 0 ? CT : 1); ++c) r[c] = to_f32(ref[(size_t)c * HW + pc]);
    }
    float vsum = 0.0f;
    for (int v = view_begin; v < view_end; ++v) vsum += vis[(size_t)(b * (V - 1) + (v - 1)) * HW + pc];   // cost_volume.py:98
    if (vis_sum != nullptr && slot == 0 && valid) vis_sum[(size_t)b * HW + p] = vsum;
    const float denom = vsum + 1e-6f;                                                                     // cost_volume.py:101
    const float* hp = hyp + (size_t)b * D * HW + pc;
    for (int d = slot; d < D; d += 4) {
        const float depth = hp[(size_t)d * HW];
        float acc[GT > 0 ? GT : 64];
        for (int g = 0; g < G; ++g) acc[g] = 0.0f;
        for (int v = view_begin; v < view_end; ++v) {
            const Homography hm = load_homography(hom + (size_t)(b * (V - 1) + (v - 1)) * 12);
            const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
            const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
            const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
            const Taps tp = make_taps(hm, qx, qy, qz, depth, H, W, half_w, half_h, nullptr);
            const T* src = feat + (size_t)(b * V + v) * C * HW;
            float ip[GT > 0 ? GT : 64];
            correlate<T, CT, GT>(src, ref, r, tp, HW, pc, C, G, ip);
            const float w = vis[(size_t)(b * (V - 1) + (v - 1)) * HW + pc];
            for (int g = 0; g < G; ++g) acc[g] += ip[g] * w;                                             // cost_volume.py:97
        }
        if (valid) {
            float* o = vol + ((size_t)(b * D + d) * HW + p) * G;
            if (normalise) for (int g = 0; g < G; ++g) acc[g] = acc[g] / denom;
            if (GT == 8) {
                reinterpret_cast<float4*>(o)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                reinterpret_cast<float4*>(o)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            } else {
                for (int g = 0; g < G; ++g) o[g] = acc[g];
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
static int launch_entropy(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int C, int G, int D,
                          int H, int W, int vb, int ve, hipStream_t st) {
    const int HW = H * W;
    const int nblk = (int)ceil_div(HW, 64);
    const size_t lds = (size_t)D * 64 * sizeof(float);
    if (lds > 160 * 1024) { set_error("warp_corr_entropy: D=%d needs %zu B of LDS (> 160 KiB)", D, lds); return MVS_ERR_UNSUPPORTED; }
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_corr_entropy_kernel<DT, CT, GT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((warp_corr_entropy_kernel<DT, CT, GT>), dim3(nblk, ve - vb, B), dim3(256), lds, st, feat, hom, hyp, ent, V, C,
                       G, D, H, W, vb, nblk);
    return check_launch("warp_corr_entropy_kernel");
}

template <int DT, int CT, int GT>
static int launch_aggregate(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                            int normalise, int B, int V, int C, int G, int D, int H, int W, int vb, int ve, hipStream_t st) {
    const int nblk = (int)ceil_div((long long)H * W, 64);
    hipLaunchKernelGGL((warp_corr_aggregate_kernel<DT, CT, GT>), dim3(nblk, 1, B), dim3(256), 0, st, feat, hom, hyp, vis, vol, vis_sum,
                       normalise, V, C, G, D, H, W, vb, ve, nblk);
    return check_launch("warp_corr_aggregate_kernel");
}

#define MVS_DISPATCH_CG(FN, DT, ...)                                              \
    do {                                                                          \
        if (G == 8 && C == 64) return FN<DT, 64, 8>(__VA_ARGS__);                 \
        if (G == 8 && C == 32) return FN<DT, 32, 8>(__VA_ARGS__);                 \
        if (G == 8 && C == 16) return FN<DT, 16, 8>(__VA_ARGS__);                 \
        if (G == 8 && C == 8) return FN<DT, 8, 8>(__VA_ARGS__);                   \
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

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_compose_homography(const float* proj, int B, int V, float* homography, void* stream) {
    if (!proj || !homography || B < 1 || V < 2) { set_error("mvs_compose_homography: bad arguments"); return MVS_ERR_ARG; }
    const int n = B * (V - 1);
    hipLaunchKernelGGL(compose_homography_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, proj, B, V, homography);
    return check_launch("compose_homography_kernel");
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

extern "C" int mvs_warp_corr_entropy_fwd(const void* features, int dtype, const float* homography, const float* hyp, float* entropy,
                                         int B, int V, int C, int G, int D, int H, int W, int view_begin, int view_end, void* stream) {
    int rc = check_corr_args("mvs_warp_corr_entropy_fwd", features, homography, hyp, dtype, B, V, C, G, D, H, W, view_begin, view_end);
    if (rc != MVS_OK) return rc;
    if (!entropy) { set_error("mvs_warp_corr_entropy_fwd: null output"); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case MVS_DTYPE_F32: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_F32, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
        case MVS_DTYPE_BF16: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_BF16, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
        default: MVS_DISPATCH_CG(launch_entropy, MVS_DTYPE_F16, features, homography, hyp, entropy, B, V, C, G, D, H, W, view_begin, view_end, st);
    }
}

extern "C" int mvs_warp_corr_aggregate_fwd(const void* features, int dtype, const float* homography, const float* hyp, const float* vis,
                                           float* volume_cl, float* vis_sum, int normalise, int B, int V, int C, int G, int D, int H,
                                           int W, int view_begin, int view_end, void* stream) {
    int rc = check_corr_args("mvs_warp_corr_aggregate_fwd", features, homography, hyp, dtype, B, V, C, G, D, H, W, view_begin, view_end);
    if (rc != MVS_OK) return rc;
    if (!vis || !volume_cl) { set_error("mvs_warp_corr_aggregate_fwd: null pointer"); return MVS_ERR_ARG; }
    if (!normalise && !vis_sum) { set_error("mvs_warp_corr_aggregate_fwd: partial mode needs vis_sum"); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case MVS_DTYPE_F32: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_F32, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
        case MVS_DTYPE_BF16: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_BF16, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
        default: MVS_DISPATCH_CG(launch_aggregate, MVS_DTYPE_F16, features, homography, hyp, vis, volume_cl, vis_sum, normalise, B, V, C, G, D, H, W, view_begin, view_end, st);
    }
}

extern "C" int mvs_volume_normalise(float* volume_cl, const float* vis_sum, int B, int D, int H, int W, int G, void* stream) {
    if (!volume_cl || !vis_sum || B < 1 || D < 1 || H < 1 || W < 1 || G < 1) { set_error("mvs_volume_normalise: bad arguments"); return MVS_ERR_ARG; }
    const size_t total = (size_t)B * D * H * W * G;
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(volume_normalise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, volume_cl, vis_sum, D, H * W, G, total);
    return check_launch("volume_normalise_kernel");
}
