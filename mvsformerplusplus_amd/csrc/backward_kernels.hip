// Backward of the visibility-weighted cost-volume aggregation (first slice): the gradient of
//
//     volume[b,d,p,g] = sum_v sim_v[b,g,d,p] * vis_v[b,p] / (sum_v vis_v[b,p] + 1e-6)                  cost_volume.py:97-101
//     sim_v[b,g,d,p]  = mean_{c in group g} ref[b,c,p] * bilinear(src_v[b,c], grid_v(b,d,p))           cost_volume.py:74-87
//
// with respect to the reference features, the source features and the visibility maps.  What the reference's autograd does
// NOT differentiate is not differentiated here either: the sampling grid is built under torch.no_grad() (warping.py:80-97: no
// gradient reaches the depth hypotheses or the cameras) and the entropy that feeds the visibility CNN is computed from
// sim.detach() (cost_volume.py:90), so the only paths are volume -> sim -> features and volume -> vis.
//
// Like the forward passes the kernel never materialises a warped volume: it projects, gathers the four taps, rebuilds sim and
// scatters in one sweep (the reference keeps [B,C,D,H,W] per view alive for grid_sample's backward).
//
//   d volume / d sim_v = vis_v / den                              den = sum_v vis_v + 1e-6
//   d L / d vis_v[p]   = sum_{g,d} gvol[g,d,p] * (sim_v[g,d,p] - volume[g,d,p]) / den[p]
//   d L / d ref[c,p]   = sum_{v,d} gs_v[g(c),d,p] * warped_v[c,d,p]               gs_v = gvol * vis_v / (den * C/G)
//   d L / d src_v[c,q] = sum_{d,p} gs_v[g(c),d,p] * ref[c,p] * w_tap(q; d,p)      (atomic scatter, zero padding: taps outside
//                                                                                  the image have no weight, grid_sample's rule)
#include "mvs_common.h"

namespace mvs {

// One work-item per (reference pixel, source view, chunk of BW_DCH depth planes): the chunk's tap sets live in registers, every
// channel is gathered for all planes of the chunk back to back (as in the forward passes), the reference-feature gradient of a
// channel is accumulated over the chunk before ONE atomic add, and the four taps of every (channel, plane) are scattered with
// atomics (global_atomic_add_f32).  Measured on the MI355X at 2 x 512 x 640, V = 5, C = 8, D = 4: 5.5 ms against 0.11 ms for the
// forward - 335 M float atomics at 61 G/s; the kernel is bound by the L2 atomic units (a one-work-item-per-pixel form with 16 x
// less parallelism took the same 5.2 ms).  The remedy is accumulating each tile's scatter in an LDS window - the forward passes'
// window machinery - and flushing one atomic per window element (~9 x fewer); not built: the step this kernel sits in spends
// 400 ms in the 3-D convolutions' autograd.
constexpr int BW_DCH = 4;

template <typename T>
__global__ __launch_bounds__(256) void warp_corr_aggregate_bwd_kernel(const T* __restrict__ feat, const float* __restrict__ hom,
                                                                      const float* __restrict__ hyp, const float* __restrict__ vis,
                                                                      const float* __restrict__ vis_sum, const float* __restrict__ vol,
                                                                      const float* __restrict__ gvol, float* __restrict__ gfeat,
                                                                      float* __restrict__ gvis, int V, int C, int G, int D, int H, int W,
                                                                      int nchunk) {
    const int HW = H * W;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.z;
    const int v = 1 + (int)blockIdx.y / nchunk, d0 = ((int)blockIdx.y % nchunk) * BW_DCH;
    if (p >= HW) return;
    const int nd = D - d0 < BW_DCH ? D - d0 : BW_DCH;
    const int y = p / W, x = p - y * W;
    const int cpg = C / G;
    const float inv_cpg = 1.0f / (float)cpg;
    const float inv_den = 1.0f / (vis_sum[(size_t)b * HW + p] + 1e-6f);
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const float fx = (float)x, fy = (float)y;
    const T* ref = feat + (size_t)b * V * C * HW;
    float* gref = gfeat + (size_t)b * V * C * HW;
    Homography hm;
    {
        const float* hp = hom + ((size_t)b * (V - 1) + (v - 1)) * 12;
#pragma unroll
        for (int i = 0; i < 9; ++i) hm.r[i] = hp[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) hm.t[i] = hp[9 + i];
    }
    const float qx = hm.r[0] * fx + hm.r[1] * fy + hm.r[2];
    const float qy = hm.r[3] * fx + hm.r[4] * fy + hm.r[5];
    const float qz = hm.r[6] * fx + hm.r[7] * fy + hm.r[8];
    const T* src = ref + (size_t)v * C * HW;
    float* gsrc = gref + (size_t)v * C * HW;
    const float visv = vis[((size_t)b * (V - 1) + (v - 1)) * HW + p];
    Taps tp[BW_DCH];
#pragma unroll
    for (int dd = 0; dd < BW_DCH; ++dd) {
        const int d = d0 + (dd < nd ? dd : 0);
        tp[dd] = make_taps(hm, qx, qy, qz, hyp[((size_t)b * D + d) * HW + p], H, W, half_w, half_h, nullptr);
    }
    float gv = 0.0f;
    for (int g = 0; g < G; ++g) {
        float go[BW_DCH], gs[BW_DCH], sim[BW_DCH];
#pragma unroll
        for (int dd = 0; dd < BW_DCH; ++dd) {
            go[dd] = dd < nd ? gvol[(((size_t)b * D + d0 + dd) * HW + p) * G + g] : 0.0f;
            gs[dd] = go[dd] * visv * inv_den * inv_cpg;
            sim[dd] = 0.0f;
        }
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = g * cpg + cc;
            const T* sp = src + (size_t)c * HW;
            float* gp = gsrc + (size_t)c * HW;
            const float rc = to_f32(ref[(size_t)c * HW + p]);
            float gr = 0.0f;
#pragma unroll
            for (int dd = 0; dd < BW_DCH; ++dd) {
                if (dd >= nd) continue;
                float wv = tp[dd].w[0] * to_f32(sp[tp[dd].off[0]]);
                wv += tp[dd].w[1] * to_f32(sp[tp[dd].off[1]]);
                wv += tp[dd].w[2] * to_f32(sp[tp[dd].off[2]]);
                wv += tp[dd].w[3] * to_f32(sp[tp[dd].off[3]]);
                sim[dd] += rc * wv;
                gr += gs[dd] * wv;
                const float gw = gs[dd] * rc;
                if (gw != 0.0f) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (tp[dd].w[k] != 0.0f) atomicAdd(gp + tp[dd].off[k], gw * tp[dd].w[k]);
                }
            }
            atomicAdd(gref + (size_t)c * HW + p, gr);
        }
#pragma unroll
        for (int dd = 0; dd < BW_DCH; ++dd)
            if (dd < nd) gv += go[dd] * (sim[dd] * inv_cpg - vol[(((size_t)b * D + d0 + dd) * HW + p) * G + g]);
    }
    float* gvp = gvis + ((size_t)b * (V - 1) + (v - 1)) * HW + p;
    if (nchunk == 1) *gvp = gv * inv_den;
    else atomicAdd(gvp, gv * inv_den);
}

template <typename T>
static int launch_bwd(const void* feat, const float* hom, const float* hyp, const float* vis, const float* vis_sum, const float* vol,
                      const float* gvol, float* gfeat, float* gvis, int B, int V, int C, int G, int D, int H, int W, hipStream_t st) {
    const int nchunk = (D + BW_DCH - 1) / BW_DCH;
    if (nchunk > 1 && hipMemsetAsync(gvis, 0, (size_t)B * (V - 1) * H * W * sizeof(float), st) != hipSuccess) {
        set_error("mvs_warp_corr_aggregate_bwd: hipMemsetAsync failed");
        return MVS_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((warp_corr_aggregate_bwd_kernel<T>), dim3(ceil_div((long long)H * W, 256), (V - 1) * nchunk, B), dim3(256), 0, st,
                       reinterpret_cast<const T*>(feat), hom, hyp, vis, vis_sum, vol, gvol, gfeat, gvis, V, C, G, D, H, W, nchunk);
    return check_launch("warp_corr_aggregate_bwd_kernel");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_warp_corr_aggregate_bwd(const void* features, int dtype, const float* homography, const float* hyp, const float* vis,
                                           const float* vis_sum, const float* volume_cl, const float* grad_volume_cl, float* grad_features,
                                           float* grad_vis, int B, int V, int C, int G, int D, int H, int W, void* stream) {
    if (!features || !homography || !hyp || !vis || !vis_sum || !volume_cl || !grad_volume_cl || !grad_features || !grad_vis) {
        set_error("mvs_warp_corr_aggregate_bwd: null pointer");
        return MVS_ERR_ARG;
    }
    if (B < 1 || V < 2 || C < 1 || G < 1 || C % G != 0 || D < 1 || H < 1 || W < 1) { set_error("mvs_warp_corr_aggregate_bwd: bad shape"); return MVS_ERR_ARG; }
    if ((long long)V * C * H * W >= (1ll << 31)) { set_error("mvs_warp_corr_aggregate_bwd: V*C*H*W must stay below 2^31"); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    // reference-view and source-view gradients are accumulated in place
    if (hipMemsetAsync(grad_features, 0, (size_t)B * V * C * H * W * sizeof(float), st) != hipSuccess) {
        set_error("mvs_warp_corr_aggregate_bwd: hipMemsetAsync failed");
        return MVS_ERR_LAUNCH;
    }
    switch (dtype) {
        case MVS_DTYPE_F32: return launch_bwd<float>(features, homography, hyp, vis, vis_sum, volume_cl, grad_volume_cl, grad_features, grad_vis, B, V, C, G, D, H, W, st);
        case MVS_DTYPE_BF16: return launch_bwd<uint16_t>(features, homography, hyp, vis, vis_sum, volume_cl, grad_volume_cl, grad_features, grad_vis, B, V, C, G, D, H, W, st);
        case MVS_DTYPE_F16: return launch_bwd<_Float16>(features, homography, hyp, vis, vis_sum, volume_cl, grad_volume_cl, grad_features, grad_vis, B, V, C, G, D, H, W, st);
    }
    set_error("mvs_warp_corr_aggregate_bwd: unknown dtype %d", dtype);
    return MVS_ERR_ARG;
}
