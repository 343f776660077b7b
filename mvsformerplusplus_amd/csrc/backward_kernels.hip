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

// A block = a 16 x 16 tile of reference pixels x one source view x one chunk of BW_DCH depth planes; one work-item per pixel.
// The chunk's tap sets live in registers, every channel is gathered for all planes of the chunk back to back (as in the forward
// passes), the reference-feature gradient of a channel is accumulated over the chunk before ONE atomic add.
//
// Source-feature gradients: all taps of the tile fall into a small window of the source image (its bounding box is reduced with
// LDS atomics).  Four channels at a time, the scatter is accumulated in an LDS image of that window (ds_add_f32) and flushed with
// one global atomic per non-zero window element - ~8 x fewer global atomics than four per (pixel, plane, channel), which is
// what bounded the first form of this kernel (335 M global_atomic_add_f32 at 61 G/s = 5.5 ms at 2 x 512 x 640, V = 5, C = 8, D = 4,
// against 0.11 ms for the forward; with the LDS image 3.7 ms, with the forward passes' 2 x 2-block tap sets 1.8 ms).  A tile whose
// window exceeds the LDS image (steep geometry) scatters straight to global.
constexpr int BW_DCH = 4;
constexpr int BW_TILE = 16;
constexpr int BW_CAP = 2048;                  // window positions of the LDS image: 2048 x 4 channels x 4 B = 32 KiB

template <typename T>
__global__ __launch_bounds__(256) void warp_corr_aggregate_bwd_kernel(const T* __restrict__ feat, const float* __restrict__ hom,
                                                                      const float* __restrict__ hyp, const float* __restrict__ vis,
                                                                      const float* __restrict__ vis_sum, const float* __restrict__ vol,
                                                                      const float* __restrict__ gvol, float* __restrict__ gfeat,
                                                                      float* __restrict__ gvis, int V, int C, int G, int D, int H, int W,
                                                                      int nchunk, int tiles_x) {
    __shared__ float accw[BW_CAP * 4];
    __shared__ int box[4];
    const int HW = H * W;
    const int tid = (int)threadIdx.x;
    const int tx = (int)blockIdx.x % tiles_x, ty = (int)blockIdx.x / tiles_x;
    const int x = tx * BW_TILE + (tid & 15), y = ty * BW_TILE + (tid >> 4);
    const bool inside = x < W && y < H;
    const int p = inside ? y * W + x : 0;
    const int b = (int)blockIdx.z;
    const int v = 1 + (int)blockIdx.y / nchunk, d0 = ((int)blockIdx.y % nchunk) * BW_DCH;
    const int nd = D - d0 < BW_DCH ? D - d0 : BW_DCH;
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
    if (tid == 0) { box[0] = 0x7fffffff; box[1] = 0x7fffffff; box[2] = -1; box[3] = -1; }
    __syncthreads();
    // tap sets as in the forward passes: one 2 x 2 block of in-bounds source pixels per plane, weights routed to its four slots
    GTap tp[BW_DCH];
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
#pragma unroll
    for (int dd = 0; dd < BW_DCH; ++dd) {
        const int d = d0 + (dd < nd ? dd : 0);
        tp[dd] = make_gtap(hm, qx, qy, qz, hyp[((size_t)b * D + d) * HW + p], H, W, cx, cy);
        if (!inside || dd >= nd) tp[dd].pk = GL_NONE;                     // this work-item contributes nothing
        if (tp[dd].pk != GL_NONE) {
            const int xb = (int)(tp[dd].pk & 0xffffu), yb = (int)(tp[dd].pk >> 16);
            atomicMin(&box[0], xb); atomicMin(&box[1], yb); atomicMax(&box[2], xb + 1); atomicMax(&box[3], yb + 1);
        } else {
            tp[dd].w00 = 0.0f; tp[dd].w01 = 0.0f; tp[dd].w10 = 0.0f; tp[dd].w11 = 0.0f;
        }
    }
    __syncthreads();
    const int xmin = box[0], ymin = box[1], ww = box[2] - xmin + 1, wh = box[3] - ymin + 1;
    const bool use_lds = box[2] >= 0 && ww * wh <= BW_CAP;
    int goff[BW_DCH], lpos[BW_DCH];                                        // top-left slot: element offset in the source map / in the LDS image
#pragma unroll
    for (int dd = 0; dd < BW_DCH; ++dd) {
        const int xb = tp[dd].pk == GL_NONE ? 0 : (int)(tp[dd].pk & 0xffffu), yb = tp[dd].pk == GL_NONE ? 0 : (int)(tp[dd].pk >> 16);
        goff[dd] = yb * W + xb;
        lpos[dd] = tp[dd].pk == GL_NONE ? 0 : (yb - ymin) * ww + (xb - xmin);
    }
    float gv = 0.0f;
    float go[BW_DCH], gs[BW_DCH], sim[BW_DCH];
    int cur_g = -1;
    for (int c4 = 0; c4 < C; c4 += 4) {
        if (use_lds) {
            for (int i = tid; i < ww * wh; i += 256) {
#pragma unroll
                for (int j = 0; j < 4; ++j) accw[j * BW_CAP + i] = 0.0f;
            }
            __syncthreads();
        }
        for (int j = 0; j < 4 && c4 + j < C; ++j) {
            const int c = c4 + j, g = c / cpg;
            if (g != cur_g) {
                if (cur_g >= 0) {
#pragma unroll
                    for (int dd = 0; dd < BW_DCH; ++dd)
                        if (dd < nd) gv += go[dd] * (sim[dd] * inv_cpg - vol[(((size_t)b * D + d0 + dd) * HW + p) * G + cur_g]);
                }
                cur_g = g;
#pragma unroll
                for (int dd = 0; dd < BW_DCH; ++dd) {
                    go[dd] = (inside && dd < nd) ? gvol[(((size_t)b * D + d0 + dd) * HW + p) * G + g] : 0.0f;
                    gs[dd] = go[dd] * visv * inv_den * inv_cpg;
                    sim[dd] = 0.0f;
                }
            }
            const T* sp = src + (size_t)c * HW;
            float* gp = gsrc + (size_t)c * HW;
            float* lp = accw + j * BW_CAP;                                  // one LDS plane per channel: neighbouring positions, neighbouring banks
            const float rc = to_f32(ref[(size_t)c * HW + p]);
            float gr = 0.0f;
#pragma unroll
            for (int dd = 0; dd < BW_DCH; ++dd) {
                const T* q = sp + goff[dd];
                float wv = tp[dd].w00 * to_f32(q[0]);
                wv += tp[dd].w01 * to_f32(q[1]);
                wv += tp[dd].w10 * to_f32(q[W]);
                wv += tp[dd].w11 * to_f32(q[W + 1]);
                sim[dd] += rc * wv;
                gr += gs[dd] * wv;
                const float gw = gs[dd] * rc;
                if (gw != 0.0f) {
                    const float w4[4] = {tp[dd].w00, tp[dd].w01, tp[dd].w10, tp[dd].w11};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (w4[k] != 0.0f) {
                            if (use_lds) atomicAdd(lp + lpos[dd] + (k >> 1) * ww + (k & 1), gw * w4[k]);
                            else atomicAdd(gp + goff[dd] + (k >> 1) * W + (k & 1), gw * w4[k]);
                        }
                }
            }
            if (inside && gr != 0.0f) atomicAdd(gref + (size_t)c * HW + p, gr);
        }
        if (use_lds) {
            __syncthreads();
            for (int i = tid; i < ww * wh; i += 256) {
                const int wy = i / ww, wx = i - wy * ww;
                float* dst = gsrc + (size_t)c4 * HW + (size_t)(ymin + wy) * W + xmin + wx;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float val = accw[j * BW_CAP + i];
                    if (val != 0.0f) atomicAdd(dst + (size_t)j * HW, val);
                }
            }
            __syncthreads();
        }
    }
    if (cur_g >= 0) {
#pragma unroll
        for (int dd = 0; dd < BW_DCH; ++dd)
            if (dd < nd) gv += go[dd] * (sim[dd] * inv_cpg - vol[(((size_t)b * D + d0 + dd) * HW + p) * G + cur_g]);
    }
    if (inside) {
        float* gvp = gvis + ((size_t)b * (V - 1) + (v - 1)) * HW + p;
        if (nchunk == 1) *gvp = gv * inv_den;
        else atomicAdd(gvp, gv * inv_den);
    }
}

template <typename T>
static int launch_bwd(const void* feat, const float* hom, const float* hyp, const float* vis, const float* vis_sum, const float* vol,
                      const float* gvol, float* gfeat, float* gvis, int B, int V, int C, int G, int D, int H, int W, hipStream_t st) {
    const int nchunk = (D + BW_DCH - 1) / BW_DCH;
    if (nchunk > 1 && hipMemsetAsync(gvis, 0, (size_t)B * (V - 1) * H * W * sizeof(float), st) != hipSuccess) {
        set_error("mvs_warp_corr_aggregate_bwd: hipMemsetAsync failed");
        return MVS_ERR_LAUNCH;
    }
    const int tiles_x = (int)ceil_div(W, BW_TILE), tiles_y = (int)ceil_div(H, BW_TILE);
    hipLaunchKernelGGL((warp_corr_aggregate_bwd_kernel<T>), dim3(tiles_x * tiles_y, (V - 1) * nchunk, B), dim3(256), 0, st,
                       reinterpret_cast<const T*>(feat), hom, hyp, vis, vis_sum, vol, gvol, gfeat, gvis, V, C, G, D, H, W, nchunk, tiles_x);
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
