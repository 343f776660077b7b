// Training-mode kernels of the 3-D U-Net regulariser: batch-statistics BatchNorm (forward and
// backward) fused with ReLU and the skip add, and the weight gradient of the (transposed) convolutions on the fp32 MFMA path.
// Reference behaviour: nn.BatchNorm3d in train mode + ReLU inside Conv3d / Deconv3d (module.py:89-165), the skip adds of
// CostRegNet / CostRegNet3D (module.py:398-408, 494-504), autograd's convolution_backward for the weights.
//
// The forward convolutions and the data gradients reuse the inference kernels (conv_bf16x3_kernels.hip) with un-folded,
// re-packed weights: the data gradient of a stride-1 convolution is the convolution with flipped, transposed taps; of a strided
// convolution the transposed convolution with the same taps; of a transposed convolution the strided convolution (training.py).
//
// Everything is channel-last fp32 [N voxels][C], C in {8, 16, 32, 64}.
#include "conv_cfg.h"

namespace mvs {

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics: per-channel sum and sum of squares in double (N reaches 10^7 voxels).
// sums[0..C) = sum x, sums[C..2C) = sum x^2 (zeroed by the entry point, accumulated with one f64 atomic per block and value).
// ------------------------------------------------------------------------------------------------
// Every BatchNorm kernel takes `groups` independent statistics sets along blockIdx.y (the visibility CNN normalises each source
// view's batch on its own, like the reference's per-view calls): group g owns voxels [g * N, (g + 1) * N) and row g of the
// [groups][...] statistics arrays.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, size_t n4, int C) {
    __shared__ double part[2 * 64];
    const int tid = (int)threadIdx.x;
    x += (size_t)blockIdx.y * n4 * 4;
    sums += (size_t)blockIdx.y * 2 * C;
    if (tid < 2 * C) part[tid] = 0.0;
    __syncthreads();
    const int q4 = C / 4;                                                  // float4s per voxel; gridDim.x * 256 is a multiple of it
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + tid;
    const int c0 = (int)(i % (size_t)q4) * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        atomicAdd(&part[c0 + k], s[k]);
        atomicAdd(&part[C + c0 + k], ss[k]);
    }
    __syncthreads();
    if (tid < 2 * C) atomicAdd(&sums[tid], part[tid]);
}

// mean / biased variance / 1 / sqrt(var + eps) from the sums (count = voxels summed, possibly over several ranks); with
// running_mean != NULL also nn.BatchNorm's momentum step of the running statistics (unbiased variance)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float* __restrict__ mean, float* __restrict__ var,
                                   float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, int C, int groups) {
    const int c = (int)threadIdx.x;
    if (c >= C) return;
    for (int g = 0; g < groups; ++g) {                                     // in order: the running statistics take one momentum step per group
        const double m = sums[(size_t)g * 2 * C + c] / count;
        double v = sums[(size_t)g * 2 * C + C + c] / count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean[g * C + c] = (float)m;
        var[g * C + c] = (float)v;
        invstd[g * C + c] = (float)(1.0 / sqrt(v + (double)eps));
        if (running_mean != nullptr) {
            const float unbiased = (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

// the momentum step alone (the reference's checkpoint recomputation repeats it in the backward pass)
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ var, double count, float momentum,
                                         float* __restrict__ running_mean, float* __restrict__ running_var, int C, int groups) {
    const int c = (int)threadIdx.x;
    if (c >= C) return;
    for (int g = 0; g < groups; ++g) {
        const float unbiased = (float)((double)var[g * C + c] * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean[g * C + c];
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// y = relu((z - mean) * invstd * gamma + beta) [+ skip]        (relu = 0: no clamp)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ skip, float* __restrict__ y, size_t n4, int C, int relu) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t goff = (size_t)blockIdx.y * n4 * 4;
    z += goff; y += goff; mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C;
    if (skip != nullptr) skip += goff;
    const int c0 = (int)(i % (size_t)(C / 4)) * 4;
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    const float in[4] = {v.x, v.y, v.z, v.w};
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sc = invstd[c0 + k] * gamma[c0 + k];
        float t = (in[k] - mean[c0 + k]) * sc + beta[c0 + k];
        if (relu) t = fmaxf(t, 0.0f);
        out[k] = t;
    }
    if (skip != nullptr) {
        const float4 s = reinterpret_cast<const float4*>(skip)[i];
        out[0] += s.x; out[1] += s.y; out[2] += s.z; out[3] += s.w;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(out[0], out[1], out[2], out[3]);
}

// backward, phase 1: g = dy * [bn(z) > 0] (relu) ; sums[0..C) = sum g (= d beta), sums[C..2C) = sum g * xhat (= d gamma)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, double* __restrict__ sums, size_t n4, int C, int relu) {
    __shared__ double part[2 * 64];
    const int tid = (int)threadIdx.x;
    {
        const size_t goff = (size_t)blockIdx.y * n4 * 4;
        dy += goff; z += goff; mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C; sums += (size_t)blockIdx.y * 2 * C;
    }
    if (tid < 2 * C) part[tid] = 0.0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + tid;
    const int c0 = (int)(i % (size_t)(C / 4)) * 4;
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mu[k] = mean[c0 + k]; is[k] = invstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; }
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
    for (; i < n4; i += stride) {
        const float4 zv = reinterpret_cast<const float4*>(z)[i], gv = reinterpret_cast<const float4*>(dy)[i];
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (zz[k] - mu[k]) * is[k];
            // the ReLU mask is rebuilt with the forward's own expression, rounding for rounding (bn_apply_kernel): t = (z - mean) * (invstd * gamma) + beta
            const float g = (!relu || (zz[k] - mu[k]) * (is[k] * ga[k]) + be[k] > 0.0f) ? gg[k] : 0.0f;
            s[k] += g;
            sx[k] += (double)g * xh;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        atomicAdd(&part[c0 + k], s[k]);
        atomicAdd(&part[C + c0 + k], sx[k]);
    }
    __syncthreads();
    if (tid < 2 * C) atomicAdd(&sums[tid], part[tid]);
}

// backward, phase 2: dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))   (batch statistics)
//                    dz = gamma * invstd * g                                        (use_batch_stats = 0: running statistics)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const double* __restrict__ sums, double count,
                                                           float* __restrict__ dz, size_t n4, int C, int relu, int use_batch_stats) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    {
        const size_t goff = (size_t)blockIdx.y * n4 * 4;
        dy += goff; z += goff; dz += goff; mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C; sums += (size_t)blockIdx.y * 2 * C;
    }
    const int c0 = (int)(i % (size_t)(C / 4)) * 4;
    const float4 zv = reinterpret_cast<const float4*>(z)[i], gv = reinterpret_cast<const float4*>(dy)[i];
    const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float is = invstd[c0 + k], ga = gamma[c0 + k];
        const float xh = (zz[k] - mean[c0 + k]) * is;
        const float g = (!relu || (zz[k] - mean[c0 + k]) * (is * ga) + beta[c0 + k] > 0.0f) ? gg[k] : 0.0f;      // the forward's expression (bn_apply_kernel)
        float t = g;
        if (use_batch_stats) t -= (float)(sums[c0 + k] / count) + xh * (float)(sums[C + c0 + k] / count);
        out[k] = t * ga * is;
    }
    reinterpret_cast<float4*>(dz)[i] = make_float4(out[0], out[1], out[2], out[3]);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of Conv3d(k3, padding 1, stride (SD,SH,SW)):
//     dW[co][ci][tap] = sum over output voxels o of  g[o][co] * a[o * stride + tap - 1][ci]
// as 27 GEMMs (one per tap) with M = co, N = ci, K = output voxels on v_mfma_f32_16x16x4_f32 (exact fp32 products).
// A block owns one 16 x 16 (co, ci) block (blockIdx.y) and walks a run of output tiles; per tile it stages 16 channels of the
// gradient tile and of the input halo tile in LDS; wave w accumulates taps w, w + 4, ... (7, 7, 7, 6) in registers over ALL its
// tiles and adds them to dW once at the end (dW zeroed by the entry point).  The weight gradient of a transposed convolution
// is the same sum with the roles of input and output exchanged (a = its output gradient, g = its input).
// ------------------------------------------------------------------------------------------------
template <int SD_, int SH_, int SW_, int TD_, int TH_, int KD_ = 3>
struct WgCfg {
    static constexpr int SD = SD_, SH = SH_, SW = SW_, TD = TD_, TH = TH_, TW = 16, KD = KD_;
    static constexpr int NTAP = KD * 9, NT = (NTAP + 3) / 4;               // taps in all, taps per wave (the last waves may own one less)
    static constexpr int ID = (TD - 1) * SD + KD, IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3;
    static constexpr int NVOX = ID * IH * IW, NVO = TD * TH * TW;
    static constexpr size_t LDS_BYTES = (size_t)(NVOX + NVO) * 16 * sizeof(float);
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ dw, int D,
                                                           int H, int W, int OD, int OH, int OW, int CA, int CB, int tiles_x, int tiles_y,
                                                           int tiles_per_batch, int ntiles) {
    constexpr int SD = Cfg::SD, SH = Cfg::SH, SW = Cfg::SW, TD = Cfg::TD, TH = Cfg::TH, IH = Cfg::IH, IW = Cfg::IW;
    HIP_DYNAMIC_SHARED(float4, lds4)
    float* la = reinterpret_cast<float*>(lds4);                            // [NVOX][16] input halo tile
    float* lg = la + Cfg::NVOX * 16;                                       // [NVO][16] output-gradient tile
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, grp = lane >> 4;
    const int nbn = (CA + 15) / 16;
    const int co0 = ((int)blockIdx.y / nbn) * 16, ci0 = ((int)blockIdx.y % nbn) * 16;
    const int nblk = (int)gridDim.x, per = (ntiles + nblk - 1) / nblk;
    const int t_begin = (int)blockIdx.x * per;
    const int t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
    if (t_begin >= t_end) return;

    constexpr int NT = Cfg::NT, NTAP = Cfg::NTAP;                          // k3: 7 taps per wave (wave 3: six); k = (1,3,3): 3, 2, 2, 2
    f32x4 acc[NT];
    int tapoff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        const int tap = wave + 4 * j < NTAP ? wave + 4 * j : NTAP - 1;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        tapoff[j] = ((kd * IH + kh) * IW + kw) * 16;
    }
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_per_batch;
        int t = tile - b * tiles_per_batch;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y, tz = t / tiles_y;
        const int oz0 = tz * TD, oy0 = ty * TH, ox0 = tx * 16;
        const int iz0 = oz0 * SD - Cfg::KD / 2, iy0 = oy0 * SH - 1, ix0 = ox0 * SW - 1;
        const float* ab = a + (size_t)b * D * H * W * CA;
        const float* gb = g + (size_t)b * OD * OH * OW * CB;
        for (int e = tid; e < Cfg::NVOX * 4; e += 256) {
            const int vox = e >> 2, q = e & 3;
            const int dx = vox % IW;
            const int t2 = vox / IW;
            const int dy = t2 % IH, dz = t2 / IH;
            const int zz = iz0 + dz, yy = iy0 + dy, xx = ix0 + dx, c = ci0 + 4 * q;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W && c < CA)
                v = *reinterpret_cast<const float4*>(ab + (((size_t)zz * H + yy) * W + xx) * CA + c);
            *reinterpret_cast<float4*>(la + vox * 16 + 4 * q) = v;
        }
        for (int e = tid; e < Cfg::NVO * 4; e += 256) {
            const int vox = e >> 2, q = e & 3;
            const int ox = vox & 15, row = vox >> 4;
            const int oz = oz0 + row / TH, oy = oy0 + row % TH, c = co0 + 4 * q;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (oz < OD && oy < OH && ox0 + ox < OW && c < CB)
                v = *reinterpret_cast<const float4*>(gb + (((size_t)oz * OH + oy) * OW + ox0 + ox) * CB + c);
            *reinterpret_cast<float4*>(lg + vox * 16 + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int k4 = 0; k4 < Cfg::NVO / 4; ++k4) {
            // the four voxels of this k-group sit in one row of 16: x = 4 * (k4 % 4) + grp
            const int row = k4 >> 2, ox = 4 * (k4 & 3) + grp;
            const int oz = row / TH, oy = row % TH;
            const float aop = lg[(row * 16 + ox) * 16 + li];              // A operand: [m = co][k = voxel]
            const float* bp = la + (((oz * SD) * IH + oy * SH) * IW + ox * SW) * 16 + li;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j == NT - 1 && wave + 4 * j >= NTAP) continue;         // the last round of taps is not full
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bp[tapoff[j]], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tap = wave + 4 * j;
        if (tap >= NTAP) continue;
        const int ci = ci0 + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 4 * grp + i;
            if (co < CB && ci < CA) atomicAdd(dw + ((size_t)co * CA + ci) * NTAP + tap, acc[j][i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight packing on the device (the training path re-packs every weight twice per iteration - forward form and data-gradient
// form; as PyTorch ops that was ~12 small launches per layer and direction).  Same layouts, bit for bit, as packing.py:
//   conv   packed[pass][step][mb][hi|lo][g][j][e] = W'[16mb + j][pass*CH + 8oc + e][tap], (tap, oc) = divmod(4 step + g, CH / 8)
//          tflip = 1: W'[co][ci][tap] = W[ci][co][ntap - 1 - tap]  (the data-gradient form of a stride-1 convolution)
//   deconv per parity class (pair of classes for Cout = 8), packing.pack_deconv_weights_bf16x3
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_split_bf16(float v, uint16_t* hi, uint16_t* lo) {
    const uint16_t h = from_f32<uint16_t>(v);
    *hi = h;
    *lo = from_f32<uint16_t>(v - to_f32(h));
}

__global__ __launch_bounds__(256) void pack_conv_bf16x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, int ntap,
                                                               int ch, int tflip, int mrep, int nstep, int total) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 7, j = (idx >> 3) & 15, g = (idx >> 7) & 3;
    int r = idx >> 9;
    const int mb = r % mrep;
    r /= mrep;
    const int step = r % nstep, pass = r / nstep;
    const int opt = ch / 8, o = 4 * step + g, tap = o / opt, oc = o - tap * opt;
    const int co = 16 * mb + j, ci = pass * ch + 8 * oc + e;
    float v = 0.0f;
    if (tap < ntap && co < cout) v = tflip ? w[((size_t)ci * cout + co) * ntap + (ntap - 1 - tap)] : w[((size_t)co * cin + ci) * ntap + tap];
    uint16_t* base = out + (((size_t)(pass * nstep + step) * mrep + mb) * 2) * 512 + (g * 16 + j) * 8 + e;
    store_split_bf16(v, base, base + 512);
}

struct DeconvPackPlan {
    int nchunk;
    int cls[8], ntap[8], nst[8], off[9];          // off in units of 1024 * mrep elements (one step of one chunk)
};

__global__ __launch_bounds__(256) void pack_deconv_bf16x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cin, int cout, int sd,
                                                                 int mrep, DeconvPackPlan plan, int total) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 7, j = (idx >> 3) & 15, g = (idx >> 7) & 3;
    int r = idx >> 9;
    const int mb = r % mrep, gstep = r / mrep;                              // step index over all chunks
    int c = 0;
    while (c + 1 < plan.nchunk && gstep >= plan.off[c + 1]) ++c;
    const int step = gstep - plan.off[c];
    const int opt = cin / 8, o = 4 * step + g, ti = o / opt, oc = o - ti * opt;
    const int cls = plan.cls[c];
    const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (sd == 2) ? (cls >> 2) : 0;
    const int nkw = pw ? 2 : 1, nkh = ph ? 2 : 1;
    const int a_w = ti % nkw, a_h = (ti / nkw) % nkh, a_d = ti / (nkw * nkh);
    const int kd = (sd == 2) ? (pd ? 2 * a_d : 1) : a_d, kh = ph ? 2 * a_h : 1, kw = pw ? 2 * a_w : 1;
    const int tap = (kd * 3 + kh) * 3 + kw;
    const int cig = 8 * oc + e;
    float v = 0.0f;
    if (ti < plan.ntap[c]) {
        if (cout == 8) {                                                   // rows 8-15: pw = 1 taps; rows 0-7: the pw = 0 tap sharing input mx
            if (j >= 8) v = w[((size_t)cig * 8 + (j - 8)) * 27 + tap];
            else if (kw == 2) v = w[((size_t)cig * 8 + j) * 27 + tap - 1];
        } else if (16 * mb + j < cout) {
            v = w[((size_t)cig * cout + 16 * mb + j) * 27 + tap];
        }
    }
    uint16_t* base = out + (((size_t)gstep * mrep + mb) * 2) * 512 + (g * 16 + j) * 8 + e;
    store_split_bf16(v, base, base + 512);
}

// d gamma / d beta as fp32 from the (per group) backward sums
__global__ void bn_param_grads_kernel(const double* __restrict__ sums, int groups, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = (int)threadIdx.x;
    if (c >= C) return;
    double b = 0.0, g = 0.0;
    for (int k = 0; k < groups; ++k) { b += sums[(size_t)k * 2 * C + c]; g += sums[(size_t)k * 2 * C + C + c]; }
    dbeta[c] = (float)b;
    dgamma[c] = (float)g;
}

template <class Cfg>
static int launch_wgrad(const float* a, const float* g, float* dw, int B, int D, int H, int W, int OD, int OH, int OW, int CA, int CB, hipStream_t st) {
    const int tx = (int)ceil_div(OW, 16), ty = (int)ceil_div(OH, Cfg::TH), tz = (int)ceil_div(OD, Cfg::TD);
    const int per_batch = tx * ty * tz, ntiles = per_batch * B;
    const int jobs = (int)ceil_div(CA, 16) * (int)ceil_div(CB, 16);
    if (Cfg::LDS_BYTES > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    // one chip-load of persistent blocks (256 CUs x 2 resident): every block ends with 27 x 256 atomic adds into ITS (co, ci) block
    // of dW, so the number of blocks per job is the depth of the same-address atomic chains (2048 blocks per job: 14 M atomics on
    // 6912 addresses, 700 us per launch on the MI355X; 512 in total: measured below)
    int nblk = 512 / jobs;
    nblk = nblk < 1 ? 1 : nblk;
    nblk = nblk > ntiles ? ntiles : nblk;
    hipLaunchKernelGGL((conv3d_wgrad_kernel<Cfg>), dim3(nblk, jobs), dim3(256), Cfg::LDS_BYTES, st, a, g, dw, D, H, W, OD, OH, OW, CA, CB, tx, ty,
                       per_batch, ntiles);
    return check_launch("conv3d_wgrad_kernel");
}

static unsigned ew_blocks(size_t n4, int C, unsigned cap) {
    // a multiple of C / 4 threads in total (each work-item then stays on one channel quad): 256 * blocks always is
    size_t b = (n4 + 255) / 256;
    if (cap && b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

static bool bn_shape_ok(const char* who, size_t N, int C) {
    if (N < 1 || (C != 8 && C != 16 && C != 32 && C != 64)) { set_error("%s: C must be 8, 16, 32 or 64 and N >= 1", who); return false; }
    return true;
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_bn_stats(const float* x_cl, double* sums, long long N, int C, int groups, void* stream) {
    if (!x_cl || !sums || groups < 1 || !bn_shape_ok("mvs_bn_stats", (size_t)N, C)) { if (!x_cl || !sums || groups < 1) set_error("mvs_bn_stats: bad arguments"); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, (size_t)groups * 2 * C * sizeof(double), st) != hipSuccess) { set_error("mvs_bn_stats: hipMemsetAsync failed"); return MVS_ERR_LAUNCH; }
    const size_t n4 = (size_t)N * C / 4;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(ew_blocks(n4, C, 2048), groups), dim3(256), 0, st, x_cl, sums, n4, C);
    return check_launch("bn_stats_kernel");
}

extern "C" int mvs_bn_finalize(const double* sums, double count, float eps, float* mean, float* var, float* invstd, float* running_mean,
                               float* running_var, float momentum, int C, int groups, void* stream) {
    if (!sums || !mean || !var || !invstd || count < 1.0 || groups < 1 || !bn_shape_ok("mvs_bn_finalize", 1, C) ||
        ((running_mean == nullptr) != (running_var == nullptr))) {
        set_error("mvs_bn_finalize: bad arguments");
        return MVS_ERR_ARG;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, count, eps, mean, var, invstd, running_mean, running_var,
                       momentum, C, groups);
    return check_launch("bn_finalize_kernel");
}

extern "C" int mvs_bn_running_update(const float* mean, const float* var, double count, float momentum, float* running_mean, float* running_var, int C,
                                     int groups, void* stream) {
    if (!mean || !var || !running_mean || !running_var || count < 1.0 || groups < 1 || !bn_shape_ok("mvs_bn_running_update", 1, C)) { set_error("mvs_bn_running_update: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mean, var, count, momentum, running_mean, running_var, C, groups);
    return check_launch("bn_running_update_kernel");
}

extern "C" int mvs_bn_relu_apply(const float* z_cl, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* skip_cl,
                                 float* y_cl, long long N, int C, int relu, int groups, void* stream) {
    if (!z_cl || !mean || !invstd || !gamma || !beta || !y_cl || groups < 1 || !bn_shape_ok("mvs_bn_relu_apply", (size_t)N, C)) { set_error("mvs_bn_relu_apply: bad arguments"); return MVS_ERR_ARG; }
    const size_t n4 = (size_t)N * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4, C, 0), groups), dim3(256), 0, (hipStream_t)stream, z_cl, mean, invstd, gamma, beta, skip_cl, y_cl, n4, C, relu);
    return check_launch("bn_apply_kernel");
}

extern "C" int mvs_bn_relu_bwd(const float* dy_cl, const float* z_cl, const float* mean, const float* invstd, const float* gamma, const float* beta,
                               double* sums, double count, float* dz_cl, long long N, int C, int relu, int use_batch_stats, int phase, int groups,
                               void* stream) {
    if (!dy_cl || !z_cl || !mean || !invstd || !gamma || !beta || !sums || groups < 1 || !bn_shape_ok("mvs_bn_relu_bwd", (size_t)N, C)) { set_error("mvs_bn_relu_bwd: bad arguments"); return MVS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = (size_t)N * C / 4;
    if (phase == 0) {                                                      // reduce: sums = [d beta | d gamma] of THIS rank's voxels
        if (hipMemsetAsync(sums, 0, (size_t)groups * 2 * C * sizeof(double), st) != hipSuccess) { set_error("mvs_bn_relu_bwd: hipMemsetAsync failed"); return MVS_ERR_LAUNCH; }
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ew_blocks(n4, C, 2048), groups), dim3(256), 0, st, dy_cl, z_cl, mean, invstd, gamma, beta, sums, n4, C, relu);
        return check_launch("bn_bwd_reduce_kernel");
    }
    if (!dz_cl || count < 1.0) { set_error("mvs_bn_relu_bwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n4, C, 0), groups), dim3(256), 0, st, dy_cl, z_cl, mean, invstd, gamma, beta, sums, count, dz_cl, n4, C, relu,
                       use_batch_stats);
    return check_launch("bn_bwd_apply_kernel");
}

extern "C" long long mvs_pack_conv_weights_elems(int cout, int cin, int ntap, int ch) {
    if (cout < 1 || cin < 1 || ntap < 1 || ch < 8 || (ch % 8) || (cin % ch)) return -1;
    const int opt = ch / 8, nstep = (ntap * opt + 3) / 4, mrep = (cout + 15) / 16;
    return (long long)(cin / ch) * nstep * mrep * 1024;
}

extern "C" int mvs_pack_conv_weights(const float* w, void* packed_bf16, int cout, int cin, int ntap, int ch, int tflip, void* stream) {
    const long long total2 = mvs_pack_conv_weights_elems(cout, cin, ntap, ch);
    if (!w || !packed_bf16 || total2 < 0) { set_error("mvs_pack_conv_weights: bad arguments"); return MVS_ERR_ARG; }
    const int opt = ch / 8, nstep = (ntap * opt + 3) / 4, mrep = (cout + 15) / 16, total = (int)(total2 / 2);
    hipLaunchKernelGGL(pack_conv_bf16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, static_cast<uint16_t*>(packed_bf16), cout,
                       cin, ntap, ch, tflip ? 1 : 0, mrep, nstep, total);
    return check_launch("pack_conv_bf16x3_kernel");
}

static bool deconv_plan(int cin, int cout, int sd, DeconvPackPlan* plan, int* mrep, long long* elems) {
    if (cin < 8 || (cin % 8) || cout < 1 || (sd != 1 && sd != 2)) return false;
    const int opt = cin / 8, ncls = (sd == 2 ? 2 : 1) * 4;
    *mrep = (cout + 15) / 16;
    plan->nchunk = 0;
    plan->off[0] = 0;
    for (int cls = 0; cls < ncls; ++cls) {
        if (cout == 8 && !(cls & 1)) continue;                               // Cout = 8: one chunk per (pd, ph) pair, the taps of its pw = 1 class
        const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (sd == 2) ? (cls >> 2) : 0;
        const int ntap = ((sd == 2) ? (pd ? 2 : 1) : 3) * (ph ? 2 : 1) * (pw ? 2 : 1);
        const int c = plan->nchunk++;
        plan->cls[c] = cls;
        plan->ntap[c] = ntap;
        plan->nst[c] = (ntap * opt + 3) / 4;
        plan->off[c + 1] = plan->off[c] + plan->nst[c];
    }
    *elems = (long long)plan->off[plan->nchunk] * (*mrep) * 1024;
    return true;
}

extern "C" long long mvs_pack_deconv_weights_elems(int cin, int cout, int sd) {
    DeconvPackPlan plan;
    int mrep;
    long long elems;
    return deconv_plan(cin, cout, sd, &plan, &mrep, &elems) ? elems : -1;
}

extern "C" int mvs_pack_deconv_weights(const float* w, void* packed_bf16, int cin, int cout, int sd, void* stream) {
    DeconvPackPlan plan;
    int mrep;
    long long elems;
    if (!w || !packed_bf16 || !deconv_plan(cin, cout, sd, &plan, &mrep, &elems)) { set_error("mvs_pack_deconv_weights: bad arguments"); return MVS_ERR_ARG; }
    const int total = (int)(elems / 2);
    hipLaunchKernelGGL(pack_deconv_bf16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, static_cast<uint16_t*>(packed_bf16), cin,
                       cout, sd, mrep, plan, total);
    return check_launch("pack_deconv_bf16x3_kernel");
}

extern "C" int mvs_conv3d_wgrad(const float* a_cl, const float* g_cl, float* dw, int B, int CA, int CB, int D, int H, int W, int kd, int sd, int sh,
                                int sw, void* stream) {
    if (!a_cl || !g_cl || !dw || B < 1 || D < 1 || H < 1 || W < 1 || CA < 4 || CB < 4 || (CA % 4) || (CB % 4)) {
        set_error("mvs_conv3d_wgrad: bad arguments (channel counts must be multiples of 4)");
        return MVS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int OD = (D - 1) / sd + 1, OH = (H - 1) / sh + 1, OW = (W - 1) / sw + 1;
    if (kd != 1 && kd != 3) { set_error("mvs_conv3d_wgrad: kernel depth %d (1 or 3)", kd); return MVS_ERR_UNSUPPORTED; }
    if (hipMemsetAsync(dw, 0, (size_t)CA * CB * kd * 9 * sizeof(float), st) != hipSuccess) { set_error("mvs_conv3d_wgrad: hipMemsetAsync failed"); return MVS_ERR_LAUNCH; }
    // k = (1,3,3), stride 1: the 2-D layers of the visibility CNN (maps as D = 1 volumes); 1 x 16 x 16 output tiles
    if (kd == 1 && sd == 1 && sh == 1 && sw == 1) return launch_wgrad<WgCfg<1, 1, 1, 1, 16, 1>>(a_cl, g_cl, dw, B, D, H, W, OD, OH, OW, CA, CB, st);
    if (kd == 1) { set_error("mvs_conv3d_wgrad: k = (1,3,3) is built for stride 1"); return MVS_ERR_UNSUPPORTED; }
    if (sd == 1 && sh == 1 && sw == 1) return launch_wgrad<WgCfg<1, 1, 1, 4, 4>>(a_cl, g_cl, dw, B, D, H, W, OD, OH, OW, CA, CB, st);
    if (sd == 1 && sh == 2 && sw == 2) return launch_wgrad<WgCfg<1, 2, 2, 2, 2>>(a_cl, g_cl, dw, B, D, H, W, OD, OH, OW, CA, CB, st);
    if (sd == 2 && sh == 2 && sw == 2) return launch_wgrad<WgCfg<2, 2, 2, 2, 2>>(a_cl, g_cl, dw, B, D, H, W, OD, OH, OW, CA, CB, st);
    set_error("mvs_conv3d_wgrad: stride (%d,%d,%d) unsupported", sd, sh, sw);
    return MVS_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------
// One conv / transposed-conv + BatchNorm(batch statistics) + ReLU [+ skip] block per call, forward and backward: the launches the
// granular entry points above make one by one from Python (pack, convolve, statistics, finalize, normalise / reduce, apply, weight
// gradient, pack, data gradient), chained on the stream in C - a training step of the coarse stages was bound by the ~20 Python-level
// operations per block, not by the GPU.  SyncBatchNorm (an all-reduce between statistics and finalize) and eval-mode BatchNorm keep
// using the granular path.
// ------------------------------------------------------------------------------------------------
static int conv_chunk_of(int cin, int sd, int sh, int sw) { return (sd == 1 && sh == 1 && sw == 1 && cin >= 16) ? 16 : 8; }

static int linear_conv(const float* x, const float* w, int transposed, int tflip, int Cin, int Cout, int kd, int sd, int sh, int sw, int B, int D, int H, int W,
                       void* wpack, const float* zero_bias, float* y, hipStream_t st) {
    int rc;
    if (transposed) {
        if ((rc = mvs_pack_deconv_weights(w, wpack, Cin, Cout, sd, st)) != MVS_OK) return rc;
        return deconv3d_dispatch_bf16x3(x, wpack, zero_bias, nullptr, y, B, Cin, Cout, D, H, W, sd, st, nullptr, nullptr, nullptr, 0);
    }
    if ((rc = mvs_pack_conv_weights(w, wpack, Cout, Cin, kd * 9, conv_chunk_of(Cin, sd, sh, sw), tflip, st)) != MVS_OK) return rc;
    return conv3d_dispatch_bf16x3(x, wpack, zero_bias, y, B, Cin, Cout, D, H, W, kd, sd, sh, sw, 0, st);
}

extern "C" int mvs_train_block_fwd(const float* a_in_cl, const float* w, int transposed, int Cin, int Cout, int kd, int sd, int sh, int sw, int B, int D,
                                   int H, int W, const float* gamma, const float* beta, float eps, float* running_mean, float* running_var,
                                   float momentum, const float* skip_cl, const float* zero_bias, void* wpack_ws, double* sums_ws, float* z_cl,
                                   float* mean, float* var, float* invstd, float* y_cl, int groups, void* stream) {
    if (!a_in_cl || !w || !gamma || !beta || !zero_bias || !wpack_ws || !sums_ws || !z_cl || !mean || !var || !invstd || !y_cl || groups < 1 || B % groups) {
        set_error("mvs_train_block_fwd: bad arguments");
        return MVS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = linear_conv(a_in_cl, w, transposed, 0, Cin, Cout, kd, sd, sh, sw, B, D, H, W, wpack_ws, zero_bias, z_cl, st);
    if (rc != MVS_OK) return rc;
    const long long OD = transposed ? (long long)D * sd : (D + 2 * (kd / 2) - kd) / sd + 1, OH = transposed ? 2LL * H : (H - 1) / sh + 1,
                    OW = transposed ? 2LL * W : (W - 1) / sw + 1;
    const long long n = (long long)(B / groups) * OD * OH * OW;
    if ((rc = mvs_bn_stats(z_cl, sums_ws, n, Cout, groups, stream)) != MVS_OK) return rc;
    if ((rc = mvs_bn_finalize(sums_ws, (double)n, eps, mean, var, invstd, running_mean, running_var, momentum, Cout, groups, stream)) != MVS_OK) return rc;
    return mvs_bn_relu_apply(z_cl, mean, invstd, gamma, beta, skip_cl, y_cl, n, Cout, 1, groups, stream);
}

extern "C" int mvs_train_block_bwd(const float* dy_cl, const float* a_in_cl, const float* z_cl, const float* mean, const float* var, const float* invstd,
                                   const float* w, int transposed, int Cin, int Cout, int kd, int sd, int sh, int sw, int B, int D, int H, int W,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                   const float* zero_bias, void* wpack_ws, double* sums_ws, float* dz_cl, float* dw, float* dgamma, float* dbeta,
                                   float* da_cl, int groups, void* stream) {
    if (!dy_cl || !a_in_cl || !z_cl || !mean || !var || !invstd || !w || !gamma || !beta || !zero_bias || !wpack_ws || !sums_ws || !dz_cl || !dw || !dgamma ||
        !dbeta || groups < 1 || B % groups) {
        set_error("mvs_train_block_bwd: bad arguments");
        return MVS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int OD = transposed ? D * sd : (D + 2 * (kd / 2) - kd) / sd + 1, OH = transposed ? 2 * H : (H - 1) / sh + 1, OW = transposed ? 2 * W : (W - 1) / sw + 1;
    const long long n = (long long)(B / groups) * OD * OH * OW;
    int rc;
    // the reference's checkpoint recomputation = a second train-mode forward: one more momentum step of the running statistics
    if (running_mean != nullptr && (rc = mvs_bn_running_update(mean, var, (double)n, momentum, running_mean, running_var, Cout, groups, stream)) != MVS_OK) return rc;
    if ((rc = mvs_bn_relu_bwd(dy_cl, z_cl, mean, invstd, gamma, beta, sums_ws, 1.0, nullptr, n, Cout, 1, 1, 0, groups, stream)) != MVS_OK) return rc;
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(1), dim3(64), 0, st, sums_ws, groups, Cout, dgamma, dbeta);
    if ((rc = check_launch("bn_param_grads_kernel")) != MVS_OK) return rc;
    if ((rc = mvs_bn_relu_bwd(dy_cl, z_cl, mean, invstd, gamma, beta, sums_ws, (double)n, dz_cl, n, Cout, 1, 1, 1, groups, stream)) != MVS_OK) return rc;
    if (transposed) {
        // weight gradient with the roles of input and output exchanged; data gradient = the strided convolution with the same taps
        if ((rc = mvs_conv3d_wgrad(dz_cl, a_in_cl, dw, B, Cout, Cin, OD, OH, OW, 3, sd, 2, 2, stream)) != MVS_OK) return rc;
        if (da_cl == nullptr) return MVS_OK;
        return linear_conv(dz_cl, w, 0, 0, Cout, Cin, 3, sd, 2, 2, B, OD, OH, OW, wpack_ws, zero_bias, da_cl, st);
    }
    if ((rc = mvs_conv3d_wgrad(a_in_cl, dz_cl, dw, B, Cin, Cout, D, H, W, kd, sd, sh, sw, stream)) != MVS_OK) return rc;
    if (da_cl == nullptr) return MVS_OK;
    if (sd == 1 && sh == 1 && sw == 1)                                     // the convolution with flipped, transposed taps
        return linear_conv(dz_cl, w, 0, 1, Cout, Cin, kd, 1, 1, 1, B, OD, OH, OW, wpack_ws, zero_bias, da_cl, st);
    if ((D % sd) || (H % sh) || (W % sw) || sh != 2 || sw != 2 || kd != 3) { set_error("mvs_train_block_bwd: strided convolutions need even input sizes and stride (1|2,2,2)"); return MVS_ERR_ARG; }
    return linear_conv(dz_cl, w, 1, 0, Cout, Cin, 3, sd, 2, 2, B, OD, OH, OW, wpack_ws, zero_bias, da_cl, st);   // the transposed convolution with the same taps
}
