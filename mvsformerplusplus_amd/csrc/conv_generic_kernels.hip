// Shape-generic exact-fp32 Conv3d / ConvTranspose3d (the layer shapes OUTSIDE the tuned tables of
// conv_cfg.h): any channel counts, any kernel size / stride / padding, transposed form with output padding; folded-BatchNorm bias,
// ReLU and the U-Net skip add in the epilogue.  This is what a regulariser built with base_ch != 8 (reference cost_volume.py:29-49:
// CostRegNet(G, G) - widths 2G / 4G / 8G) or with in_channels != base_channels (the 1x1x1 `inner` convolution, module.py:385-388,
// 481-484) runs on, and what the standalone Conv3d / Deconv3d wrappers (module.py:89-165) use for a layer no shipped config has.
//
// It is a plain FMA kernel, not an MFMA one: no shipped configuration reaches it (both released configs have base_ch = 8), so it is
// built for coverage and exactness - fp32 products, fp32 accumulation in tap-major / channel-minor order - not for the roofline; the
// hot path of the shipped configs stays on conv_kernels.hip / conv_bf16x3_kernels.hip.
//
// Mapping: one work-item = one output voxel x COT consecutive output channels (blockIdx.y = channel chunk, blockIdx.z = batch).
// Activations are channel-last [B, D, H, W, C]: a work-item reads its voxel's Cin-run contiguously (16-byte loads when Cin % 4 == 0),
// neighbouring lanes read neighbouring voxels.  Weights are [tap][Cin][Cout] fp32; their address is wave-uniform (it depends on the
// tap, the input channel and the block's channel chunk only), so they come through the scalar cache, not the vector memory pipe.
#include "conv_cfg.h"

namespace mvs {

struct GenericConvArgs {
    const float* x;        // [B, D, H, W, Cin]
    const float* w;        // [kd*kh*kw][Cin][Cout]
    const float* bias;     // [Cout] or null
    const float* skip;     // [B, OD, OH, OW, Cout] or null: added AFTER bias / ReLU (module.py:403-405)
    float* y;              // [B, OD, OH, OW, Cout]
    int Cin, Cout, D, H, W, OD, OH, OW;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int relu;
};

// input index along one axis for output index o and kernel index k; false when the tap does not exist for this output
template <bool TRANSPOSED>
__device__ __forceinline__ bool tap_index(int o, int k, int s, int p, int n, int* i) {
    if (TRANSPOSED) {
        // ConvTranspose: o = i * s - p + k  <=>  i = (o + p - k) / s, only when the division is exact
        const int t = o + p - k;
        if (t < 0) return false;
        const int q = t / s;
        if (q * s != t || q >= n) return false;
        *i = q;
        return true;
    }
    const int q = o * s - p + k;
    if (q < 0 || q >= n) return false;
    *i = q;
    return true;
}

// one tap's contribution: acc[c] += sum_ci x[ci] * w[ci][co0 + c].  FULL = the block's COT output channels all exist: the weight run
// w[ci][co0 .. co0 + COT) is contiguous and its address wave-uniform, so the compiler fetches it with wide scalar loads (s_load_dwordx8)
// instead of one s_load_dword per weight; the tail chunk of Cout reads a clamped (valid) address per channel and never stores those lanes.
template <int COT, bool VEC4, bool FULL>
__device__ __forceinline__ void generic_tap(const float* __restrict__ xp, const float* __restrict__ wp, int Cin, int Cout, int nvalid, float* acc) {
    if (VEC4) {
        for (int ci = 0; ci < Cin; ci += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + ci);
            const float* w0 = wp + (size_t)ci * Cout;
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                const int cc = FULL ? c : (c < nvalid ? c : 0);
                acc[c] = fmaf(xv.x, w0[cc], acc[c]);
                acc[c] = fmaf(xv.y, w0[Cout + cc], acc[c]);
                acc[c] = fmaf(xv.z, w0[2 * Cout + cc], acc[c]);
                acc[c] = fmaf(xv.w, w0[3 * Cout + cc], acc[c]);
            }
        }
    } else {
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xp[ci];
            const float* w0 = wp + (size_t)ci * Cout;
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                const int cc = FULL ? c : (c < nvalid ? c : 0);
                acc[c] = fmaf(xv, w0[cc], acc[c]);
            }
        }
    }
}

template <int COT, bool TRANSPOSED, bool VEC4, bool FULL>
__device__ __forceinline__ void generic_voxel(const GenericConvArgs& a, const float* __restrict__ xb, int oz, int oy, int ox, int co0, int nvalid, float* acc) {
    for (int kz = 0; kz < a.kd; ++kz) {
        int iz;
        if (!tap_index<TRANSPOSED>(oz, kz, a.sd, a.pd, a.D, &iz)) continue;
        for (int ky = 0; ky < a.kh; ++ky) {
            int iy;
            if (!tap_index<TRANSPOSED>(oy, ky, a.sh, a.ph, a.H, &iy)) continue;
            for (int kx = 0; kx < a.kw; ++kx) {
                int ix;
                if (!tap_index<TRANSPOSED>(ox, kx, a.sw, a.pw, a.W, &ix)) continue;
                const float* xp = xb + (((size_t)iz * a.H + iy) * a.W + ix) * a.Cin;
                const float* wp = a.w + (size_t)((kz * a.kh + ky) * a.kw + kx) * a.Cin * a.Cout + co0;
                generic_tap<COT, VEC4, FULL>(xp, wp, a.Cin, a.Cout, nvalid, acc);
            }
        }
    }
}

template <int COT, bool TRANSPOSED, bool VEC4>
__global__ __launch_bounds__(256) void conv3d_generic_kernel(GenericConvArgs a) {
    const long long nvox = (long long)a.OD * a.OH * a.OW;
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const int co0 = (int)blockIdx.y * COT;
    const int b = (int)blockIdx.z;
    const int ox = (int)(v % a.OW);
    const long long t = v / a.OW;
    const int oy = (int)(t % a.OH), oz = (int)(t / a.OH);
    const float* xb = a.x + (size_t)b * a.D * a.H * a.W * a.Cin;
    const int nvalid = a.Cout - co0 < COT ? a.Cout - co0 : COT;          // block-uniform

    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;
    if (nvalid == COT) generic_voxel<COT, TRANSPOSED, VEC4, true>(a, xb, oz, oy, ox, co0, nvalid, acc);
    else generic_voxel<COT, TRANSPOSED, VEC4, false>(a, xb, oz, oy, ox, co0, nvalid, acc);

    const size_t o = ((size_t)b * nvox + (size_t)v) * a.Cout + co0;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        if (c >= nvalid) break;
        float r = acc[c] + (a.bias ? a.bias[co0 + c] : 0.0f);
        if (a.relu) r = fmaxf(r, 0.0f);
        if (a.skip) r += a.skip[o + c];
        a.y[o + c] = r;
    }
}

template <int COT>
static int launch_generic(const GenericConvArgs& a, int B, int transposed, hipStream_t st) {
    const long long nvox = (long long)a.OD * a.OH * a.OW;
    const dim3 grid(ceil_div(nvox, 256), ceil_div(a.Cout, COT), (unsigned)B);
    const bool vec4 = (a.Cin % 4) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15u) == 0;      // 16-byte loads of the channel run
    if (transposed) {
        if (vec4) hipLaunchKernelGGL((conv3d_generic_kernel<COT, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3d_generic_kernel<COT, true, false>), grid, dim3(256), 0, st, a);
    } else {
        if (vec4) hipLaunchKernelGGL((conv3d_generic_kernel<COT, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3d_generic_kernel<COT, false, false>), grid, dim3(256), 0, st, a);
    }
    return check_launch("conv3d_generic_kernel");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_conv3d_generic_fwd(const float* x_cl, const float* w_tck, const float* bias, const float* skip_cl, float* y_cl, int B,
                                      int Cin, int Cout, int D, int H, int W, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh,
                                      int sw, int pd, int ph, int pw, int transposed, int relu, void* stream) {
    if (!x_cl || !w_tck || !y_cl || B < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_conv3d_generic_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (kd < 1 || kh < 1 || kw < 1 || sd < 1 || sh < 1 || sw < 1 || pd < 0 || ph < 0 || pw < 0) { set_error("mvs_conv3d_generic_fwd: bad kernel / stride / padding"); return MVS_ERR_ARG; }
    if (B > 65535 || (Cout + 7) / 8 > 65535) { set_error("mvs_conv3d_generic_fwd: B or Cout beyond the launch grid"); return MVS_ERR_UNSUPPORTED; }
    const int n[3] = {D, H, W}, o[3] = {OD, OH, OW}, k[3] = {kd, kh, kw}, s[3] = {sd, sh, sw}, p[3] = {pd, ph, pw};
    for (int i = 0; i < 3; ++i) {
        if (transposed) {
            // (n - 1) s - 2 p + k + output_padding, 0 <= output_padding < s (ConvTranspose3d)
            const int base = (n[i] - 1) * s[i] - 2 * p[i] + k[i];
            if (o[i] < base || o[i] >= base + s[i] || o[i] < 1) { set_error("mvs_conv3d_generic_fwd: transposed output size %d not in [%d, %d) on axis %d", o[i], base, base + s[i], i); return MVS_ERR_ARG; }
        } else {
            const int full = n[i] + 2 * p[i] - k[i];
            if (full < 0 || o[i] != full / s[i] + 1) { set_error("mvs_conv3d_generic_fwd: output size %d != (%d + 2*%d - %d) / %d + 1 on axis %d", o[i], n[i], p[i], k[i], s[i], i); return MVS_ERR_ARG; }
        }
    }
    GenericConvArgs a;
    a.x = x_cl; a.w = w_tck; a.bias = bias; a.skip = skip_cl; a.y = y_cl;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.OD = OD; a.OH = OH; a.OW = OW;
    a.kd = kd; a.kh = kh; a.kw = kw; a.sd = sd; a.sh = sh; a.sw = sw; a.pd = pd; a.ph = ph; a.pw = pw;
    a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (Cout == 1) return launch_generic<1>(a, B, transposed, st);
    if (Cout <= 4) return launch_generic<4>(a, B, transposed, st);
    return launch_generic<8>(a, B, transposed, st);
}

// One source of truth for "is there a tuned MFMA kernel for this layer shape": the X-macro tables of conv_cfg.h that the dispatchers
// of conv_kernels.hip / conv_bf16x3_kernels.hip expand.  The host mirror asks before it packs weights (module.py: Conv3d / Deconv3d /
// the U-Nets fall to mvs_conv3d_generic_fwd otherwise).
extern "C" int mvs_conv3d_is_tuned(int Cin, int Cout, int kd, int sd, int sh, int sw) {
#define MVS_X(CI, CO, KD, SD, SH, SW, TD, TH, CH) \
    if (Cin == CI && Cout == CO && kd == KD && sd == SD && sh == SH && sw == SW) return 1;
    MVS_CONV_TABLE(MVS_X)
#undef MVS_X
    return 0;
}

extern "C" int mvs_deconv3d_is_tuned(int Cin, int Cout, int sd) {
#define MVS_X(CI, CO, SD, TDM, THM) \
    if (Cin == CI && Cout == CO && sd == SD) return 1;
    MVS_DECONV_TABLE(MVS_X)
#undef MVS_X
    return 0;
}
