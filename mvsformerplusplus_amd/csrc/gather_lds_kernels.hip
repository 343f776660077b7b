// LDS-staged gather, translation unit 1 of 5 (gather_lds.h): shape support, the entropy pass with fp32 windows, the hand-off packer.
#include "gather_lds.h"

namespace mvs {

// MVS_GATHER_WINDOW=f32: fp32 windows in every format (A/B measurements of the fp16 window)
bool gl_window_f16_enabled() {
    static const int on = [] { const char* e = getenv("MVS_GATHER_WINDOW"); return (e && e[0] == 'f' && e[1] == '3') ? 0 : 1; }();
    return on != 0;
}

bool gl_supported(int C, int G, int D, int H, int W) {
    if (G != 8 || !(C == 8 || C == 16 || C == 32 || C == 64)) return false;
    if (W % GL_XALIGN != 0 || W < GL_XALIGN || H < 2 || W > 65535 || H > 65535) return false;
    if ((long long)D * H * W > 0x7fffffffLL) return false;                    // 32-bit voxel offsets inside one batch item
    if ((long long)H * W >= (1LL << 28)) return false;                        // 32-bit BYTE offsets of 16-byte positions (the direct form's buffer loads)
    return (size_t)D * (256 / gl_slots(D)) * sizeof(float) <= 64 * 1024;
}

// the fp16-correlation keeping pass exists for every LDS-staged shape; the fp32-correlation (exact) one for D > GL_DCH only
bool gl_keep_supported(int C, int G, int D, int H, int W) { return gl_supported(C, G, D, H, W); }
bool gl_keep32_supported(int C, int G, int D, int H, int W) { return gl_supported(C, G, D, H, W) && D > GL_DCH; }


int gl_launch_entropy_w16(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H, int W,
                          int vb, int ve, hipStream_t st);      // gather_lds_entropy_w16_kernels.hip

int gl_launch_entropy(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H, int W,
                      int vb, int ve, int w16, hipStream_t st) {
    if (w16 && gl_window_f16_enabled()) return gl_launch_entropy_w16(feat, dtype, layout, hom, hyp, ent, B, V, C, D, H, W, vb, ve, st);
    GL_DISPATCH(gl_launch_entropy_t, feat, hom, hyp, ent, B, V, D, H, W, vb, ve, st);
}

// ------------------------------------------------------------------------------------------------
// hand-off packer: planar [N, C, HW] of any feature dtype -> octet-tiled [N, C/8, HW, 8] of the
// requested dtype.  A feature producer that emits the tiled layout directly skips this pass.
// grid = (ceil(HW / 256), C / 8, N)
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pack_features_kernel(const TI* __restrict__ in, TO* __restrict__ out, int C, unsigned HW) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= HW) return;
    const size_t n = blockIdx.z, o = blockIdx.y;
    const TI* src = in + (n * C + o * 8) * HW + p;
    TO* dst = out + ((n * (C / 8) + o) * HW + p) * 8;
    typedef TO to8 __attribute__((ext_vector_type(8)));
    to8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float f = to_f32(src[(size_t)c * HW]);
        // narrowing to fp16: clamped to the fp16 range, like the fp16 windows' staging (gather_lds.h) and the emitter's epilogue - the same values
        if (std::is_same<TO, _Float16>::value && !std::is_same<TI, _Float16>::value) f = fminf(fmaxf(f, -65504.0f), 65504.0f);
        if (sizeof(TI) == 2 && sizeof(TO) == 2) v[c] = *reinterpret_cast<const TO*>(&src[(size_t)c * HW]);      // same 2-byte type: bit copy
        else v[c] = from_f32<TO>(f);
    }
    *reinterpret_cast<to8*>(dst) = v;
}

template <typename TI, typename TO>
static int pack_features_t(const void* in, void* out, int N, int C, int H, int W, hipStream_t st) {
    const unsigned HW = (unsigned)H * (unsigned)W;
    hipLaunchKernelGGL((pack_features_kernel<TI, TO>), dim3(ceil_div(HW, 256), C / 8, N), dim3(256), 0, st, static_cast<const TI*>(in),
                       static_cast<TO*>(out), C, HW);
    return check_launch("pack_features_kernel");
}

int pack_features_dispatch(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W, hipStream_t st) {
    if (out_dtype == MVS_DTYPE_F32) {
        switch (in_dtype) {
            case MVS_DTYPE_F32: return pack_features_t<float, float>(in, out, N, C, H, W, st);
            case MVS_DTYPE_BF16: return pack_features_t<uint16_t, float>(in, out, N, C, H, W, st);
            default: return pack_features_t<_Float16, float>(in, out, N, C, H, W, st);
        }
    }
    if (out_dtype == MVS_DTYPE_BF16) {
        switch (in_dtype) {
            case MVS_DTYPE_F32: return pack_features_t<float, uint16_t>(in, out, N, C, H, W, st);
            case MVS_DTYPE_BF16: return pack_features_t<uint16_t, uint16_t>(in, out, N, C, H, W, st);
            default: set_error("mvs_pack_features: fp16 -> bf16 is not a hand-off the reference produces"); return MVS_ERR_UNSUPPORTED;
        }
    }
    switch (in_dtype) {
        case MVS_DTYPE_F32: return pack_features_t<float, _Float16>(in, out, N, C, H, W, st);
        case MVS_DTYPE_F16: return pack_features_t<_Float16, _Float16>(in, out, N, C, H, W, st);
        default: set_error("mvs_pack_features: bf16 -> fp16 is not a hand-off the reference produces"); return MVS_ERR_UNSUPPORTED;
    }
}

}  // namespace mvs

namespace mvs { MVS_DEFINE_SAT_READER(sat_read_gather) }
