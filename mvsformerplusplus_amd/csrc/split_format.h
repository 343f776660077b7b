// The split activation format (MVS_PREC_BF16X3_SPLIT) and the hi / lo split helpers of the split-bf16 convolution kernels
// (conv_bf16x3_kernels.hip).
#pragma once
#include "mvs_common.h"

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];                 // round to nearest even (v_cvt_pk_bf16_f32)
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// ---- the split activation format (MVS_PREC_BF16X3_SPLIT) ---------------------------------------------------------------
// Between the layers of the inference U-Net the activations live in HBM ALREADY SPLIT: channel-last, per voxel C / 8 octets of
// [hi x8 | lo x8] bf16 = the same 4 bytes per element as fp32.  The producing epilogue splits each value once; the consumers'
// staging is a plain copy of 32-byte runs into the LDS image (round 2 split every staged element - halo voxels included, 2.5x
// the tile for the 4x4x16 stride-1 tile - on the consumer side: ~45 VALU per 8 channels in the memory phase of every tile).
// An accumulator lane holds 4 consecutive channels of a voxel = one QUAD of an octet; its partner 16 lanes up holds the other
// quad of the same octet and voxel.  v_permlane16_swap exchanges the halves so that the even lane row owns hi x8 and the odd row
// lo x8 of the octet: one 16-byte store per lane, 32 contiguous bytes per lane pair.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16x2(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float bf16_lo_f32(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi_f32(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// store the channel quad `v` of this lane (quad g & 1 of the octet at `octet`) - EVERY lane of the wave must call (lane exchange);
// `guard` = the voxel exists
__device__ __forceinline__ void split_store_quad(float* octet, int g, const float4& v, bool guard) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    __bf16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)x[j];
        l[j] = (__bf16)(x[j] - (float)h[j]);
    }
    unsigned h0 = pack_bf16x2(h[0], h[1]), h1 = pack_bf16x2(h[2], h[3]), l0 = pack_bf16x2(l[0], l[1]), l1 = pack_bf16x2(l[2], l[3]);
    // rows 1, 3 of the first operand <-> rows 0, 2 of the second: even rows end up with {own hi, partner's hi}, odd rows with
    // {partner's lo, own lo}
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
    if (guard) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(octet) + (g & 1) * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// the fp32 values of channel quad q of the octet at `octet` (skip connections): raw = {hi, hi, lo, lo} dwords
__device__ __forceinline__ float4 split_raw_quad(const float* octet, int q) {
    const u32x2 h = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(octet) + q * 8);
    const u32x2 l = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(octet) + 16 + q * 8);
    // (scalars first: __builtin_bit_cast applied directly to an ext-vector ELEMENT expression yields element 0 with this clang)
    const unsigned h0 = h[0], h1 = h[1], l0 = l[0], l1 = l[1];
    return make_float4(__builtin_bit_cast(float, h0), __builtin_bit_cast(float, h1), __builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1));
}
__device__ __forceinline__ float4 split_join_quad(const float4& raw) {
    const unsigned h0 = __builtin_bit_cast(unsigned, raw.x), h1 = __builtin_bit_cast(unsigned, raw.y);
    const unsigned l0 = __builtin_bit_cast(unsigned, raw.z), l1 = __builtin_bit_cast(unsigned, raw.w);
    return make_float4(bf16_lo_f32(h0) + bf16_lo_f32(l0), bf16_hi_f32(h0) + bf16_hi_f32(l0), bf16_lo_f32(h1) + bf16_lo_f32(l1), bf16_hi_f32(h1) + bf16_hi_f32(l1));
}
// staged 32-byte run (one voxel x one octet) -> the LDS image [hi x8 | lo x8]
template <bool SPLIT>
__device__ __forceinline__ void stage_to_lds(char* dst, const float4& u, const float4& v) {
    if (SPLIT) {
        *reinterpret_cast<float4*>(dst) = u;
        *reinterpret_cast<float4*>(dst + 16) = v;
    } else {
        bf16x8 hi, lo;
        split8(u, v, hi, lo);
        *reinterpret_cast<bf16x8*>(dst) = hi;
        *reinterpret_cast<bf16x8*>(dst + 16) = lo;
    }
}

}  // namespace mvs
