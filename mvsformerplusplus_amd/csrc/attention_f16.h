// Interface between transformer_kernels.hip (qkv operand writer, C ABI) and attention_f16_kernels.hip (the 16-bit flash attention, MVS_PREC_ATTN16).
#pragma once
#include <hip/hip_runtime.h>

namespace mvs {

constexpr int kAttnPad = 256;        // token padding of the fp16 operand buffers: a multiple of every key block and query block

int launch_attention16(const void* q, const void* kp, const void* vp, float* out, int B, int n, int heads, int variant, hipStream_t st);

}  // namespace mvs
