// TEST INFRASTRUCTURE ONLY - never part of the product build.
//
// A tiny host-side stand-in for <hip/hip_runtime.h> that lets the *unchanged* kernel sources under
// mvsformerplusplus_amd/csrc/ be compiled for x86 (clang++ -x c++ -I tests/hipemu) and executed on
// the CPU of the GPU-less build container.  Each workgroup runs as a set of cooperative fibers
// (one per work-item) scheduled round-robin in lane order on one OS thread:
//   * __syncthreads()             -> workgroup barrier over the fibers
//   * __shfl* / MFMA builtins     -> wave-level (64 lanes) exchange through a per-wave buffer, using
//                                    the gfx950 operand/result lane maps documented in
//                                    cdna_hip_programming.md section 3
// It exists to check index arithmetic, tiling, fragment layouts and barrier placement of the
// real kernels against the oracle before spending GPU minutes.  The package never loads the
// resulting library; only tests/ do (tests/hipemu_build.py).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <utility>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hipemu {
struct Ids { dim3 tid, bid, bdim, gdim; };
extern Ids cur;                      // ids of the fiber that is running right now
void syncthreads();
int syncthreads_and(int pred);
int lane_id();
// every live lane of the calling wave deposits `size` bytes; returns once all have, with all[64*size]
// holding every lane's deposit (dead lanes: zeros)
void wave_gather(const void* mine, void* all, size_t size);
void* dyn_smem();
void wave_sync();            // all live lanes of the calling wave rendezvous (the emulated lanes are not lock-step)
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (hipemu::cur.tid)
#define blockIdx (hipemu::cur.bid)
#define blockDim (hipemu::cur.bdim)
#define gridDim (hipemu::cur.gdim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());
#define __syncthreads() hipemu::syncthreads()
#define __syncthreads_and(p) hipemu::syncthreads_and((p) ? 1 : 0)

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu: no error"; }
template <class F>
static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
// occupancy / device queries: a tiny "chip" (3 CUs x 1 block) so that persistent kernels really loop over several tiles per block
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 3; return hipSuccess; }
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }
#define HIP_SYMBOL(x) x
template <class T>
static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <class T>
static inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }

template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    hipemu::run_grid(grid, block, shmem, body);
}

// ---- wave-level data movement -------------------------------------------------------------
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    T all[64];
    hipemu::wave_gather(&v, all, sizeof(T));
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    return all[base + (src & (width - 1))];
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    T all[64];
    hipemu::wave_gather(&v, all, sizeof(T));
    int l = hipemu::lane_id();
    int t = l ^ mask;
    if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
    return all[t];
}
template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    T all[64];
    hipemu::wave_gather(&v, all, sizeof(T));
    int l = hipemu::lane_id();
    int t = l + (int)delta;
    if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
    return all[t];
}

// ---- MFMA (gfx950 lane maps; f32 result is a k-ordered fmaf chain, bitwise like the hardware) ----
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D: col = l&15, row = 4*(l>>4) + r.
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    struct AB { float a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id(), col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(all[row + 16 * k].a, all[col + 16 * k].b, acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col = l&31,
// row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    struct AB { float a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id(), col = l & 31, h = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(all[row + 32 * k].a, all[col + 32 * k].b, acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32

// v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15], j<8; D as 16x16 above.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
    struct AB { hipemu_bf16x8 a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id(), col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 8; ++j) acc += (float)all[row + 16 * k].a[j] * (float)all[col + 16 * k].b[j];   // bf16 products are exact in fp32
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_f32_16x16x32_bf16

// v_mfma_f32_16x16x32_f16: the same lane maps with fp16 operands (products of two fp16 values are exact in fp32)
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
    struct AB { hipemu_f16x8 a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id(), col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 8; ++j) acc += (float)all[row + 16 * k].a[j] * (float)all[col + 16 * k].b[j];
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_f32_16x16x32_f16
// v_mfma_f32_16x16x16_f16 (the K = 16 form): lane l supplies A[i=l&15][k=4*(l>>4)+j], B[k=4*(l>>4)+j][col=l&15], j<4; D as 16x16 above.
typedef _Float16 hipemu_f16x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x16f16(hipemu_f16x4 a, hipemu_f16x4 b, hipemu_f32x4 c, int, int, int) {
    struct AB { hipemu_f16x4 a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id(), col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) acc += (float)all[row + 16 * k].a[j] * (float)all[col + 16 * k].b[j];
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x16f16 hipemu_mfma_f32_16x16x16f16
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {a (bytes 4..7), b (bytes 0..3)} (selectors 0..7 only)
static inline unsigned hipemu_perm(unsigned a, unsigned b, unsigned sel) {
    const unsigned long long src = ((unsigned long long)a << 32) | b;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((src >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
#define __builtin_amdgcn_perm hipemu_perm
static inline float hipemu_fmed3f(float a, float b, float c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }
#define __builtin_amdgcn_fmed3f hipemu_fmed3f

static inline int hipemu_readfirstlane(int v) {
    int all[64];
    hipemu::wave_gather(&v, all, sizeof(int));
    return all[0];
}
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane
typedef unsigned hipemu_u32x2_t __attribute__((ext_vector_type(2)));
static inline int hipemu_readlane(int v, int src) {
    int all[64];
    hipemu::wave_gather(&v, all, sizeof(int));
    return all[src & 63];
}
#define __builtin_amdgcn_readlane hipemu_readlane
#define __builtin_amdgcn_fence(order, scope) ((void)0)      // the emulated lanes rendezvous in wave_barrier / __syncthreads

// v_permlane16_swap: lanes 16-31 / 48-63 of the first operand <-> lanes 0-15 / 32-47 of the second; returns {new first, new second}
static inline hipemu_u32x2_t hipemu_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    struct AB { unsigned a, b; } mine{a, b}, all[64];
    hipemu::wave_gather(&mine, all, sizeof(AB));
    const int l = hipemu::lane_id();
    hipemu_u32x2_t r;
    r[0] = (l & 16) ? all[l - 16].b : a;
    r[1] = (l & 16) ? b : all[l + 16].a;
    return r;
}
#define __builtin_amdgcn_permlane16_swap hipemu_permlane16_swap

// buffer descriptors (raw buffer loads): base + range; a dword whose offset lies beyond the range reads as zero
struct __amdgpu_buffer_rsrc_t { const char* base; unsigned bytes; };
static inline __amdgpu_buffer_rsrc_t hipemu_make_buffer_rsrc(void* p, short, int num, int) {
    return __amdgpu_buffer_rsrc_t{static_cast<const char*>(p), (unsigned)num};
}
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_buffer_rsrc
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
template <int N>
static inline void hipemu_buffer_read(__amdgpu_buffer_rsrc_t r, int voff, int soff, void* dst) {
    const unsigned off = (unsigned)voff + (unsigned)soff;
    memset(dst, 0, N);
    for (int b = 0; b < N; b += (N < 4 ? N : 4))
        if ((unsigned long long)off + b + (N < 4 ? N : 4) <= r.bytes) memcpy(static_cast<char*>(dst) + b, r.base + off + b, N < 4 ? N : 4);
}
static inline hipemu_u32x4 hipemu_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { hipemu_u32x4 v; hipemu_buffer_read<16>(r, voff, soff, &v); return v; }
static inline hipemu_u32x2 hipemu_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { hipemu_u32x2 v; hipemu_buffer_read<8>(r, voff, soff, &v); return v; }
static inline unsigned hipemu_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { unsigned v; hipemu_buffer_read<4>(r, voff, soff, &v); return v; }
static inline unsigned short hipemu_raw_buffer_load_b16(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { unsigned short v; hipemu_buffer_read<2>(r, voff, soff, &v); return v; }
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu_raw_buffer_load_b64
#define __builtin_amdgcn_raw_buffer_load_b32 hipemu_raw_buffer_load_b32
#define __builtin_amdgcn_raw_buffer_load_b16 hipemu_raw_buffer_load_b16

// DPP (gfx9 controls used by the kernels): quad_perm 0x00-0xFF, row_shr:n 0x111-0x11F, wave_shl:1 0x130, wave_shr:1 0x138, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143.  A lane whose row / bank is masked off, or whose source
// lane does not exist, keeps `old` (bound_ctrl:0 would give 0 for a missing source).
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    int all[64];
    hipemu::wave_gather(&src, all, sizeof(int));
    const int l = hipemu::lane_id();
    const int row = l >> 4, in_row = l & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; from = in_row >= n ? l - n : -1; }
    else if (ctrl == 0x130) from = l < 63 ? l + 1 : -1;
    else if (ctrl == 0x138) from = l > 0 ? l - 1 : -1;
    else if (ctrl == 0x140) from = (l & ~15) | (15 - in_row);
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x142) from = row > 0 ? 16 * row - 1 : -1;
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
    else { fprintf(stderr, "hipemu: unsupported DPP control 0x%x\n", ctrl); abort(); }
    if (from < 0) return bound_ctrl ? 0 : old;
    return all[from];
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_logf(x) log2f(x)
static inline int hipemu_any(int pred) {
    int all[64];
    hipemu::wave_gather(&pred, all, sizeof(int));
    for (int i = 0; i < 64; ++i) if (all[i]) return 1;
    return 0;
}
#define __any(p) hipemu_any((p) ? 1 : 0)
static inline float hipemu_half_of(unsigned h2, int hi) {
    const unsigned short b = (unsigned short)(hi ? (h2 >> 16) : (h2 & 0xffffu));
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}
#define MVS_FMA_MIX_LO(h2, w, acc) fmaf(hipemu_half_of((h2), 0), (w), (acc))
#define MVS_FMA_MIX_HI(h2, w, acc) fmaf(hipemu_half_of((h2), 1), (w), (acc))
#define MVS_OPAQUE_REG "r"      // x86 register class for the kernels' opaque-value asm
#define MVS_OPAQUE_SREG "r"
#define MVS_NO_OPAQUE_VEC 1    // 128-bit bf16 vectors have no x86 asm register class; the laundering is a GPU register-allocation hint only

// LDS-DMA (global_load_lds_dwordx4): the emulated copy lands at once; every fiber is one lane
#define MVS_GLOBAL_LOAD_LDS16(gsrc, ldst) memcpy(static_cast<char*>(static_cast<void*>(ldst)) + 16 * hipemu::lane_id(), (gsrc), 16)
#define MVS_WAIT_VMEM() ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_getreg(imm) 0
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()

// ---- device math spellings used by the kernels ----
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; *p = v < o ? v : o; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; *p = v > o ? v : o; return o; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
