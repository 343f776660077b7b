// Microbenchmark (GPU box only): cost of one wave64 vector-memory instruction on gfx950 as a function of its width,
// alignment, lane overlap and exec mask - the hardware model the gather kernels are designed against.
// Throughput-bound on purpose: 32 independent loads in flight per wave, 64 rounds over the same per-block footprint
// (run-time zero stride keeps the compiler from hoisting), so every round after the first hits in L1.
//   hipcc --offload-arch=gfx950 -O3 scripts/l1_ubench.hip -o scripts/l1_ubench.bin && scripts/l1_ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int HW = 1152 * 1536, W = 1536, NL = 32, ROUNDS = 64;   // loads per thread and round
struct __attribute__((packed, aligned(4))) F2 { float x, y; };
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

enum { DWORD_SPARSE8, DWORD_SPARSE_JIT, DWORD, DWORD_HALFMASK, DWORD_1LANE, X2_ALIGNED, X2_OVERLAP_EVEN, X2_OVERLAP_ODD, X2_STRIDE2_UNALIGNED, X4_ALIGNED, X4_OVERLAP, X2_ROWPAIR, LDS_B32, LDS_B64_OVERLAP };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ f, float* __restrict__ out, int shift) {
    __shared__ float lds[4096 + 8];
    const int lane = threadIdx.x & 63;
    const unsigned p = (blockIdx.x * 256 + threadIdx.x) % (unsigned)(HW - 4 * W);
    const unsigned wave0 = p - lane;                            // first pixel of the wave
    float acc[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[i] = 0.f;
    if (MODE == LDS_B32 || MODE == LDS_B64_OVERLAP) {
        for (int i = threadIdx.x; i < 4096 + 8; i += 256) lds[i] = f[p + i];
        __syncthreads();
    }
#pragma unroll 1
    for (int it = 0; it < ROUNDS; ++it)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const float* s = f + (size_t)(i & 7) * HW + (i >> 3) * 4 + it * shift;     // 8 planes x 4 small shifts (16-byte aligned)
        if (MODE == DWORD) acc[i] += s[p];
        if (MODE == DWORD_SPARSE8) acc[i] += s[(lane & 7) == 0 ? p : wave0];                      // 8 lanes read their own element, 56 share one
        if (MODE == DWORD_SPARSE_JIT) acc[i] += s[p + ((lane * 5 + i) & 3)];                       // lane-consecutive with 0..3 elements of jitter
        if (MODE == DWORD_HALFMASK) { if (lane & 1) acc[i] += s[p]; }
        if (MODE == DWORD_1LANE) { if (lane == 0) acc[i] += s[p]; }
        if (MODE == X2_ALIGNED) { const float2 v = *reinterpret_cast<const float2*>(s + wave0 + lane * 2); acc[i] += v.x + v.y; }
        if (MODE == X2_OVERLAP_EVEN) { const F2 v = *reinterpret_cast<const F2*>(s + p); acc[i] += v.x + v.y; }
        if (MODE == X2_OVERLAP_ODD) { const F2 v = *reinterpret_cast<const F2*>(s + p + 1); acc[i] += v.x + v.y; }
        if (MODE == X2_STRIDE2_UNALIGNED) { const F2 v = *reinterpret_cast<const F2*>(s + wave0 + lane * 2 + 1); acc[i] += v.x + v.y; }
        if (MODE == X4_ALIGNED) { const float4 v = *reinterpret_cast<const float4*>(s + wave0 + lane * 4); acc[i] += v.x + v.y + v.z + v.w; }
        if (MODE == X4_OVERLAP) { const F4u v = *reinterpret_cast<const F4u*>(s + p); acc[i] += v.x + v.y + v.z + v.w; }
        if (MODE == X2_ROWPAIR) { const F2 v = *reinterpret_cast<const F2*>(s + p + (i & 1) * W); acc[i] += v.x + v.y; }
        if (MODE == LDS_B32) acc[i] += lds[threadIdx.x + i * 97 % 3800 + it * shift];
        if (MODE == LDS_B64_OVERLAP) { const F2 v = *reinterpret_cast<const F2*>(lds + threadIdx.x + i * 97 % 3800 + it * shift); acc[i] += v.x + v.y; }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
float run(int blocks, const float* f, float* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, f, out, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, f, out, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    float *f, *out;
    const int blocks = HW / 256;                               // one thread per pixel, like the warp kernels
    CHECK(hipMalloc(&f, (size_t)(HW + 16 * W) * 8 * 4));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemset(f, 0, (size_t)(HW + 16 * W) * 8 * 4));
    const int reps = 20;
    struct { const char* name; float ms; int bytes; } r[] = {
        {"dword, lane-consecutive            ", run<DWORD>(blocks, f, out, reps), 4},
        {"dword, 8 own + 56 on one address   ", run<DWORD_SPARSE8>(blocks, f, out, reps), 4},
        {"dword, consecutive + 0..3 jitter   ", run<DWORD_SPARSE_JIT>(blocks, f, out, reps), 4},
        {"dword, odd lanes only (exec mask)  ", run<DWORD_HALFMASK>(blocks, f, out, reps), 4},
        {"dword, lane 0 only (exec mask)     ", run<DWORD_1LANE>(blocks, f, out, reps), 4},
        {"dwordx2, 8B-aligned, contiguous    ", run<X2_ALIGNED>(blocks, f, out, reps), 8},
        {"dwordx2, lanes overlap (x, x+1)    ", run<X2_OVERLAP_EVEN>(blocks, f, out, reps), 8},
        {"dwordx2, lanes overlap, +1 element ", run<X2_OVERLAP_ODD>(blocks, f, out, reps), 8},
        {"dwordx2, contiguous, 4B-misaligned ", run<X2_STRIDE2_UNALIGNED>(blocks, f, out, reps), 8},
        {"dwordx4, 16B-aligned, contiguous   ", run<X4_ALIGNED>(blocks, f, out, reps), 16},
        {"dwordx4, lanes overlap (x..x+3)    ", run<X4_OVERLAP>(blocks, f, out, reps), 16},
        {"dwordx2 overlap, alternating rows  ", run<X2_ROWPAIR>(blocks, f, out, reps), 8},
        {"ds_read_b32                        ", run<LDS_B32>(blocks, f, out, reps), 4},
        {"ds_read_b64, lanes overlap         ", run<LDS_B64_OVERLAP>(blocks, f, out, reps), 8},
    };
    const double instr = (double)blocks * 4 * NL * ROUNDS;              // wave instructions per launch
    printf("%-36s %8s %16s %16s\n", "wave64 load (32 in flight)", "ms", "clk/instr/CU", "lane-B/clk/CU");
    for (auto& x : r) {
        const double clk = x.ms * 1e-3 * 2.4e9 * 256 / instr;
        printf("%-36s %8.4f %16.2f %16.1f\n", x.name, x.ms, clk, 64.0 * x.bytes / clk);
    }
    return 0;
}
