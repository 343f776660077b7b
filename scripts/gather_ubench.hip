// Microbenchmark (GPU box only): what limits per-lane gathers on gfx950 - lanes/clk in the texture addresser or bytes/clk
// out of L1/L2?  Every variant reads a 1152x1536x8-channel fp32 feature map (56.6 MB, like stage 4) with the access
// pattern of the warp kernels (64 consecutive pixels per wave, +small disparity shift per depth plane) and sums the taps.
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_ubench.hip -o gpurun_out/gather_ubench && gpurun_out/gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int H = 1152, W = 1536, C = 8, D = 4, HW = H * W;
struct __attribute__((packed, aligned(4))) F2 { float x, y; };
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

// planar [C][HW]; per (pixel, plane, channel): 4 scalar taps
__global__ __launch_bounds__(256) void k_planar_dword(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* s = f + (size_t)c * HW;
            acc += s[o] + s[o + 1] + s[o + W] + s[o + W + 1];
        }
    }
    out[p] = acc;
}
// planar; 2 unaligned 8-byte pair loads per (pixel, plane, channel)  [current kernels]
__global__ __launch_bounds__(256) void k_planar_pair(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* s = f + (size_t)c * HW;
            const F2 a = *reinterpret_cast<const F2*>(s + o), b = *reinterpret_cast<const F2*>(s + o + W);
            acc += a.x + a.y + b.x + b.y;
        }
    }
    out[p] = acc;
}
// planar; each lane serves TWO adjacent pixels with one unaligned 16-byte load per row (x0..x0+3)
__global__ __launch_bounds__(256) void k_planar_quad2px(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int p = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* s = f + (size_t)c * HW;
            const F4u a = *reinterpret_cast<const F4u*>(s + o), b = *reinterpret_cast<const F4u*>(s + o + W);
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        }
    }
    out[p] = acc;
}
// planar; only the RIGHT tap of each row is loaded (2 dword loads per channel and plane); the left tap comes from the
// neighbouring lane through a DPP wave shift (valid wherever consecutive pixels project to consecutive columns)
__global__ __launch_bounds__(256) void k_planar_dpp(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* s = f + (size_t)c * HW;
            const float tr = s[o + 1], br = s[o + W + 1];
            const float tl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tr), 0x138, 0xf, 0xf, false));
            const float bl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, br), 0x138, 0xf, 0xf, false));
            acc += tl + tr + bl + br;
        }
    }
    out[p] = acc;
}
// channel-last [HW][C]; lane = pixel; both x-taps of a row are 2*C contiguous floats -> 4 x 16-byte loads per row
__global__ __launch_bounds__(256) void k_cl_lane_pixel(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
        const float4* r0 = reinterpret_cast<const float4*>(f + (size_t)o * C);
        const float4* r1 = reinterpret_cast<const float4*>(f + (size_t)(o + W) * C);
#pragma unroll
        for (int q = 0; q < 2 * C / 4; ++q) {
            const float4 a = r0[q], b = r1[q];
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        }
    }
    out[p] = acc;
}
// channel-last; lane = (pixel, 16-byte piece of the 64-byte two-tap run): 4 lanes per pixel, 16 pixels per wave
__global__ __launch_bounds__(256) void k_cl_lane_piece(const float* __restrict__ f, float* __restrict__ out, int shift) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t >> 2, q = t & 3;
    if (p >= HW - 4 * W) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const unsigned o = p + d + shift;
        const float4 a = reinterpret_cast<const float4*>(f + (size_t)o * C)[q];
        const float4 b = reinterpret_cast<const float4*>(f + (size_t)(o + W) * C)[q];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) out[p] = acc;
}
// LDS-staged: a block stages the rows it needs (channel-planar tile) with coalesced loads, then gathers with ds_read
__global__ __launch_bounds__(256) void k_lds_staged(const float* __restrict__ f, float* __restrict__ out, int shift) {
    __shared__ float tile[C][2][256 + 16];
    const int p0 = blockIdx.x * 256, tid = threadIdx.x;
    if (p0 >= HW - 4 * W - 512) return;
    for (int c = 0; c < C; ++c)
        for (int r = 0; r < 2; ++r)
            for (int i = tid; i < 256 + 16; i += 256) tile[c][r][i] = f[(size_t)c * HW + p0 + shift + r * W + i];
    __syncthreads();
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const int o = tid + d;
#pragma unroll
        for (int c = 0; c < C; ++c) acc += tile[c][0][o] + tile[c][0][o + 1] + tile[c][1][o] + tile[c][1][o + 1];
    }
    out[p0 + tid] = acc;
}

template <class K>
float run(K kern, int blocks, const float* f, float* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, f, out, 3);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, f, out, 3 + i);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    float *f, *out;
    CHECK(hipMalloc(&f, (size_t)(HW + 8 * W) * C * 4));
    CHECK(hipMalloc(&out, (size_t)HW * 4));
    CHECK(hipMemset(f, 0, (size_t)(HW + 8 * W) * C * 4));
    const int reps = 20;
    const double taps = (double)HW * D * C * 4;      // tap values consumed per launch
    struct { const char* name; float ms; } r[] = {
        {"planar, 4 dword taps          ", run(k_planar_dword, HW / 256, f, out, reps)},
        {"planar, 2 unaligned 8B pairs  ", run(k_planar_pair, HW / 256, f, out, reps)},
        {"planar, 16B per row, 2 px/lane", run(k_planar_quad2px, HW / 512, f, out, reps)},
        {"planar, right taps + DPP shift", run(k_planar_dpp, HW / 256, f, out, reps)},
        {"chan-last, lane=pixel, 16B x8 ", run(k_cl_lane_pixel, HW / 256, f, out, reps)},
        {"chan-last, 4 lanes/pixel, 16B ", run(k_cl_lane_piece, HW / 64, f, out, reps)},
        {"LDS-staged rows, ds_read taps ", run(k_lds_staged, HW / 256, f, out, reps)},
    };
    printf("%-34s %9s %12s %14s\n", "variant (C=8, D=4, 1152x1536)", "ms", "Gtaps/s", "tap B/clk/CU");
    for (auto& x : r) printf("%-34s %9.4f %12.1f %14.2f\n", x.name, x.ms, taps / x.ms / 1e6, taps * 4 / (x.ms * 1e-3) / (256 * 2.4e9));
    return 0;
}
