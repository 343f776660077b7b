// Feasibility microbenchmark (GPU box only) for a workgroup-level LDS-staged gather: same arithmetic as the warp kernels
// (bilinear tap pairs of 2 rows x 4 depth planes per pixel and channel, times the reference feature), stage-4 shapes
// (1152x1536, C = 8, 4 source views).  Variant G gathers straight from global memory with 8-byte pair loads (what the
// shipped kernels do); variant L stages the footprint of a 64x4 pixel tile (all 8 channels) into LDS with coalesced dword
// loads, double-buffered over the source views, and takes its taps with LDS reads.
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_lds_ubench.hip -o scripts/gather_lds_ubench.bin && scripts/gather_lds_ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int H = 1152, W = 1536, HW = H * W, C = 8, D = 4, V = 4;
constexpr int TW = 64, TH = 4;                 // pixel tile of a 256-thread block
constexpr int MARGIN = 8;                      // source columns beyond the tile that the 4 planes may reach
constexpr int FW = TW + MARGIN, FH = TH + 1;   // staged footprint (floats per row, rows)
struct __attribute__((packed, aligned(4))) F2 { float x, y; };

// source column of pixel (x, y) on plane d of view v: x + small positive shift, smooth plus per-pixel jitter
__device__ __forceinline__ float src_x(int x, int y, int d, int v) {
    const float jit = (float)(((x * 7 + y * 13) & 15)) * (1.0f / 16.0f);       // 0 .. 0.94
    return (float)x + 0.3f * (float)(d + 1) * (float)(v + 1) * 0.29f + 0.5f + jit;
}

__global__ __launch_bounds__(256) void k_global(const float* __restrict__ f, const float* __restrict__ ref, float* __restrict__ out) {
    const int tx = blockIdx.x % (W / TW), ty = blockIdx.x / (W / TW);
    const int x = tx * TW + (threadIdx.x & 63), y = ty * TH + (threadIdx.x >> 6);
    const int p = y * W + x;
    float total = 0.f;
    for (int v = 0; v < V; ++v) {
        const float* fv = f + (size_t)v * C * HW;
        unsigned top[D], bot[D];
        float w0[D], w1[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float sx = src_x(x, y, d, v);
            int xb = (int)floorf(sx);
            w1[d] = sx - (float)xb; w0[d] = 1.f - w1[d];
            xb = xb > W - 2 ? W - 2 : xb;
            const int yb = y + 1 < H ? y + 1 : H - 1;
            top[d] = y * W + xb; bot[d] = yb * W + xb;
        }
        float acc[D] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* sp = fv + (size_t)c * HW;
            const float r = ref[(size_t)c * HW + p];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const F2 t = *reinterpret_cast<const F2*>(sp + top[d]), b = *reinterpret_cast<const F2*>(sp + bot[d]);
                float wv = 0.7f * w0[d] * t.x;
                wv += 0.7f * w1[d] * t.y;
                wv += 0.3f * w0[d] * b.x;
                wv += 0.3f * w1[d] * b.y;
                acc[d] += r * wv;
            }
        }
        total += acc[0] + acc[1] + acc[2] + acc[3];
    }
    out[p] = total;
}

__global__ __launch_bounds__(256) void k_lds(const float* __restrict__ f, const float* __restrict__ ref, float* __restrict__ out) {
    __shared__ float tile[2][C][FH][FW];                        // 2 x 8 x 5 x 72 floats = 23 KB
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % (W / TW), ty = blockIdx.x / (W / TW);
    const int x0 = tx * TW, y0 = ty * TH;
    const int lx = tid & 63, ly = tid >> 6;
    const int x = x0 + lx, y = y0 + ly;
    const int p = y * W + x;
    constexpr int NE = (C * FH * FW + 255) / 256;               // staged elements per work-item
    float stg[NE];
    auto issue = [&](int v) {
        const float* fv = f + (size_t)v * C * HW;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            const int col = e % FW, r2 = e / FW, row = r2 % FH, c = r2 / FH;
            int gx = x0 + col, gy = y0 + row;
            gx = gx < W ? gx : W - 1; gy = gy < H ? gy : H - 1;
            stg[i] = c < C ? fv[(size_t)c * HW + gy * W + gx] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        float* dst = &tile[buf][0][0][0];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            if (e < C * FH * FW) dst[e] = stg[i];
        }
    };
    issue(0);
    commit(0);
    __syncthreads();
    float rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) rr[c] = ref[(size_t)c * HW + p];
    float total = 0.f;
    for (int v = 0; v < V; ++v) {
        const int buf = v & 1;
        if (v + 1 < V) issue(v + 1);                            // global loads in flight during this view's math
        int top[D], bot[D];
        float w0[D], w1[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float sx = src_x(x, y, d, v);
            int xb = (int)floorf(sx);
            w1[d] = sx - (float)xb; w0[d] = 1.f - w1[d];
            xb = xb > W - 2 ? W - 2 : xb;
            const int yb = y + 1 < H ? y + 1 : H - 1;
            top[d] = ly * FW + (xb - x0); bot[d] = (yb - y0) * FW + (xb - x0);
        }
        float acc[D] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* sp = &tile[buf][c][0][0];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const F2 t = *reinterpret_cast<const F2*>(sp + top[d]), b = *reinterpret_cast<const F2*>(sp + bot[d]);
                float wv = 0.7f * w0[d] * t.x;
                wv += 0.7f * w1[d] * t.y;
                wv += 0.3f * w0[d] * b.x;
                wv += 0.3f * w1[d] * b.y;
                acc[d] += rr[c] * wv;
            }
        }
        total += acc[0] + acc[1] + acc[2] + acc[3];
        if (v + 1 < V) {
            commit(buf ^ 1);
            __syncthreads();
        }
    }
    out[p] = total;
}

template <class K>
float run(K kern, const float* f, const float* ref, float* out, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = (W / TW) * (H / TH);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, f, ref, out);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, f, ref, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    float *f, *ref, *o1, *o2;
    const size_t nf = (size_t)V * C * HW;
    CHECK(hipMalloc(&f, (nf + 4 * W) * 4));
    CHECK(hipMalloc(&ref, (size_t)C * HW * 4));
    CHECK(hipMalloc(&o1, (size_t)HW * 4));
    CHECK(hipMalloc(&o2, (size_t)HW * 4));
    std::vector<float> h(nf + 4 * W);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CHECK(hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ref, h.data() + 12345, (size_t)C * HW * 4, hipMemcpyHostToDevice));
    const float tg = run(k_global, f, ref, o1, 10), tl = run(k_lds, f, ref, o2, 10);
    std::vector<float> a(HW), b(HW);
    CHECK(hipMemcpy(a.data(), o1, (size_t)HW * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), o2, (size_t)HW * 4, hipMemcpyDeviceToHost));
    double md = 0;
    for (int i = 0; i < HW; ++i) { if ((i % W) < W - TW) md = fmax(md, fabs((double)a[i] - b[i])); }
    printf("global pair loads : %.3f ms\nLDS-staged tile   : %.3f ms   (%.2fx)   max |diff| %.2e\n", tg, tl, tg / tl, md);
    return 0;
}
