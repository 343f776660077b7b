// Microbenchmark: do the matrix pipe and the other pipes of a gfx950 SIMD overlap ACROSS two waves of one SIMD?
// One workgroup of 512 work-items per CU = two waves per SIMD; waves 0-3 run workload X, waves 4-7 workload Y (or idle).
// Prints the wall time of X alone, Y alone and X || Y.   hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { IDLE = 0, MFMA = 1, VALU = 2, LDSR = 3, MFMA_SELF_VALU = 4, MFMA32 = 5, MFMA32_SELF_VALU = 6, SALU = 7, MFMA_AGPR = 8, MFMA_AGPR_SELF_VALU = 9 };
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int W>
__device__ __forceinline__ void work(int iters, float* out, char* lds) {
    const int lane = threadIdx.x & 63;
    if (W == MFMA || W == MFMA_SELF_VALU) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = (f32x4){0, 0, 0, 0};
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = (float)lane + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
                if (W == MFMA_SELF_VALU) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) v[(k + j) & 7] = __builtin_fmaf(v[(k + j) & 7], 1.0001f, 0.5f);
                }
            }
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3] + v[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (W == MFMA_AGPR || W == MFMA_AGPR_SELF_VALU) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = (f32x4){0, 0, 0, 0};
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = (float)lane + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[k]) : "v"(a), "v"(b));
                if (W == MFMA_AGPR_SELF_VALU) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) v[(k + j) & 7] = __builtin_fmaf(v[(k + j) & 7], 1.0001f, 0.5f);
                }
            }
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3] + v[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (W == MFMA32 || W == MFMA32_SELF_VALU) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) c[k][e] = 0.0f;
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = (float)lane + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
                if (W == MFMA32_SELF_VALU) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[(2 * k + j) & 7] = __builtin_fmaf(v[(2 * k + j) & 7], 1.0001f, 0.5f);
                }
            }
        }
        float s = 0;
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += c[k][e];
        for (int k = 0; k < 8; ++k) s += v[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (W == SALU) {
        int x = iters;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { x = x * 3 + 1; asm volatile("" : "+s"(x)); }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)x;
    } else if (W == VALU) {
        float v[24];
        for (int k = 0; k < 24; ++k) v[k] = (float)lane + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        }
        float s = 0;
        for (int k = 0; k < 24; ++k) s += v[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (W == LDSR) {
        f32x4 acc = {0, 0, 0, 0};
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += p[k * 64];
            asm volatile("" ::: "memory");
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    }
}

template <int X, int Y>
__global__ __launch_bounds__(512) void k(int ix, int iy, float* out) {
    __shared__ char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    if (threadIdx.x < 256) work<X>(ix, out, lds);
    else work<Y>(iy, out, lds);
}

template <int X, int Y>
float run(int ix, int iy, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<X, Y>), dim3(256), dim3(512), 0, 0, ix, iy, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<X, Y>), dim3(256), dim3(512), 0, 0, ix, iy, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1000;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int IM = 4000;          // 32000 MFMAs per wave: 512 K cycles at 16 cycles each
    const int IV = 5300;          // 127 K VALU: ~509 K cycles at 4 cycles each
    const int IL = 8000;          // 64 K ds_read_b128
    printf("MFMA alone                 %8.1f us\n", run<MFMA, IDLE>(IM, 0, out));
    printf("VALU alone                 %8.1f us\n", run<VALU, IDLE>(IV, 0, out));
    printf("LDS-read alone             %8.1f us\n", run<LDSR, IDLE>(IL, 0, out));
    printf("MFMA || MFMA (2 waves)     %8.1f us\n", run<MFMA, MFMA>(IM, IM, out));
    printf("VALU || VALU               %8.1f us\n", run<VALU, VALU>(IV, IV, out));
    printf("MFMA || VALU               %8.1f us\n", run<MFMA, VALU>(IM, IV, out));
    printf("MFMA || LDS-read           %8.1f us\n", run<MFMA, LDSR>(IM, IL, out));
    printf("VALU || LDS-read           %8.1f us\n", run<VALU, LDSR>(IV, IL, out));
    printf("MFMA+3 VALU in one wave    %8.1f us\n", run<MFMA_SELF_VALU, IDLE>(IM, 0, out));
    printf("(MFMA+3 VALU) || same      %8.1f us\n", run<MFMA_SELF_VALU, MFMA_SELF_VALU>(IM, IM, out));
    printf("MFMA32x32x16 alone         %8.1f us  (4000 x 4 MFMAs of 32 cycles)\n", run<MFMA32, IDLE>(IM, 0, out));
    printf("MFMA32 || VALU             %8.1f us\n", run<MFMA32, VALU>(IM, IV, out));
    printf("MFMA32 || LDS-read         %8.1f us\n", run<MFMA32, LDSR>(IM, IL, out));
    printf("MFMA32+6 VALU in one wave  %8.1f us  (the same 96 K VALU as the 16x16 case)\n", run<MFMA32_SELF_VALU, IDLE>(IM, 0, out));
    printf("MFMA(acc in AGPR) alone    %8.1f us\n", run<MFMA_AGPR, IDLE>(IM, 0, out));
    printf("MFMA(AGPR) || VALU         %8.1f us\n", run<MFMA_AGPR, VALU>(IM, IV, out));
    printf("MFMA(AGPR) || LDS-read     %8.1f us\n", run<MFMA_AGPR, LDSR>(IM, IL, out));
    printf("MFMA(AGPR)+3 VALU one wave %8.1f us\n", run<MFMA_AGPR_SELF_VALU, IDLE>(IM, 0, out));
    printf("SALU alone                 %8.1f us\n", run<SALU, IDLE>(8000, 0, out));
    printf("MFMA || SALU               %8.1f us\n", run<MFMA, SALU>(IM, 8000, out));
    printf("VALU || SALU               %8.1f us\n", run<VALU, SALU>(IV, 8000, out));
    return 0;
}
