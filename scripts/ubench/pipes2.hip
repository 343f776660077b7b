// Microbenchmark, round 4: issue rates of the gfx950 SIMD pipes at 1 / 2 / 4 waves per SIMD and how they overlap
//   (a) inside ONE wave (an MFMA followed by n filler instructions of one kind), and
//   (b) ACROSS waves of one SIMD (half the waves run role X, half role Y).
// the round-3 version (overlap.hip) ran 1-2 waves per SIMD, one VALU opcode, no s_setprio.
// Every instruction is an `asm volatile` statement, so the issue order in the binary is the order written here.
// One workgroup per CU (grid = 256), 256 x W work-items: waves w, w+4, w+8 ... share a SIMD.  Times are shader cycles
// read with s_memtime by every wave (average over waves), so DVFS does not enter.
//     hipcc --offload-arch=gfx950 -O3 pipes2.hip -o pipes2 && ./pipes2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Role { IDLE = 0, FMA, PKFMA, EXP, EXP16, CVT, MAX3, LDS128, LDS64, M16K32, M16K16, M32K16, M32K8, SALU,
            // one wave: 8 x { MFMA 16x16x32, n fillers }
            M16_FMA1, M16_FMA2, M16_FMA4, M16_EXP1, M16_EXP2, M16_EXP4, M16_LDS1, M16_LDS2, M16_CVT2, M16_MIX,
            M32_FMA4, M32_FMA8, M32_EXP4, M32_MIX, NROLES };
static const char* kNames[] = {"idle", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_exp_f16", "v_cvt_pkrtz", "v_max3_f32", "ds_read_b128", "ds_read_b64",
                               "mfma16x16x32f16", "mfma16x16x16f16", "mfma32x32x16f16", "mfma32x32x8f16", "s_mul",
                               "m16+1fma", "m16+2fma", "m16+4fma", "m16+1exp", "m16+2exp", "m16+4exp", "m16+1lds128", "m16+2lds128", "m16+2cvt", "m16+2exp+2fma+1cvt+1lds64",
                               "m32+4fma", "m32+8fma", "m32+4exp", "m32+4exp+4fma+2cvt+2lds64"};
// instructions per loop iteration (for the rate read-out): {primary, filler}
struct Cnt { int prim, fill; };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(X) REP8(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int R>
__device__ __forceinline__ float body(int iters, const char* lds) {
    const int lane = threadIdx.x & 63;
    float v[16], t[16];
    for (int k = 0; k < 16; ++k) { v[k] = 0.25f + 0.001f * (lane + k); t[k] = 0.f; }
    const float ca = 0.999f, cb = 0.001f;
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(0.01f * (lane + e)); b8[e] = (_Float16)(0.02f * (lane - e)); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = (f32x4){0, 0, 0, 0};
    f32x16 cc[2];
    for (int k = 0; k < 2; ++k) for (int e = 0; e < 16; ++e) cc[k][e] = 0.f;
    f32x4 ld[4];
    for (int k = 0; k < 4; ++k) ld[k] = (f32x4){0, 0, 0, 0};
    f32x2 ld2[4];
    for (int k = 0; k < 4; ++k) ld2[k] = (f32x2){0, 0};
    f32x2 pv[8];
    for (int k = 0; k < 8; ++k) pv[k] = (f32x2){0.25f, 0.5f};
    const f32x2 pa = {0.999f, 0.999f}, pb = {0.001f, 0.001f};
    unsigned pk[8];
    for (int k = 0; k < 8; ++k) pk[k] = 0;
    const unsigned la = (unsigned)(size_t)0 + lane * 16;          // LDS byte address (the kernel's only __shared__ array starts at 0)
    int sx = iters;

#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(ca), "v"(cb));
#define PKF(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[(k) & 7]) : "v"(pa), "v"(pb));
#define EXPI(k) asm volatile("v_exp_f32 %0, %1" : "+v"(t[k]) : "v"(v[k]));
#define EXPH(k) asm volatile("v_exp_f16 %0, %1" : "+v"(t[k]) : "v"(v[k]));
#define CVTI(k) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "+v"(pk[(k) & 7]) : "v"(v[k]), "v"(v[(k + 1) & 15]));
#define MX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(t[k]) : "v"(v[k]), "v"(v[(k + 1) & 15]));
#define L128(k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(ld[(k) & 3]) : "v"(la), "n"(((k) & 7) * 1024));
#define L64(k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "+v"(ld2[(k) & 3]) : "v"(la), "n"(((k) & 7) * 1024));
#define M16(k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[(k) & 7]) : "v"(a8), "v"(b8));
#define M16L(k) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c[(k) & 7]) : "v"(a4), "v"(b4));
#define M32(k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cc[(k) & 1]) : "v"(a8), "v"(b8));
#define M32L(k) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(cc[(k) & 1]) : "v"(a4), "v"(b4));
#define SMUL(k) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sx));
#define WAITL asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    for (int i = 0; i < iters; ++i) {
        if (R == FMA) { REP16(FMA) }
        if (R == PKFMA) { REP16(PKF) }
        if (R == EXP) { REP16(EXPI) }
        if (R == EXP16) { REP16(EXPH) }
        if (R == CVT) { REP16(CVTI) }
        if (R == MAX3) { REP16(MX3) }
        if (R == LDS128) { REP16(L128) WAITL }
        if (R == LDS64) { REP16(L64) WAITL }
        if (R == M16K32) { REP8(M16) }
        if (R == M16K16) { REP8(M16L) }
        if (R == M32K16) { REP8(M32) }
        if (R == M32K8) { REP8(M32L) }
        if (R == SALU) { REP16(SMUL) }
#define G_FMA1(k) M16(k) FMA(k)
#define G_FMA2(k) M16(k) FMA(k) FMA(k + 8)
#define G_FMA4(k) M16(k) FMA(k) FMA(k + 8) FMA((k + 4) & 7) FMA(((k + 4) & 7) + 8)
#define G_EXP1(k) M16(k) EXPI(k)
#define G_EXP2(k) M16(k) EXPI(k) EXPI(k + 8)
#define G_EXP4(k) M16(k) EXPI(k) EXPI(k + 8) EXPI((k + 4) & 7) EXPI(((k + 4) & 7) + 8)
#define G_LDS1(k) M16(k) L128(k)
#define G_LDS2(k) M16(k) L128(k) L128(k + 4)
#define G_CVT2(k) M16(k) CVTI(k) CVTI(k + 8)
#define G_MIX(k) M16(k) EXPI(k) FMA(k + 8) EXPI((k + 4) & 7) FMA(((k + 4) & 7) + 8) CVTI(k) L64(k)
        if (R == M16_FMA1) { REP8(G_FMA1) }
        if (R == M16_FMA2) { REP8(G_FMA2) }
        if (R == M16_FMA4) { REP8(G_FMA4) }
        if (R == M16_EXP1) { REP8(G_EXP1) }
        if (R == M16_EXP2) { REP8(G_EXP2) }
        if (R == M16_EXP4) { REP8(G_EXP4) }
        if (R == M16_LDS1) { REP8(G_LDS1) WAITL }
        if (R == M16_LDS2) { REP8(G_LDS2) WAITL }
        if (R == M16_CVT2) { REP8(G_CVT2) }
        if (R == M16_MIX) { REP8(G_MIX) WAITL }
#define H_FMA4(k) M32(k) FMA(k) FMA(k + 8) FMA((k + 4) & 7) FMA(((k + 4) & 7) + 8)
#define H_FMA8(k) H_FMA4(k) FMA((k + 1) & 7) FMA(((k + 1) & 7) + 8) FMA((k + 5) & 7) FMA(((k + 5) & 7) + 8)
#define H_EXP4(k) M32(k) EXPI(k) EXPI(k + 8) EXPI((k + 4) & 7) EXPI(((k + 4) & 7) + 8)
#define H_MIX(k) H_EXP4(k) FMA((k + 1) & 7) FMA(((k + 1) & 7) + 8) FMA((k + 5) & 7) FMA(((k + 5) & 7) + 8) CVTI(k) CVTI(k + 8) L64(k) L64(k + 4)
        if (R == M32_FMA4) { REP8(H_FMA4) }
        if (R == M32_FMA8) { REP8(H_FMA8) }
        if (R == M32_EXP4) { REP8(H_EXP4) }
        if (R == M32_MIX) { REP8(H_MIX) WAITL }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = (float)sx;
    for (int k = 0; k < 16; ++k) s += v[k] + t[k];
    for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][3] + pv[k][0] + pv[k][1] + (float)pk[k];
    for (int k = 0; k < 2; ++k) s += cc[k][0] + cc[k][15];
    for (int k = 0; k < 4; ++k) s += ld[k][0] + ld[k][3] + ld2[k][1];
    return s;
}

static Cnt counts(int r) {
    switch (r) {
        case IDLE: return {0, 0};
        case LDS128: case LDS64: case FMA: case PKFMA: case EXP: case EXP16: case CVT: case MAX3: case SALU: return {16, 0};
        case M16K32: case M16K16: case M32K16: case M32K8: return {8, 0};
        case M16_FMA1: case M16_EXP1: case M16_LDS1: return {8, 8};
        case M16_FMA2: case M16_EXP2: case M16_LDS2: case M16_CVT2: return {8, 16};
        case M16_FMA4: case M16_EXP4: return {8, 32};
        case M16_MIX: return {8, 48};
        case M32_FMA4: case M32_EXP4: return {8, 32};
        case M32_FMA8: return {8, 64};
        case M32_MIX: return {8, 96};
    }
    return {0, 0};
}

// waves of group (wave >> 2) & 1 == 0 run X, the others Y; PRIO: s_setprio 1 on the X waves (static, before the loop)
template <int X, int Y, int PRIO>
__global__ __launch_bounds__(1024) void k(int ix, int iy, float* out, unsigned long long* cyc) {
    __shared__ char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool isx = ((wave >> 2) & 1) == 0;
    float s;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (isx) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        s = body<X>(ix, lds);
    } else {
        s = body<Y>(iy, lds);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;

// returns average cycles of the X waves and of the Y waves
template <int X, int Y, int PRIO = 0>
void run(int waves_per_simd, int ix, int iy, double& cx, double& cy) {
    const int threads = 256 * waves_per_simd;
    hipMemset(g_cyc, 0, 256 * 16 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<X, Y, PRIO>), dim3(256), dim3(threads), 0, 0, ix, iy, g_out, g_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), g_cyc, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double sx = 0, sy = 0; int nx = 0, ny = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 4 * waves_per_simd; ++w) {
            if (((w >> 2) & 1) == 0) { sx += (double)h[b * 16 + w]; ++nx; } else { sy += (double)h[b * 16 + w]; ++ny; }
        }
    cx = nx ? sx / nx : 0; cy = ny ? sy / ny : 0;
}

template <int X>
void single(const char* label) {
    const int it = 2000;
    Cnt c = counts(X);
    printf("%-28s", label);
    for (int w : {1, 2, 4}) {
        double cx, cy;
        // all waves run X: with w == 1 only waves 0-3 exist; for w >= 2 both halves run X
        if (w == 1) run<X, IDLE>(1, it, 0, cx, cy); else { run<X, X>(w, it, it, cx, cy); cx = 0.5 * (cx + cy); }
        const double per_iter = cx / it;                       // cycles per loop iteration per wave, w waves per SIMD share the SIMD
        // SIMD-level issue interval of one primary instruction = cycles per iteration / (w waves x prim per iteration)
        printf("  W=%d: %7.2f cyc/iter/wave  %6.2f cyc per prim per SIMD", w, per_iter, per_iter / (w * c.prim));
    }
    printf("   [%d prim + %d filler per iter]\n", c.prim, c.fill);
}

template <int X, int Y, int PRIO = 0>
void pair(const char* label, int w) {
    const int it = 2000;
    double ax, ay, bx, by, cx, cy;
    run<X, IDLE>(w, it, 0, ax, ay);            // X waves alone (the Y slots idle)
    run<IDLE, Y>(w, 0, it, bx, by);            // Y alone
    run<X, Y, PRIO>(w, it, it, cx, cy);        // together
    printf("%-44s W=%d(%d+%d)  X alone %8.0f  Y alone %8.0f  | together X %8.0f (x%.2f)  Y %8.0f (x%.2f)  max/sum-of-alone = %.2f\n", label, w, w / 2, w / 2,
           ax, by, cx, cx / ax, cy, cy / by, (cx > cy ? cx : cy) / (ax + by));
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_cyc, 256 * 16 * 8);
    printf("== single role, all waves the same; cycles are s_memtime ticks ==\n");
    single<FMA>(kNames[FMA]); single<PKFMA>(kNames[PKFMA]); single<EXP>(kNames[EXP]); single<EXP16>(kNames[EXP16]); single<CVT>(kNames[CVT]);
    single<MAX3>(kNames[MAX3]); single<LDS128>(kNames[LDS128]); single<LDS64>(kNames[LDS64]); single<SALU>(kNames[SALU]);
    single<M16K32>(kNames[M16K32]); single<M16K16>(kNames[M16K16]); single<M32K16>(kNames[M32K16]); single<M32K8>(kNames[M32K8]);
    printf("== one wave = MFMA + fillers (cyc per prim per SIMD = cycles one MFMA and its fillers take) ==\n");
    single<M16_FMA1>(kNames[M16_FMA1]); single<M16_FMA2>(kNames[M16_FMA2]); single<M16_FMA4>(kNames[M16_FMA4]);
    single<M16_EXP1>(kNames[M16_EXP1]); single<M16_EXP2>(kNames[M16_EXP2]); single<M16_EXP4>(kNames[M16_EXP4]);
    single<M16_LDS1>(kNames[M16_LDS1]); single<M16_LDS2>(kNames[M16_LDS2]); single<M16_CVT2>(kNames[M16_CVT2]); single<M16_MIX>(kNames[M16_MIX]);
    single<M32_FMA4>(kNames[M32_FMA4]); single<M32_FMA8>(kNames[M32_FMA8]); single<M32_EXP4>(kNames[M32_EXP4]); single<M32_MIX>(kNames[M32_MIX]);
    printf("== role pairs across the waves of a SIMD (half X, half Y) ==\n");
    for (int w : {2, 4}) {
        pair<M16K32, FMA>("mfma16 || v_fma", w);
        pair<M16K32, EXP>("mfma16 || v_exp", w);
        pair<M16K32, LDS128>("mfma16 || ds_read_b128", w);
        pair<M16K32, CVT>("mfma16 || v_cvt_pkrtz", w);
        pair<M32K16, FMA>("mfma32 || v_fma", w);
        pair<M32K16, EXP>("mfma32 || v_exp", w);
        pair<M32K16, LDS128>("mfma32 || ds_read_b128", w);
        pair<FMA, EXP>("v_fma || v_exp", w);
        pair<FMA, LDS128>("v_fma || ds_read_b128", w);
        pair<FMA, SALU>("v_fma || s_mul", w);
        pair<M16K32, M16K32>("mfma16 || mfma16", w);
        pair<M16K32, FMA, 1>("mfma16 (prio 1) || v_fma", w);
        pair<FMA, M16K32, 1>("v_fma (prio 1) || mfma16", w);
    }
    return 0;
}
