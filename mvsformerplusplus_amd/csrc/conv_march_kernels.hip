// Row-marching form of the 16 -> 16 stride-1 3x3x3 convolution of the split-bf16 U-Net (module.py:89-126 Conv3d + folded BN +
// ReLU; the layers conv2 of CostRegNet / CostRegNet3D, module.py:372-383, 457-477), split activation format in and out.
//
// Why (round 3, ISA census of the one-tile kernel conv3d_mfma_bf16x3_kernel<16,16>): a 4x4x16 tile costs a wave ~700 VALU
// instructions against 168 MFMAs - 380 of them are address arithmetic for the 1296 staged 32-byte runs of the 6x6x18 halo tile
// (2.5x the tile's own voxels), ~90 the per-step tap-offset selects - and every workgroup pays launch, one exposed HBM latency
// and a store drain for 2.7 K MFMA cycles of work per wave.  Here a workgroup owns a cross-section of TD z-planes x 30 columns
// and MARCHES along y:
//   * per input row it stages exactly one y-row of the cross-section (+ halo columns / planes): 1.2-1.6x amplification, one
//     pointer increment per staged run, the loads of row r + 1 in flight under the MFMAs of row r (double-buffered row image);
//   * every B fragment read from LDS feeds all THREE kh taps: input row r contributes to output rows r - 1, r, r + 1 (three
//     rotating accumulator sets) - 9 MFMAs per 16-byte operand read instead of 3;
//   * all packed weights (3 kh x 5 steps x hi/lo = 30 A fragments, 120 VGPRs) stay in registers for the life of the workgroup
//     (gathered once from the tile kernels' packed layout: no second weight format), the per-step LDS offsets are computed once:
//     the row loop issues no weight load and no tap arithmetic.
// Contraction per input row and wave: k = (kd, kw, cin) = 144 -> 5 steps of 32 (the tenth tap is zero weights), x 3 kh x 3
// split terms x 2 column groups = 90 MFMAs; wave w owns z-plane w of the cross-section, both 16-column groups (columns 30, 31
// of the 32 are computed and discarded).
#include "conv_cfg.h"
#include "split_format.h"

#include <cstdlib>
#include <type_traits>

#ifndef MVS_MARCH_ABL
#define MVS_MARCH_ABL 0            // ablation (scripts/bench_layer.py): 1 no activation loads, 2 no output stores, 3 one MFMA term of three
#endif

namespace mvs {

template <int TD_>
struct MarchCfg {
    static constexpr int TD = TD_;                     // z-planes per workgroup = waves
    static constexpr int TW = 30, PW = 32;             // output columns, staged columns (x0 - 1 .. x0 + 30)
    static constexpr int ZR = TD + 2;                  // staged z-rows (z0 - 1 .. z0 + TD)
    static constexpr int NVOX = ZR * PW;
    static constexpr int SB = 32;                      // bytes per voxel within an octet plane: [hi x8 | lo x8]
    // octet planes offset by 16 B modulo the 256-byte bank row (conflict-free ds_read_b128 groups, as in BfConv); + 2 voxels: the
    // discarded columns 30, 31 of the last z-row read 2 voxels past it
    static constexpr int PLANE = ((NVOX + 2) * SB + 255) / 256 * 256 + 16;
    static constexpr int BUF = 2 * PLANE;              // one staged row (two octets)
    static constexpr int LDS_BYTES = 2 * BUF;          // double-buffered
    static_assert(TD == 4, "one wave per z-plane; the staging maps 256 work-items onto TD * 32 * 2 inner runs");
};

// grid = (column strips * z tiles * y segments, B)
template <class Cfg>
__global__ __launch_bounds__(256, 2) void conv3d_march16_kernel(const float* __restrict__ x, const void* __restrict__ wp, const float* __restrict__ bias,
                                                             float* __restrict__ y, int D, int H, int W, int relu, int nstrip, int nzt, int SH) {
    constexpr int TD = Cfg::TD, PW = Cfg::PW, SB = Cfg::SB, PLANE = Cfg::PLANE, BUF = Cfg::BUF;
    HIP_DYNAMIC_SHARED(float4, lds4)
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    int blk = (int)blockIdx.x;
    const int strip = blk % nstrip;
    blk /= nstrip;
    const int zt = blk % nzt, seg = blk / nzt;
    const int x0 = strip * Cfg::TW, z0 = zt * TD, y0 = seg * SH;
    const int y1 = y0 + SH < H ? y0 + SH : H;
    const int b = (int)blockIdx.y;
    const size_t plane_elems = (size_t)H * W * 16;
    const char* xb = reinterpret_cast<const char*>(x + (size_t)b * D * plane_elems);
    float* yb = y + (size_t)b * D * plane_elems;
    const unsigned rowbytes = (unsigned)W * 64u;

    // ---- all packed weights of the layer, gathered once from the tile kernels' layout: step T of that layout covers channel
    //      octets 4T .. 4T + 3 of the (tap-major) k-range, lane group g' owns octet 4T + g'; here lane group g owns
    //      (tap j = 2s + (g >> 1) of the 9 (kd, kw) taps, octet g & 1) for each kh ----
    bf16x8 wh[3][5], wl[3][5];
    {
        const bf16x8* wq = reinterpret_cast<const bf16x8*>(wp);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int j = 2 * s + (g >> 1);
                const int jj = j < 9 ? j : 8;
                const int kd = jj / 3, kw = jj - kd * 3;
                const int o = ((kd * 3 + kh) * 3 + kw) * 2 + (g & 1);          // channel octet in the tile kernels' k order
                const int T = o >> 2, go = o & 3;
                bf16x8 h = wq[(size_t)(T * 2) * 64 + li + 16 * go], l = wq[(size_t)(T * 2 + 1) * 64 + li + 16 * go];
                if (j >= 9) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { h[e] = (__bf16)0.0f; l[e] = (__bf16)0.0f; }
                }
                wh[kh][s] = h;
                wl[kh][s] = l;
#ifndef MVS_NO_OPAQUE_VEC
                asm volatile("" : "+v"(wh[kh][s]), "+v"(wl[kh][s]));           // keep them in registers (never re-load the invariant memory)
#endif
            }
    }
    const float4 bb = *reinterpret_cast<const float4*>(bias + 4 * g);

    // ---- staging.  Inner z-rows: item = (z-row 1 .. TD, column, octet) = 32 contiguous bytes, one per work-item; consecutive
    //      work-items = consecutive runs of a z-row.  The two halo z-rows (z0 - 1 and z0 + TD): 128 runs = 256 16-byte pieces, one
    //      per work-item.  32-bit byte offsets from the batch item's base (conv3d_march_usable checks the volume size). ----
    unsigned soff0, soff1;
    bool itv0, itv1;
    int dst0, dst1;
    {
        const int zr = 1 + (tid >> 6), col = (tid >> 1) & 31, oc = tid & 1;
        const int z = z0 - 1 + zr, xx = x0 - 1 + col;
        itv0 = z < D && xx >= 0 && xx < W;
        soff0 = itv0 ? (unsigned)((((long long)z * H + (y0 - 1)) * W + xx) * 64 + oc * 32) : 0u;   // row y0 - 1 may be -1: wraps, never dereferenced then
        dst0 = oc * PLANE + (zr * PW + col) * SB;
    }
    {
        const int hr = tid >> 1, half = tid & 1;
        const int zr = (hr >> 6) ? Cfg::ZR - 1 : 0, col = (hr >> 1) & 31, oc = hr & 1;
        const int z = z0 - 1 + zr, xx = x0 - 1 + col;
        itv1 = z >= 0 && z < D && xx >= 0 && xx < W;
        soff1 = itv1 ? (unsigned)((((long long)z * H + (y0 - 1)) * W + xx) * 64 + oc * 32 + half * 16) : 0u;
        dst1 = oc * PLANE + (zr * PW + col) * SB + half * 16;
    }
    // The loads are unconditional, from a clamped address, and the zeroing of out-of-volume runs happens at commit time: a select
    // (or a branch) next to the loads makes the compiler wait for them right there, ahead of the row's MFMAs.
    float4 su, sv, sh;
    bool okA = false, okB = false;
    auto issue = [&](int r) {                                       // loads of input row r
        const bool rowok = r >= 0 && r < H;
        okA = rowok && itv0;
        okB = rowok && itv1;
        const float4* p0 = reinterpret_cast<const float4*>(xb + (okA ? soff0 : 0u));
        const float4* p1 = reinterpret_cast<const float4*>(xb + (okB ? soff1 : 0u));
#if MVS_MARCH_ABL == 1                                              // ablation: no activation loads
        su = sv = sh = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        (void)p0; (void)p1;
#else
        su = p0[0];
        sv = p0[1];
        sh = p1[0];
#endif
        soff0 += rowbytes;
        soff1 += rowbytes;
    };
    auto commit = [&](int par) {                                    // zeros outside the volume
        char* d = ldsb + par * BUF;
        *reinterpret_cast<float4*>(d + dst0) = make_float4(okA ? su.x : 0.0f, okA ? su.y : 0.0f, okA ? su.z : 0.0f, okA ? su.w : 0.0f);
        *reinterpret_cast<float4*>(d + dst0 + 16) = make_float4(okA ? sv.x : 0.0f, okA ? sv.y : 0.0f, okA ? sv.z : 0.0f, okA ? sv.w : 0.0f);
        *reinterpret_cast<float4*>(d + dst1) = make_float4(okB ? sh.x : 0.0f, okB ? sh.y : 0.0f, okB ? sh.z : 0.0f, okB ? sh.w : 0.0f);
    };

    // ---- B operand offsets of the 5 steps (this lane's tap and octet), relative to a row image ----
    int boff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int j = 2 * s + (g >> 1), jj = j < 9 ? j : 8;
        const int kd = jj / 3, kw = jj - kd * 3;
        boff[s] = (g & 1) * PLANE + ((wave + kd) * PW + li + kw) * SB;
    }

    f32x4 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[a][n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // output addressing: lane (li, g) of column group n holds channels 4g .. 4g + 3 of voxel (z0 + wave, y, x0 + 16 n + li)
    const int zo = z0 + wave;
    bool outv[2];
    unsigned ooff[2];                                              // in floats, row y = 0
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 16 * n + li;
        outv[n] = zo < D && c < Cfg::TW && x0 + c < W;
        ooff[n] = outv[n] ? (unsigned)((((long long)zo * H) * W + x0 + c) * 16 + (g >> 1) * 8) : 0u;
    }

    const int nrows = y1 - y0 + 2;                                 // input rows y0 - 1 .. y1
    issue(y0 - 1);
    commit(0);
    __syncthreads();

    // one input row: q = local row index (r = y0 - 1 + q); PH = q % 6 fixes the row image (q & 1) and the accumulator rotation
    // (q % 3) at compile time.  Input row r feeds output row r - 1 + (2 - kh): accumulator set (q + 2 - kh) % 3 ... i.e. kh = 0 -> the
    // set of output row r + 1, kh = 2 -> the set of output row r - 1, which is complete after this row.
    auto row = [&](auto phase, int q) {
        constexpr int PH = decltype(phase)::value;
        constexpr int PAR = PH & 1, R3 = PH % 3;
        const int r = y0 - 1 + q;
        if (q + 1 < nrows) issue(r + 1);
        const char* img = ldsb + PAR * BUF;
        bf16x8 bh[2][2], bl[2][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bh[0][n] = *reinterpret_cast<const bf16x8*>(img + boff[0] + n * 16 * SB);
            bl[0][n] = *reinterpret_cast<const bf16x8*>(img + boff[0] + n * 16 * SB + 16);
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int cur = MVS_MARCH_ABL == 7 ? 0 : (s & 1), nxt = cur ^ 1;
            if (s + 1 < 5 && MVS_MARCH_ABL != 7) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bh[nxt][n] = *reinterpret_cast<const bf16x8*>(img + boff[s + 1] + n * 16 * SB);
                    bl[nxt][n] = *reinterpret_cast<const bf16x8*>(img + boff[s + 1] + n * 16 * SB + 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // term-outer: consecutive MFMAs go to six different accumulators
#if MVS_MARCH_ABL != 3
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4& a = acc[(R3 + 3 - kh) % 3][n];          // output row r + 1 - kh lives in set (q + 1 - kh) % 3 (q = r - y0 + 1)
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[kh][s], bh[cur][n], a, 0, 0, 0);
                }
#endif
#if MVS_MARCH_ABL != 3
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4& a = acc[(R3 + 3 - kh) % 3][n];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[kh][s], bl[cur][n], a, 0, 0, 0);
                }
#endif
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4& a = acc[(R3 + 3 - kh) % 3][n];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[kh][s], bh[cur][n], a, 0, 0, 0);
                }
        }
        // the next row's image first (its loads were issued before the MFMAs; vmcnt retires in order, so the epilogue's stores must
        // come after this wait, not before it), then output row r - 1 (kh = 2 was its last contribution): set (R3 + 1) % 3
        if (q + 1 < nrows && MVS_MARCH_ABL != 5) commit(PAR ^ 1);
        constexpr int DONE = (R3 + 1) % 3;
        const int yo = r - 1;
        const bool emit = yo >= y0 && yo < y1 && (MVS_MARCH_ABL != 2 || acc[DONE][0][0] == 12345.678f);     // ablation 2: no stores
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x4& a = acc[DONE][n];
#if MVS_MARCH_ABL == 4
            if (a[0] != 12345.678f) { a = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; continue; }
#endif
            float4 v = make_float4(a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w);
            if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            split_store_quad(yb + (ooff[n] + (unsigned)(emit ? yo : 0) * (unsigned)W * 16u), g, v, emit && outv[n]);
            a = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
#if MVS_MARCH_ABL != 6
        __syncthreads();
#endif
    };
    for (int q = 0; q < nrows; ++q) {
        switch (q % 6) {
            case 0: row(std::integral_constant<int, 0>(), q); break;
            case 1: row(std::integral_constant<int, 1>(), q); break;
            case 2: row(std::integral_constant<int, 2>(), q); break;
            case 3: row(std::integral_constant<int, 3>(), q); break;
            case 4: row(std::integral_constant<int, 4>(), q); break;
            default: row(std::integral_constant<int, 5>(), q); break;
        }
    }
}

// smallest volume (voxels per batch item) the marching form is used for; below it the launch cannot fill the chip with segments of a
// useful height and the tile kernels win.  MVS_MARCH_MIN_VOXELS overrides it (tests run the marching form on small volumes).
static long long march_min_voxels() {
    const char* e = getenv("MVS_MARCH_MIN_VOXELS");            // read per call: the tests switch it between cases
    return e != nullptr ? atoll(e) : (1LL << 60);     // off by default: measured slower than the tile kernel (profiles/r03_conv_march_ab.txt)
}

bool conv3d_march_usable(int Cin, int Cout, int kd, int sd, int sh, int sw, int D, int H, int W, int split) {
#ifdef MVS_NO_MARCH
    return false;
#endif
    const long long nvox = (long long)D * H * W;
    return split && Cin == 16 && Cout == 16 && kd == 3 && sd == 1 && sh == 1 && sw == 1 && nvox >= march_min_voxels() && nvox * 64 < (1LL << 32);
}

int conv3d_march_bf16x3(const float* x, const void* wp, const float* bias, float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
    typedef MarchCfg<4> Cfg;
    const int nstrip = (int)ceil_div(W, Cfg::TW), nzt = (int)ceil_div(D, Cfg::TD);
    // segment height: the launch takes ceil(blocks / resident) rounds of (SH + 2) row iterations (2 warm-up rows per segment)
    const long long per_seg = (long long)nstrip * nzt * B, resident = 2 * 256;
    int SH = H;
    long long best = -1;
    for (int sg = 1; sg <= (H + 7) / 8; ++sg) {
        const int sh = (int)ceil_div(H, sg);
        const long long rounds = (per_seg * ceil_div(H, sh) + resident - 1) / resident;
        const long long cost = rounds * (sh + 2);
        if (best < 0 || cost < best) { best = cost; SH = sh; }
    }
    const int nseg = (int)ceil_div(H, SH);
    hipLaunchKernelGGL((conv3d_march16_kernel<Cfg>), dim3(nstrip * nzt * nseg, B), dim3(256), Cfg::LDS_BYTES, st, x, wp, bias, y, D, H, W, relu, nstrip,
                       nzt, SH);
    return check_launch("conv3d_march16_kernel");
}

}  // namespace mvs
