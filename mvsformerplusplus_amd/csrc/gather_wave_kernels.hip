// Host-side dispatch of the wave-autonomous gather passes (kernels: gather_wave.h).
#include "gather_wave.h"

namespace mvs {

// ------------------------------------------------------------------------------------------------
// host-side dispatch (called from the C entry points in warp_kernels.hip)
// ------------------------------------------------------------------------------------------------
static int gw_slots(int D) {
    const int nch = (D + GW_DCH - 1) / GW_DCH;
    return nch >= 8 ? 8 : (nch >= 4 ? 4 : (nch >= 2 ? 2 : 1));
}

bool gw_supported(int C, int G, int D, int H, int W) {
    if (G != 8 || !(C == 8 || C == 16 || C == 32 || C == 64)) return false;
    if (W % 8 != 0 || W < 8 || H < 2 || W > 65535 || H > 65535) return false;
    if ((long long)D * H * W > 0x7fffffffLL) return false;                    // 32-bit voxel offsets inside one batch item
    if ((long long)C * H * W * 4 > 0xffffffffLL) return false;                // one view behind a buffer descriptor
    return (size_t)D * (64 / gw_slots(D)) * sizeof(float) <= 16 * 1024;     // per-wave sim scratch of pass 1
}

template <int DT, int NOCT, int NS, bool TILED>
static int gw_launch_entropy_t(const void* feat, const float* hom, const float* hyp, float* ent, int B, int V, int D, int H, int W, int vb,
                               int ve, hipStream_t st) {
    typedef GwTile<NS> Tile;
    const int ntx = (int)ceil_div(W, Tile::BW), nty = (int)ceil_div(H, Tile::PH);
    const int nblk = ntx * nty;
    // plenty of tiles: one wave walks all views of its tile (hypotheses loaded once); few tiles (coarse stages): one workgroup per
    // (tile, view) so that the chip fills
    const int vpb = (long long)nblk * B >= 4096 ? ve - vb : 1;
    const bool direct = NS == 1 && D <= GW_DCH;
    const int wave_lds = (int)(GW_WIN_BYTES + (direct ? 0 : ((size_t)D * Tile::NP * sizeof(float) + 15) / 16 * 16));
    const size_t lds = (size_t)4 * wave_lds;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gw_entropy_kernel<DT, NOCT, NS, TILED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gw_entropy_kernel<DT, NOCT, NS, TILED>), dim3(nblk, ceil_div(ve - vb, vpb), B), dim3(256), lds, st, feat, hom, hyp, ent, V, D, H,
                       W, vb, ve, vpb, ntx, nblk, wave_lds);
    return check_launch("gw_entropy_kernel");
}

template <int DT, int NOCT, int NS, bool TILED>
static int gw_launch_aggregate_t(const void* feat, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                                 int normalise, int B, int V, int D, int H, int W, int vb, int ve, hipStream_t st) {
    typedef GwTile<NS> Tile;
    const int ntx = (int)ceil_div(W, Tile::BW), nty = (int)ceil_div(H, Tile::PH);
    const int nblk = ntx * nty;
    const int nch = (D + GW_DCH - 1) / GW_DCH, niter = (nch + NS - 1) / NS;
    const size_t lds = (size_t)4 * GW_WIN_BYTES;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gw_aggregate_kernel<DT, NOCT, NS, TILED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gw_aggregate_kernel<DT, NOCT, NS, TILED>), dim3(nblk, niter, B), dim3(256), lds, st, feat, hom, hyp, vis, vol, vis_sum,
                       normalise, V, D, H, W, vb, ve, ntx, nblk);
    return check_launch("gw_aggregate_kernel");
}

#define GW_DISPATCH_NS(FN, DTV, NOCTV, ...)                                                    \
    switch (gw_slots(D) * 2 + (layout == MVS_LAYOUT_OCTET_TILED ? 1 : 0)) {                    \
        case 2: return FN<DTV, NOCTV, 1, false>(__VA_ARGS__);                                  \
        case 3: return FN<DTV, NOCTV, 1, true>(__VA_ARGS__);                                   \
        case 4: return FN<DTV, NOCTV, 2, false>(__VA_ARGS__);                                  \
        case 5: return FN<DTV, NOCTV, 2, true>(__VA_ARGS__);                                   \
        case 8: return FN<DTV, NOCTV, 4, false>(__VA_ARGS__);                                  \
        case 9: return FN<DTV, NOCTV, 4, true>(__VA_ARGS__);                                   \
        case 16: return FN<DTV, NOCTV, 8, false>(__VA_ARGS__);                                 \
        default: return FN<DTV, NOCTV, 8, true>(__VA_ARGS__);                                  \
    }
#define GW_DISPATCH_C(FN, DTV, ...)                                                            \
    switch (C) {                                                                               \
        case 8: GW_DISPATCH_NS(FN, DTV, 1, __VA_ARGS__)                                        \
        case 16: GW_DISPATCH_NS(FN, DTV, 2, __VA_ARGS__)                                       \
        case 32: GW_DISPATCH_NS(FN, DTV, 4, __VA_ARGS__)                                       \
        default: GW_DISPATCH_NS(FN, DTV, 8, __VA_ARGS__)                                       \
    }
#define GW_DISPATCH(FN, ...)                                                                   \
    do {                                                                                       \
        switch (dtype) {                                                                       \
            case MVS_DTYPE_F32: GW_DISPATCH_C(FN, MVS_DTYPE_F32, __VA_ARGS__)                  \
            case MVS_DTYPE_BF16: GW_DISPATCH_C(FN, MVS_DTYPE_BF16, __VA_ARGS__)                \
            default: GW_DISPATCH_C(FN, MVS_DTYPE_F16, __VA_ARGS__)                             \
        }                                                                                      \
    } while (0)

int gw_launch_entropy(const void* feat, int dtype, int layout, const float* hom, const float* hyp, float* ent, int B, int V, int C, int D, int H, int W,
                      int vb, int ve, hipStream_t st) {
    GW_DISPATCH(gw_launch_entropy_t, feat, hom, hyp, ent, B, V, D, H, W, vb, ve, st);
}

int gw_launch_aggregate(const void* feat, int dtype, int layout, const float* hom, const float* hyp, const float* vis, float* vol, float* vis_sum,
                        int normalise, int B, int V, int C, int D, int H, int W, int vb, int ve, hipStream_t st) {
    GW_DISPATCH(gw_launch_aggregate_t, feat, hom, hyp, vis, vol, vis_sum, normalise, B, V, D, H, W, vb, ve, st);
}

}  // namespace mvs
