// Tile configurations shared by the fp32-exact (conv_kernels.hip) and split-bf16 (conv_bf16x3_kernels.hip) MFMA
// convolution kernels.
#pragma once
#include "mvs_common.h"

namespace mvs {

// ------------------------------------------------------------------------------------------------
// Conv3d (kd,3,3), padding (kd/2,1,1), stride (SD,SH,SW)                      module.py:89-126
// ------------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int KD_, int SD_, int SH_, int SW_, int TD_, int TH_, int CH_>
struct ConvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, KD = KD_, SD = SD_, SH = SH_, SW = SW_, TD = TD_, TH = TH_, CH = CH_;
    static constexpr int TW = 16;
    static constexpr int PD = KD / 2;
    static constexpr int ID = (TD - 1) * SD + KD, IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3;
    static constexpr int NVOX = ID * IH * IW;
    static constexpr int S = CH + 4;                 // padded voxel stride (floats)
    static constexpr int QC = CH / 4;                // channel quads per tap and pass
    static constexpr int NPASS = CIN / CH;
    static constexpr int NTAP = KD * 9;
    static constexpr int NSTEP = (NTAP * QC + 3) / 4;
    static constexpr int MREP = (COUT + 15) / 16;
    static constexpr int NB = TD * TH;
    static constexpr int NREP = NB / 4;
    static constexpr size_t LDS_BYTES = (size_t)NVOX * S * sizeof(float);
    static_assert(CIN % CH == 0 && CH % 4 == 0 && NB % 4 == 0, "bad conv tile configuration");
};

template <int CIN_, int COUT_, int SD_, int TDM_, int THM_>
struct DeconvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, SD = SD_, TDM = TDM_, THM = THM_;
    static constexpr int LD = (SD == 2) ? TDM + 1 : TDM + 2;
    static constexpr int ZO = (SD == 2) ? 0 : 1;          // LDS z index of tile-local m = 0
    static constexpr int LH = THM + 1, LW = 17;
    static constexpr int NVOX = LD * LH * LW;
    static constexpr int S = CIN + 4;
    static constexpr int QC = CIN / 4;
    static constexpr int NQ = CIN / 16;
    static constexpr int MREP = (COUT + 15) / 16;
    static constexpr int NB = TDM * THM;
    static constexpr int NREP = NB / 4;
    static constexpr size_t LDS_BYTES = (size_t)NVOX * S * sizeof(float);
    static_assert(CIN % 16 == 0 && NB % 4 == 0, "bad deconv tile configuration");
};


// One row per instantiated kernel: X(CIN, COUT, KD, SD, SH, SW, TD, TH, CH).
// stride-1: 4x4x16 outputs, 16-channel chunks (648-voxel halo tile, 51 KiB -> 3 blocks/CU); the 64->64 layer lives on
// the coarsest U-Net level (few voxels) and uses 2x4x16 tiles to keep all CUs busy; strided: 2x4x16 outputs, 8-channel
// chunks (57-71 KiB -> 2 blocks/CU), except the 8->16 (1,2,2) layer of stage 3/4 which is HBM-bound and runs faster with
// 2x2x16 tiles (32 KiB -> 5 blocks/CU: more loads in flight; measured 0.238 -> 0.195 ms, the other layers lose);
// 2-D (visibility CNN): 1x16x16 outputs.  8->16 stride 1 is CostRegNet's 3x3x3 `prob` head (one real output row of 16).  Deconvs 16->8 and 32->16 at (1,2,2): 2x4 input rows (0.376 -> 0.340 ms).
// The (2,2,2) layers only run on the small stage-1/2 volumes (tens of blocks on 256 CUs): 2x2x16 tiles double the
// number of blocks and halve each block's serial work (0.338 -> 0.257 ms for the six layers).
#ifndef MVS_T6464_TD
#define MVS_T6464_TD 2
#endif
#ifndef MVS_HEAD_TD
#define MVS_HEAD_TD 2
#define MVS_HEAD_TH 4
#endif
#ifndef MVS_T816_TD
#define MVS_T816_TD 4
#define MVS_T816_TH 2
#endif
#ifndef MVS_T1616_TD
#define MVS_T1616_TD 4
#endif
#ifndef MVS_T3232_TD
#define MVS_T3232_TD 4
#endif
#ifndef MVS_T1616_TH
#define MVS_T1616_TH 4
#endif
#ifndef MVS_T3232_TH
#define MVS_T3232_TH 4
#endif
#define MVS_CONV_TABLE(X)            \
    X(16, 16, 3, 1, 1, 1, MVS_T1616_TD, MVS_T1616_TH, 16)  \
    X(32, 32, 3, 1, 1, 1, MVS_T3232_TD, MVS_T3232_TH, 16)  \
    X(64, 64, 3, 1, 1, 1, MVS_T6464_TD, 4, 16)  \
    X(8, 16, 3, 1, 1, 1, MVS_HEAD_TD, MVS_HEAD_TH, 8)    \
    X(8, 16, 3, 2, 2, 2, 2, 2, 8)    \
    X(16, 32, 3, 2, 2, 2, 2, 2, 8)   \
    X(32, 64, 3, 2, 2, 2, 2, 2, 8)   \
    X(8, 16, 3, 1, 2, 2, MVS_T816_TD, MVS_T816_TH, 8)    \
    X(16, 32, 3, 1, 2, 2, 2, 4, 8)   \
    X(32, 64, 3, 1, 2, 2, 2, 4, 8)   \
    X(16, 16, 1, 1, 1, 1, 1, 16, 16) \
    X(16, 8, 1, 1, 1, 1, 1, 16, 16)

// X(CIN, COUT, SD, TDM, THM)
#ifndef MVS_D168_TDM
#define MVS_D168_TDM 4
#define MVS_D168_THM 2
#endif
#define MVS_DECONV_TABLE(X) \
    X(64, 32, 2, 2, 2)      \
    X(32, 16, 2, 2, 2)      \
    X(16, 8, 2, 2, 2)       \
    X(64, 32, 1, 2, 2)      \
    X(32, 16, 1, 2, 4)      \
    X(16, 8, 1, MVS_D168_TDM, MVS_D168_THM)

// precision of the MFMA contraction (C ABI: MVS_PREC_*)
// logits != NULL (persistent Cin = 8 kernels only): output channel 0 is written as a planar volume [B,OD,OH,OW] instead of y
// split != 0: x, y (and skip) are in the split activation format of MVS_PREC_BF16X3_SPLIT (conv_bf16x3_kernels.hip)
int conv3d_dispatch_bf16x3(const float* x, const void* wp, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W,
                           int kd, int sd, int sh, int sw, int relu, hipStream_t st, float* logits = nullptr, int split = 0);
int vis_weight_stream_bf16x3(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                             const float* b3, const float* w4, const float* b4, float* vis, int N, int H, int W, hipStream_t st, int f16 = 0);
// prob_w / prob_b / logits != NULL (Cout == 8 only): the 1x1x1 `prob` head is applied in the epilogue and the planar logits
// [B,OD,OH,OW] are written instead of y
int deconv3d_dispatch_bf16x3(const float* x, const void* wp, const float* bias, const float* skip, float* y, int B, int Cin, int Cout,
                             int D, int H, int W, int sd, hipStream_t st, const float* prob_w = nullptr, const float* prob_b = nullptr,
                             float* logits = nullptr, int relu = 1, int split = 0);

}  // namespace mvs
