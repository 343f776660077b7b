/* mvs_hip.h - C ABI of libmvs_hip.so: the MI355X (gfx950) implementation of MVSFormer++'s
 * depth-inference hot path (homography warp -> group-wise correlation cost volume ->
 * 3D-conv regularisation -> depth regression), SURVEY.md section 8.
 *
 * The reference (maybeLx/MVSFormerPlusPlus @ 2025-01-14) is pure Python/PyTorch and has no
 * FFI of its own; the seam a maintainer binds is the set of Python callables cited next to each
 * entry point below (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes
 * stub and the `patch_model()` swap.
 *
 * Calling convention
 *   - every pointer is a DEVICE pointer unless its name ends in _host
 *   - tensors are dense row-major with the shape given in the comment; B = batch
 *   - `stream` is a hipStream_t passed as void*; work is only enqueued, never synchronised
 *   - return value: MVS_OK (0) or an MVS_ERR_* code; mvs_last_error() gives the message
 *   - no hidden global state, no allocation: scratch memory is passed in by the caller
 *     (mvs_*_workspace_bytes tells how much)
 *
 * Internal activation layout ("channel-last"): [B, D, H, W, C] fp32, C in {8,16,32,64}.
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_ABI_VERSION 11

enum { MVS_OK = 0, MVS_ERR_ARG = 1, MVS_ERR_UNSUPPORTED = 2, MVS_ERR_LAUNCH = 3, MVS_ERR_WORKSPACE = 4 };
enum { MVS_DTYPE_F32 = 0, MVS_DTYPE_BF16 = 1, MVS_DTYPE_F16 = 2 };
/* feature-map layouts accepted by the two gather passes:
 *   MVS_LAYOUT_PLANAR       [B,V,C,H,W]          what the reference's FPN / FMT emit (models/module.py:257-270, models/FMT.py:195-197)
 *   MVS_LAYOUT_OCTET_TILED  [B,V,C/8,H,W,8]      the hand-off layout: the 8 channels of an octet are one
 *                                                contiguous 32-byte (fp32) / 16-byte (bf16, fp16) run; mvs_pack_features converts,
 *                                                a producer that writes it directly skips that pass              */
enum { MVS_LAYOUT_PLANAR = 0, MVS_LAYOUT_OCTET_TILED = 1 };
/* depth / confidence head modes, cost_volume.py:108-128 */
enum { MVS_HEAD_CE_EVAL = 0, MVS_HEAD_CE_TRAIN = 1, MVS_HEAD_REG = 2 };
/* regulariser kinds, cost_volume.py:41-49 */
enum { MVS_REG_COSTREGNET = 0, MVS_REG_COSTREGNET3D = 1 };
/* contraction precision of the MFMA convolutions:
 *   MVS_PREC_FP32    v_mfma_f32_16x16x4_f32, bit-exact fp32 fmaf chain (weights packed fp32)
 *   MVS_PREC_BF16X3  three-term split-bf16 product on v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32
 *                    accumulate, ~2^-16 relative product error; weights packed as hi/lo bf16)
 *   MVS_PREC_BF16P   mvs_tr_attention_fwd only: as BF16X3 (four-term scores, split v) but the softmax probabilities
 *                    enter p.v as one bf16 term (the reference's flash-attn path keeps q, k, v and p in bf16)
 *   MVS_PREC_BF16X3_SPLIT  the BF16X3 contraction with the activations IN HBM already split: every x_cl / skip_cl / y_cl / feat_cl /
 *                    volume_cl of the call is channel-last with, per voxel, C / 8 octets of [hi x8 | lo x8] bf16 (hi = bf16(x),
 *                    lo = bf16(x - hi); the same 4 bytes per element as fp32, element type still declared float).  The producing
 *                    epilogue splits once, the consumers' staging is a copy.  Logits stay planar fp32.  The inference U-Net runs in
 *                    this form (mvs_regnet_fwd / mvs_regnet_logits_fwd / mvs_conv3d_logits_fwd and the single layers accept it);
 *                    the training path and the fp32-contraction path keep fp32 activations.
 *   MVS_PREC_F16X2   fp16 ACTIVATIONS: every x_cl / skip_cl / y_cl / feat_cl / volume_cl of the call is channel-last _Float16 (declared
 *                    float*; 2 bytes per element), weights packed as fp16 hi + lo, two MFMA terms w_hi.x + w_lo.x on
 *                    v_mfma_f32_16x16x32_f16, fp32 accumulation, fp32 logits.  Final depth vs the fp32 oracle: 5.5e-5 relative L1 on plain
 *                    inputs, 4.2e-4 on the x30-logits stress set (bar 1e-3); the reference's own GPU path runs these layers under bf16
 *                    autocast (test.py:250).  Values beyond +-65504 overflow: the aggregate pass clamps the volume it writes.
 *   MVS_PREC_F16     as MVS_PREC_F16X2 with ONE fp16 term per weight (w_lo never read: half the MFMAs, half the weight bytes); the packed
 *                    weights are the MVS_PREC_F16X2 ones.  Depth error vs the fp32 oracle: 7e-5 plain / 4.8e-4 on the x30-logits stress set
 *                    (F16X2: 5.5e-5 / 4.2e-4; scripts/study_weight_precision.py).  mvs_conv3d_logits_fwd keeps both terms.
 *   MVS_PREC_F16MIX  one term on the layers with >= 32 channels on both sides or 64 on one (the U-Net's conv4 .. conv7), two on the others -
 *                    indistinguishable from MVS_PREC_F16X2 in the error study.  The format of the fine stages (ndepth <= model_th) under the
 *                    host mirror's default policy "stagemix" (round 5; its coarse stages run MVS_PREC_BF16X3_SPLIT), and of a bare regulariser.
 */
enum { MVS_PREC_FP32 = 0, MVS_PREC_BF16X3 = 1, MVS_PREC_BF16P = 2, MVS_PREC_BF16X3_SPLIT = 3, MVS_PREC_F16X2 = 4, MVS_PREC_ATTN16 = 5, MVS_PREC_F16 = 6, MVS_PREC_F16MIX = 7 };
/* format of the cost volume mvs_warp_corr_aggregate_fwd / mvs_volume_normalise leave behind: fp32 [B,D,H,W,8], the split activation
 * format of MVS_PREC_BF16X3_SPLIT, or fp16 [B,D,H,W,8] for MVS_PREC_F16X2 (aggregate only; normalised volumes of 8 groups only;
 * partial sums are always fp32) */
enum { MVS_VOLUME_F32 = 0, MVS_VOLUME_SPLIT = 1, MVS_VOLUME_F16 = 2 };
/* how the LDS-staged gather holds the SOURCE-view window: fp32 (exact; the fp32-equivalent formats), or one fp16 octet per position -
 * half the LDS reads; the source features are rounded to fp16 once (exact for fp16 / in-range bf16 features), everything else stays fp32.
 * MVS_VOLUME_F16 (aggregate) and mvs_warp_corr_entropy_keep_fwd imply MVS_GATHER_F16; kernels outside the LDS-staged form ignore it.
 * The reference rounds the features to a 16-bit type under its autocast (test.py:250, cost_volume.py:67).                              */
enum { MVS_GATHER_F32 = 0, MVS_GATHER_F16 = 1 };
/* format of the per-view group correlations mvs_warp_corr_entropy_keep_fwd keeps for mvs_corr_aggregate_fwd ([B,V-1,D,H,W,8]):
 * MVS_CORR_F16 = fp16 octets from fp16 source windows (16 B per voxel and view; the fp16 formats' fine stages), MVS_CORR_F32 = fp32 octets
 * from fp32 windows (32 B; EXACT: the streamed pass 2 then equals the second gather - the coarse stages of the default policy, whose
 * depth schedules the next stage's hypotheses: the fp16 correlations' 2^-11 is what an ill-conditioned cascade amplifies).               */
enum { MVS_CORR_F16 = 0, MVS_CORR_F32 = 1 };
/* epilogues of mvs_tr_linear_fwd */
enum { MVS_TR_EPI_BIAS = 0, MVS_TR_EPI_GELU = 1, MVS_TR_EPI_RES_LN = 2 };

int mvs_abi_version(void);

/* fp16 activation format (MVS_PREC_F16X2): every store of an activation tensor clamps to +-65504.  A clamp that really fired means the
 * default format degraded a value the fp32-equivalent format (MVS_PREC_BF16X3) would have kept.  Returns the number of work-items, summed
 * over all launches on the CURRENT device since the last reset, that stored at least one value beyond the fp16 range (cost volume writes
 * of mvs_warp_corr_aggregate_fwd / mvs_volume_to_f16, every convolution / transposed-convolution epilogue); reset != 0 clears it.
 * Synchronises the device (a device-to-host copy of three counters): call it between batches, not per launch.                        */
unsigned long long mvs_f16_saturation_count(int reset);
const char* mvs_last_error(void);

/* ---- a1 + warping.py:80-82 ---------------------------------------------------------------------
 * proj [B,V,2,4,4] (0 = extrinsic, 1 = intrinsic; datasets/general_eval.py:211-216).
 * For every source view v>=1: P = E.clone(); P[:3,:4] = K[:3,:3] @ E[:3,:4] (cost_volume.py:68-71),
 * M = P_v @ inverse(P_0); homography[b, v-1] = {M[:3,:3] row-major (9), M[:3,3] (3)}.          */
int mvs_compose_homography(const float* proj, int B, int V, float* homography /*[B,V-1,12]*/, void* stream);
/* Round 5, cascade prologue (DINOv2_mvsformer_model.py:127-150): the homographies of ALL stages (n_stages <= 8 proj tensors [B,V,2,4,4],
 * host array of device pointers) -> homography [n_stages,B,V-1,12], and - hyp != NULL - stage 1's hypotheses (mvs_init_range_fwd:
 * depth_values [B,N] -> hyp [B,D,H,W]) in ONE launch instead of n_stages + 1.                                                */
int mvs_cascade_prologue_fwd(const float* const* proj_host_ptrs, int n_stages, int B, int V, float* homography,
                             const float* depth_values, int N, int inverse, float* hyp, int D, int H, int W, void* stream);
/* same from already-composed 4x4 projections (the argument form of homo_warping_3D_with_mask) */
int mvs_homography_from_proj(const float* src_proj /*[B,4,4]*/, const float* ref_proj /*[B,4,4]*/, int B,
                             float* homography /*[B,12]*/, void* stream);

/* ---- a2/a3: models/warping.py:69-109 homo_warping_3D_with_mask ---------------------------------
 * src_fea [B,C,H,W] (dtype), depth [B,D] (depth_is_volume=0) or [B,D,H,W] (1)
 * -> warped [B,C,D,H,W] fp32, proj_mask [B,D,H,W] uint8 (either may be NULL).                   */
int mvs_homo_warp_fwd(const void* src_fea, int dtype, const float* homography /*[B,12]*/, const float* depth,
                      int depth_is_volume, float* warped, uint8_t* proj_mask, int B, int C, int D, int H, int W,
                      void* stream);

/* ---- a2-a5 fused: warp + group-wise correlation + softmax-entropy, cost_volume.py:65-92 --------
 * features [B,V,C,H,W] (dtype; view 0 = reference), hyp [B,D,H,W]
 * -> entropy [B,V-1,H,W].  No [C,D,H,W] or [G,D,H,W] intermediate is materialised.
 * Only source views in [view_begin, view_end) (1-based view indices) are processed.
 * gather_format: MVS_GATHER_F32 / MVS_GATHER_F16 (above).                                        */
int mvs_warp_corr_entropy_fwd(const void* features, int dtype, int layout, const float* homography /*[B,V-1,12]*/,
                              const float* hyp, float* entropy, int B, int V, int C, int G, int D, int H, int W,
                              int view_begin, int view_end, int gather_format, void* stream);

/* ---- pass 1 that KEEPS the per-view group correlations + the streaming pass 2 ------------------------------------------------
 * cost_volume.py:74-101 with the [B,G,D,H,W] per-view correlation the reference materialises kept as [B,V-1,D,H,W,8] in `corr_format`
 * (MVS_CORR_F16: clamped to the fp16 range, mvs_f16_saturation_count; MVS_CORR_F32: exact) instead of warping twice.
 * mvs_warp_corr_entropy_keep_fwd = the entropy pass over ALL source views + `corr`; mvs_corr_aggregate_fwd =
 * sum_v vis_v * corr_v / (sum_v vis_v + 1e-6) -> the normalised volume [B,D,H,W,8] in `volume_format` (MVS_VOLUME_F16 / _SPLIT / _F32:
 * the regulariser's format is independent of the gather's).  Built where mvs_gather_keeps_correlations() returns 1 (the LDS-staged
 * gather's shapes; MVS_CORR_F32 additionally needs D > 4); MVS_ERR_UNSUPPORTED elsewhere - the two-gather pair above covers every
 * shape.  Whether the stream pays (16 / 32 B per voxel and view written and read against a second gather) is the caller's policy.   */
int mvs_gather_keeps_correlations(int layout, int C, int G, int D, int H, int W);
int mvs_warp_corr_entropy_keep_fwd(const void* features, int dtype, int layout, const float* homography /*[B,V-1,12]*/,
                                   const float* hyp, float* entropy, void* corr, int corr_format, int B, int V, int C, int G, int D,
                                   int H, int W, void* stream);
int mvs_corr_aggregate_fwd(const void* corr, int corr_format, const float* vis /*[B,V-1,H,W]*/, void* volume_cl, int volume_format,
                           int B, int V, int D, int H, int W, void* stream);

/* ---- section 8f #4, producer side (round 5): the feature side's LAST 3x3 convolution emitting the hand-off layout ----------------
 * Replaces Conv2d(Cin, Cout, 3, padding=1[, bias]) [+ folded BatchNorm2d] [+ Swish] followed by torch.stack / mvs_pack_features:
 * models/FMT.py:195-197 (smooth_1/2/3: (32,32), (16,16), (8,8), no bias, act 0) and models/module.py:257-270 (FPNDecoder.out1/2/3:
 * (64,32), (64,16), (64,8), bias = folded BatchNorm shift, act 1 = Swish).  x [N,Cin,H,W] planar (in_dtype), image n at
 * x + n * in_batch_stride elements; w_packed = packing.pack_conv_weights_bf16x3(w[:, :, None], min(Cin, 32)) (split-bf16, three MFMA
 * terms: fp32-equivalent); bias [Cout] fp32 or NULL; tiled [.., Cout/8, H, W, 8] (out_dtype), image n at tiled + n * out_batch_stride
 * elements (so that view v of [B,V,C/8,H,W,8] can be written in place: out_batch_stride = V * Cout * H * W).  mvs_feature_conv_is_built
 * says which (Cin, Cout) pairs exist; others return MVS_ERR_UNSUPPORTED.                                                          */
int mvs_feature_conv_is_built(int Cin, int Cout);
int mvs_conv2d3x3_tiles_fwd(const void* x, int in_dtype, const void* w_packed, const float* bias, int act, void* tiled, int out_dtype,
                            int N, int Cin, int Cout, int H, int W, long long in_batch_stride, long long out_batch_stride,
                            void* stream);

/* ---- section 8f #4: feature hand-off -------------------------------------------------------------------
 * features [N,C,H,W] (dtype) -> tiled [N,C/8,H,W,8] (out_dtype); C % 8 == 0.  fp32 / bf16 / fp16 in, any of them out except
 * bf16 <-> fp16.  The cast to a 2-byte type rounds to nearest even, like the .to(bfloat16) the reference's autocast applies. */
int mvs_pack_features(const void* features, int dtype, void* tiled, int out_dtype, int N, int C, int H, int W, void* stream);

/* ---- a5: visibility CNN, cost_volume.py:36,93 + module.py:168-197 ------------------------------
 * entropy [N,H,W] -> vis [N,H,W] = sigmoid(conv1x1(CBR(16->8)(CBR(16->16)(CBR(1->16)(entropy))))).
 * Packed parameters are produced by the Python side (mvsformerplusplus_amd/packing.py):
 *   w1 [9][16] + b1[16] (BN folded), w2/w3 = MFMA-packed 3x3 weights (layout below), b2[16], b3[8]
 *   (padded to 16), w4[8], b4[1].  workspace >= mvs_vis_workspace_bytes(N,H,W).                  */
size_t mvs_vis_workspace_bytes(int N, int H, int W, int precision);
int mvs_vis_weight_fwd(const float* entropy, const float* w1, const float* b1, const void* w2, const float* b2,
                       const void* w3, const float* b3, const float* w4, const float* b4, float* vis,
                       void* workspace, size_t workspace_bytes, int N, int H, int W, int precision, void* stream);

/* the first (1->16 3x3 + BN + ReLU -> [N,H,W,16] channel-last) and last (8->1 1x1 + sigmoid) layer of the chain above
 * on their own; the two middle layers are mvs_conv3d_bn_relu_fwd with kd = 1 */
int mvs_vis_conv1_fwd(const float* entropy, const float* w1, const float* b1, float* out_cl16, int N, int H, int W, void* stream);
int mvs_vis_out_fwd(const float* x_cl8, const float* w4, const float* b4, float* vis, int N, int H, int W, void* stream);

/* ---- a4 + a6: recompute warp + correlation, weight by visibility, aggregate over views ----------
 * cost_volume.py:79-101.  vis [B,V-1,H,W].  volume_cl [B,D,H,W,G] channel-last.
 *   normalise = 1 : volume = sum_v ip_v*vis_v / (sum_v vis_v + 1e-6)              (single GPU)
 *   normalise = 0 : volume = partial sum over [view_begin, view_end), vis_sum [B,H,W] = partial
 *                   sum of vis (view-sharded multi-GPU: all-reduce both, then mvs_volume_normalise) */
int mvs_warp_corr_aggregate_fwd(const void* features, int dtype, int layout, const float* homography, const float* hyp,
                                const float* vis, float* volume_cl, float* vis_sum, int normalise, int volume_format, int B,
                                int V, int C, int G, int D, int H, int W, int view_begin, int view_end, void* stream);
/* ---- slab exchange of the view-sharded latency mode (no reference counterpart - the reference is
 * single-GPU; the sums follow cost_volume.py:97-101).  Message j = rows [row_begin[j], row_end[j]) of a PARTIAL fp32 volume
 * [B,D,H,W,G] followed by the same rows of the partial visibility sum [B,H,W], i.e. B*D*rows*W*G + B*rows*W floats (G: v10).
 * mvs_slab_pack: one launch writes the messages of all n_ranks destinations (send_host_ptrs[j] = device buffer, NULL = no message).
 * mvs_slab_reduce: slab_out = sum over ranks j = 0 .. n_ranks-1, in that order, of rank j's partial of MY rows [row_begin, row_end):
 * the own slice (j == my_rank) read in place from (volume_cl, vis_sum), the others from recv_host_ptrs[j] (NULL = rank j sent
 * nothing).  The pointer arrays are HOST arrays of device pointers (at most 16 ranks).                                           */
int mvs_slab_pack(const float* volume_cl, const float* vis_sum, float* const* send_host_ptrs, const int* row_begin, const int* row_end,
                  int n_ranks, int B, int D, int H, int W, int G, void* stream);
int mvs_slab_reduce(const float* volume_cl, const float* vis_sum, float* const* recv_host_ptrs, int n_ranks, int my_rank, float* slab_out,
                    int row_begin, int row_end, int B, int D, int H, int W, int G, void* stream);
/* volume_cl /= (vis_sum + 1e-6) in place; volume_format = MVS_VOLUME_SPLIT additionally converts it to the split format */
int mvs_volume_normalise(float* volume_cl, const float* vis_sum, int B, int D, int H, int W, int G, int volume_format, void* stream);
/* fp32 volume [B,D,H,W,8] (vis_sum != NULL: divided by vis_sum + 1e-6 first) -> fp16 [B,D,H,W,8] in a separate buffer, clamped to the
 * fp16 range: the cost volume of the MVS_PREC_F16X2 U-Net where the aggregate pass could not write it directly (view-sharded
 * multi-GPU partial sums; shapes outside the LDS-staged gather)                                                                */
/* 1 when the two gather passes take the LDS-staged kernels for this shape (the only ones that write MVS_VOLUME_SPLIT / _F16 directly) */
int mvs_gather_is_lds_staged(int layout, int C, int G, int D, int H, int W);
int mvs_volume_to_f16(const float* volume_cl, const float* vis_sum, void* out_f16, int B, int D, int H, int W, int G, void* stream);

/* ---- section 8f #2 (first slice): backward of mvs_warp_corr_aggregate_fwd(normalise = 1) ----------------------------
 * The gradient the reference's autograd produces for cost_volume.py:74-101: the sampling grid is built under torch.no_grad()
 * (warping.py:80) and the entropy from sim.detach() (cost_volume.py:90), so gradients reach the features and the visibility
 * maps only.  features planar [B,V,C,H,W] (fp32 / bf16 / fp16), volume_cl = the forward's output, vis_sum [B,H,W] = sum_v vis_v,
 * grad_volume_cl [B,D,H,W,G]; outputs: grad_features [B,V,C,H,W] fp32 (zeroed here, source views accumulated with atomics),
 * grad_vis [B,V-1,H,W].  Any C, G with C % G == 0.                                                                          */
int mvs_warp_corr_aggregate_bwd(const void* features, int dtype, const float* homography, const float* hyp, const float* vis,
                                const float* vis_sum, const float* volume_cl, const float* grad_volume_cl,
                                float* grad_features, float* grad_vis, int B, int V, int C, int G, int D, int H, int W,
                                void* stream);

/* ---- a7: Conv3d + folded BatchNorm3d + ReLU, module.py:89-126 ----------------------------------
 * x_cl [B,D,H,W,Cin] -> y_cl [B,OD,OH,OW,Cout]; kernel (kd,3,3), kd in {1,3}, padding (kd/2,1,1),
 * stride (sd,sh,sw) in {1,2}.  Implicit GEMM on MFMA; `precision` = MVS_PREC_* selects the contraction and the
 * packed-weight format (packing.pack_conv_weights / pack_conv_weights_bf16x3, DESIGN.md "MFMA weight packing");
 * bias [max(Cout,16)] fp32.                                                                                */
int mvs_conv3d_bn_relu_fwd(const float* x_cl, const void* w_packed, const float* bias, float* y_cl, int B, int Cin,
                           int Cout, int D, int H, int W, int kd, int sd, int sh, int sw, int relu, int precision,
                           void* stream);

/* ---- a10: CostRegNet's 3x3x3 `prob` head (Conv3d(8, 1, 3, padding 1, bias=False), module.py:391,407) on the MFMA path:
 * x_cl [B,D,H,W,8] -> logits [B,D,H,W] planar.  w_packed = the [1,8,3,3,3] weight zero-padded to 16 output rows and packed like
 * a Conv3d(8,16) (packing.pack_conv_weights_bf16x3); bias [16] (zeros for the reference's bias-free layer).  MVS_PREC_BF16X3 only;
 * the exact-fp32 form of the same head is mvs_prob_regress_fwd(prob_ksize = 3).                                               */
int mvs_conv3d_logits_fwd(const float* x_cl, const void* w_packed, const float* bias, float* logits, int B, int D, int H,
                          int W, int precision, void* stream);

/* ---- section 8f #2: training-mode regulariser (batch-statistics BatchNorm, weight gradients) -------------------------------
 * Channel-last fp32 [N voxels][C], C in {8,16,32,64}; `groups` independent statistics sets of N voxels each (tensor [groups][N][C],
 * statistics [groups][...]; the visibility CNN normalises each source view's batch on its own).  The forward convolutions and the DATA gradients of the training path are
 * mvs_conv3d_bn_relu_fwd (relu = 0, zero bias) / mvs_deconv3d_linear_fwd with un-folded, re-packed weights (training.py).
 *   mvs_bn_stats       sums[2C] (double) = per-channel [sum x | sum x^2]                       nn.BatchNorm3d, training=True
 *   mvs_bn_finalize    mean / biased var / 1/sqrt(var+eps) from sums and the voxel count (after an optional SyncBN all-reduce);
 *                      running_mean / running_var != NULL: nn.BatchNorm's momentum step (unbiased variance) in the same launch;
 *                      mvs_bn_running_update repeats that step alone
 *   mvs_bn_relu_apply  y = relu((z-mean)*invstd*gamma+beta) [+ skip]                            module.py:120-125, 402-405
 *   mvs_bn_relu_bwd    phase 0: sums[2C] = [d beta | d gamma] of dy through the ReLU mask; phase 1: dz (count = voxels of all
 *                      ranks, use_batch_stats = 0 for eval-mode BatchNorm inside a training graph)
 *   mvs_conv3d_wgrad   dW[CB][CA][kd*9] of Conv3d(k (kd,3,3), kd = 1 | 3, 'same' padding, stride) from input a_cl [B,D,H,W,CA] and output gradient
 *                      g_cl [B,OD,OH,OW,CB] on the fp32 MFMA path; a transposed convolution's weight gradient is the same call
 *                      with a = its output gradient and g = its input (result in ConvTranspose3d's [Cin][Cout][27] layout)    */
int mvs_deconv3d_linear_fwd(const float* x_cl, const void* w_packed, const float* bias, float* y_cl, int B, int Cin, int Cout,
                            int D, int H, int W, int sd, int precision, void* stream);
int mvs_bn_stats(const float* x_cl, double* sums, long long N, int C, int groups, void* stream);
int mvs_bn_finalize(const double* sums, double count, float eps, float* mean, float* var, float* invstd, float* running_mean,
                    float* running_var, float momentum, int C, int groups, void* stream);
int mvs_bn_running_update(const float* mean, const float* var, double count, float momentum, float* running_mean,
                          float* running_var, int C, int groups, void* stream);
int mvs_bn_relu_apply(const float* z_cl, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* skip_cl, float* y_cl, long long N, int C, int relu, int groups, void* stream);
int mvs_bn_relu_bwd(const float* dy_cl, const float* z_cl, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, double* sums, double count, float* dz_cl, long long N, int C, int relu,
                    int use_batch_stats, int phase, int groups, void* stream);
/* One conv / transposed-conv + batch-statistics BatchNorm + ReLU [+ skip] block per call (the launches of the granular entry points
 * chained in C): forward = pack, linear convolution -> z, statistics, finalize (+ running-statistics step), normalise -> y;
 * backward = running-statistics step (the reference's checkpoint recomputation), reduce -> dgamma / dbeta, dz, weight gradient dw
 * ([Cout][Cin][kd*9], ConvTranspose: [Cin][Cout][27]), data gradient da (NULL: not needed).  w = the layer's own fp32 weight;
 * zero_bias: >= 64 zeros; wpack_ws: bf16 scratch of mvs_pack_*_weights_elems elements (max over the block's three packings);
 * sums_ws: double [groups][2C]; B, D, H, W = the block INPUT's batch and size.  SyncBatchNorm / eval-mode BatchNorm: use the
 * granular entry points (an all-reduce sits between statistics and finalize).                                                  */
int mvs_train_block_fwd(const float* a_in_cl, const float* w, int transposed, int Cin, int Cout, int kd, int sd, int sh, int sw,
                        int B, int D, int H, int W, const float* gamma, const float* beta, float eps, float* running_mean,
                        float* running_var, float momentum, const float* skip_cl, const float* zero_bias, void* wpack_ws,
                        double* sums_ws, float* z_cl, float* mean, float* var, float* invstd, float* y_cl, int groups,
                        void* stream);
int mvs_train_block_bwd(const float* dy_cl, const float* a_in_cl, const float* z_cl, const float* mean, const float* var,
                        const float* invstd, const float* w, int transposed, int Cin, int Cout, int kd, int sd, int sh, int sw,
                        int B, int D, int H, int W, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, const float* zero_bias, void* wpack_ws, double* sums_ws,
                        float* dz_cl, float* dw, float* dgamma, float* dbeta, float* da_cl, int groups, void* stream);

/* MFMA weight packing on the device (bit-identical to packing.pack_conv_weights_bf16x3 / pack_deconv_weights_bf16x3; used by the
 * training path, which re-packs every un-folded weight each iteration).  *_elems = number of bf16 elements of the packed tensor
 * (-1: unsupported shape).  w: Conv3d [cout][cin][ntap] (tflip = 1: read as [cin][cout] with reversed taps = the data-gradient form
 * of a stride-1 convolution) / ConvTranspose3d [cin][cout][27].                                                               */
long long mvs_pack_conv_weights_elems(int cout, int cin, int ntap, int ch);
int mvs_pack_conv_weights(const float* w, void* packed_bf16, int cout, int cin, int ntap, int ch, int tflip, void* stream);
long long mvs_pack_deconv_weights_elems(int cin, int cout, int sd);
int mvs_pack_deconv_weights(const float* w, void* packed_bf16, int cin, int cout, int sd, void* stream);
int mvs_conv3d_wgrad(const float* a_cl, const float* g_cl, float* dw, int B, int CA, int CB, int D, int H, int W, int kd,
                     int sd, int sh, int sw, void* stream);

/* ---- a7: ConvTranspose3d(k3, padding 1, stride (sd,2,2), output_padding (sd-1,1,1)) + BN + ReLU,
 * then + skip (module.py:129-165, 402-405, 467-481, 498-501).
 * x_cl [B,D,H,W,Cin] -> y_cl [B,D*sd,2H,2W,Cout]; skip_cl has y's shape (NULL = no skip).          */
int mvs_deconv3d_bn_relu_add_fwd(const float* x_cl, const void* w_packed, const float* bias, const float* skip_cl,
                                 float* y_cl, int B, int Cin, int Cout, int D, int H, int W, int sd, int precision,
                                 void* stream);

/* ---- a7-a9, layer shapes OUTSIDE the tuned tables: shape-generic exact-fp32 Conv3d / ConvTranspose3d -------------------------------
 * Replaces nn.Conv3d / nn.ConvTranspose3d (+ folded BatchNorm3d, ReLU, skip add) of module.py:89-165 for any channel counts, kernel
 * size, stride and padding (dilation 1, groups 1): what a regulariser built with base_ch != 8 (cost_volume.py:29-49: CostRegNet(G, G),
 * widths 2G / 4G / 8G), the 1x1x1 `inner` convolution of in_channels != base_channels (module.py:385-388, 481-484) and a `prob` head of
 * G != 8 channels run on.  No shipped config reaches it: an FMA kernel built for coverage and exactness, not for the roofline.
 *   x_cl [B,D,H,W,Cin] fp32 -> y_cl [B,OD,OH,OW,Cout] fp32;  w_tck [kd*kh*kw][Cin][Cout] fp32 (BN folded; a ConvTranspose3d weight
 *   [Cin,Cout,kd,kh,kw] is laid out the same way, NOT flipped: the kernel gathers i = (o + p - k) / s);  bias [Cout] or NULL;
 *   skip_cl [B,OD,OH,OW,Cout] or NULL, added after bias and ReLU (module.py:403-405).
 *   transposed = 0: O = (N + 2p - k) / s + 1 per axis;  transposed = 1: O = (N - 1) s - 2p + k + output_padding, 0 <= output_padding < s
 *   (the caller states OD, OH, OW; anything else is MVS_ERR_ARG).  Cout == 1 writes what is also a planar [B,OD,OH,OW] volume.          */
int mvs_conv3d_generic_fwd(const float* x_cl, const float* w_tck, const float* bias, const float* skip_cl, float* y_cl, int B,
                           int Cin, int Cout, int D, int H, int W, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh,
                           int sw, int pd, int ph, int pw, int transposed, int relu, void* stream);
/* 1 when mvs_conv3d_bn_relu_fwd / mvs_deconv3d_bn_relu_add_fwd have a tuned MFMA kernel for the layer shape (kernel (kd,3,3), padding
 * (kd/2,1,1) / the k3 p1 (sd,2,2) transposed form), else 0: the host mirror routes the layer to mvs_conv3d_generic_fwd then */
int mvs_conv3d_is_tuned(int Cin, int Cout, int kd, int sd, int sh, int sw);
int mvs_deconv3d_is_tuned(int Cin, int Cout, int sd);

/* the last U-Net layer with the 1x1x1 `prob` head in its epilogue (Cout = 8): x_cl [B,D,H,W,Cin] -> logits [B,D*sd,2H,2W]
 * = prob(skip + relu(bn(deconv(x)))); MVS_PREC_BF16X3 only.  mvs_regnet_logits_fwd chains it after the other eight layers. */
int mvs_deconv3d_prob_fwd(const float* x_cl, const void* w_packed, const float* bias, const float* skip_cl, const float* prob_w,
                          const float* prob_b, float* logits, int B, int Cin, int D, int H, int W, int sd, int precision,
                          void* stream);

/* ---- a8/a9: whole regulariser U-Net (CostRegNet / CostRegNet3D), module.py:367-408 / 453-504 ----
 * volume_cl [B,D,H,W,8] -> feat_cl [B,D,H,W,8] = conv0 + relu(bn(deconv11(...))) (input of `prob`).
 * params: 9 packed weight pointers + 9 bias pointers in layer order conv1..conv6, conv7, conv9, conv11.*/
size_t mvs_regnet_workspace_bytes(int kind, int B, int D, int H, int W);
int mvs_regnet_fwd(int kind, const float* volume_cl, const void* const* w_packed, const float* const* bias,
                   float* feat_cl, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int precision,
                   void* stream);
/* the same U-Net with a 1x1x1 `prob` head (CostRegNet3D, module.py:486,502: prob_w [8], prob_b [1]) applied in the epilogue
 * of the last ConvTranspose3d: volume_cl -> logits [B,D,H,W] = prob_volume_pre; the 8-channel full-resolution feature
 * volume is never written (MVS_PREC_BF16X3 only).  Follow with mvs_softmax_regress_fwd.                              */
int mvs_regnet_logits_fwd(int kind, const float* volume_cl, const void* const* w_packed, const float* const* bias,
                          const float* prob_w, const float* prob_b, float* logits, void* workspace, size_t workspace_bytes,
                          int B, int D, int H, int W, int precision, void* stream);

/* ---- a8/a9 `prob` + a10 + a11: logits, softmax, depth regression, confidence --------------------
 * feat_cl [B,D,H,W,8]; prob_w: [8] (+ prob_b[1]) for the 1x1x1 head (CostRegNet3D, module.py:486)
 * or [27][8] tap-major for the 3x3x3 head without bias (CostRegNet, module.py:391; prob_b = NULL).
 * hyp [B,D,H,W].  mode: MVS_HEAD_*.  conf_n: window of conf_regression for MVS_HEAD_REG (0 = max prob).
 * Outputs (any of prob_volume / prob_volume_pre may be NULL): depth [B,H,W], conf [B,H,W],
 * prob_volume [B,D,H,W], prob_volume_pre [B,D,H,W].                                                */
int mvs_prob_regress_fwd(const float* feat_cl, const float* prob_w, const float* prob_b, int prob_ksize,
                         const float* hyp, float tmp, int mode, int conf_n, float* depth, float* conf,
                         float* prob_volume, float* prob_volume_pre, int B, int D, int H, int W, void* stream);
/* same head on precomputed logits [B,D,H,W] (depth_regression / conf_regression callers) */
int mvs_softmax_regress_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth,
                            float* conf, float* prob_volume, int B, int D, int H, int W, void* stream);
/* Round 5: a stage's head fused with the NEXT stage's inverse-depth schedule (mvs_softmax_regress_fwd + mvs_schedule_inverse_range_fwd with
 * shift = 0 in one launch; DINOv2_mvsformer_model.py:133-148 + cost_volume.py:105-131): besides depth / conf / prob_volume it writes
 * next_hyp [B,next_D,2H,2W] from this stage's depth and hypotheses (module.py:707-724, `ratio` = the next stage's depth_interals_ratio).
 * depth / conf / prob_volume are bit-identical to mvs_softmax_regress_fwd's; next_hyp equals the stand-alone schedule's up to FMA
 * contraction of the two translation units (<= 2e-6 of its range; bit-identical on the shipped build).                               */
int mvs_softmax_regress_schedule_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth, float* conf,
                                     float* prob_volume, float ratio, float* next_hyp, int next_D, int B, int D, int H, int W,
                                     void* stream);
/* Round 5: the LAST cascade stage's head with a16 fused (DINOv2_mvsformer_model.py:167-177): besides depth / conf / prob_volume it writes
 * conf_avg [B,H,W] = (sum_i nearest-upsampled prev_conf[i] + conf) / (n_prev + 1); prev_conf[i] is [B, H >> shift[i], W >> shift[i]]
 * (host arrays of n_prev <= 7 device pointers / shifts, earliest stage first: the summation order of mvs_confidence_average).    */
int mvs_softmax_regress_confavg_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth,
                                    float* conf, float* prob_volume, const float* const* prev_conf_host_ptrs,
                                    const int* prev_shifts_host, int n_prev, float* conf_avg, int B, int D, int H, int W,
                                    void* stream);

/* module.py:649-671 as free functions on a probability volume p [B,D,H,W]:
 * depth_regression: out = sum_d p*depth_values (depth_values [B,D,H,W]); conf_regression: window sum of n. */
int mvs_depth_regression_fwd(const float* p, const float* depth_values, float* out, int B, int D, int H, int W, void* stream);
int mvs_conf_regression_fwd(const float* p, int n, float* out, int B, int D, int H, int W, void* stream);

/* ---- a13-a15: hypothesis ranges, module.py:674-741 ----------------------------------------------*/
int mvs_init_range_fwd(const float* depth_values /*[B,N]*/, int N, int inverse, float* hyp /*[B,D,H,W]*/, int B,
                       int D, int H, int W, void* stream);
/* the per-pixel form of the two (module.py:683-688, 698-703): depth_values [B,H,W,N], every pixel's own first / last depth */
int mvs_init_range_pixel_fwd(const float* depth_values /*[B,H,W,N]*/, int N, int inverse, float* hyp /*[B,D,H,W]*/, int B,
                             int D, int H, int W, void* stream);
/* prev_depth [B,H/2,W/2], prev_hyp [B,Dprev,H/2,W/2] -> hyp [B,D,H,W]; ratio = depth_interals_ratio; shift != 0: the reference's
 * `shift=True` branch (module.py:712-715), statement by statement                                                             */
int mvs_schedule_inverse_range_fwd(const float* prev_depth, const float* prev_hyp, int Dprev, float ratio, int shift,
                                   float* hyp, int B, int D, int H, int W, void* stream);
/* interval [B] = depth_interals_ratio * depth_interval, or (interval_per_pixel != 0) [B,H/2,W/2] (module.py:727-741) */
int mvs_schedule_range_fwd(const float* prev_depth, const float* interval, int interval_per_pixel, float* hyp, int B, int D,
                           int H, int W, void* stream);

/* ---- a16: confidence fusion, DINOv2_mvsformer_model.py:167-177 -----------------------------------
 * out[b,y,x] = mean_s conf_s[b, y >> shift_s, x >> shift_s] (nearest upsample), n_stages <= 8.      */
int mvs_confidence_average(const float* const* conf_host_ptrs, const int* shifts_host, int n_stages, float* out,
                           int B, int H, int W, void* stream);

/* ==== section 8f #1: stage-1 transformer regulariser of the shipped config =========================
 * PureTransformerCostReg module.py:602-646 (FlashAttnBlock :535-583, FFN :507-532, LayerNorm3D :586-599), softmax
 * attention models/dino/layers/attention.py:76-101,141-170, Frustoconical PE models/position_encoding.py:138-189.
 * Tokens are rows [B, n, 64] fp32, n = (D/rd)(H/rh)(W/rw), token index (td*H/rh + th)*W/rw + tw (attention is
 * invariant to the order; the reference's "(h w d)" order, module.py:573, is not reproduced).  All contractions
 * are MVS_PREC_BF16X3 (fp32-equivalent); other precisions return MVS_ERR_UNSUPPORTED.  Packed weights come from
 * packing.pack_linear_bf16x3 (layout in mvsformerplusplus_amd/packing.py).                                        */

/* get_position_3d(normalize=True), position_encoding.py:138-163.  K [B,3,3] = proj[:,0,1,:3,:3] of the stage,
 * hyp [B,D,H,W], depth_values [B*n] (only its min / max are used).  range [6] = {height_min, height_max, width_min,
 * width_max, depth_min, depth_max}: with compute_range != 0 the first four are measured over the whole volume
 * (stage 1) and written, otherwise they are read (later stages reuse stage 1's, DINOv2_mvsformer_model.py:152-160);
 * the last two are always written.  -> position3d [B,3,D,H,W] in 0..1.                                            */
size_t mvs_position3d_workspace_bytes(void);
int mvs_position3d_fwd(const float* K, const float* hyp, const float* depth_values, int n_depth_values, float* range,
                       int compute_range, float* workspace, size_t workspace_bytes, float* position3d, int B, int D,
                       int H, int W, void* stream);

/* get_position_3d(normalize=False), position_encoding.py:146-149: position3d [B,3,D,H,W] = K^-1 [x, y, 1] * depth, no ranges (v10) */
int mvs_position3d_raw_fwd(const float* K, const float* hyp, float* position3d, int B, int D, int H, int W, void* stream);
/* PositionEncoding3D(position3d, C, rescale) as a tensor of its own, position_encoding.py:164-189: position3d [B,3,N] (N = D*H*W) ->
 * pe [B,3C,N], channel ax*C + 2f = sin(pos_ax * rescale * div_f), + 1 = cos; div_term [C/2] (device) = the frequencies
 * exp(2f * (-ln 1e4 / C)) as the caller computed them; C even.  The hot path never materialises the encoding (mvs_tr_embed_fwd
 * evaluates it per token); this is the standalone form for callers of the function (v10)                                        */
int mvs_position_encoding3d_fwd(const float* position3d, const float* div_term, float* pe, int B, int C, float rescale, long long N,
                                void* stream);

/* x + pe_proj(PositionEncoding3D(position3d, 8)) (module.py:631-635, position_encoding.py:166-189; skipped when
 * position3d == NULL), `down` = Conv3d(8, 64, kernel = stride = (rd,rh,rw)) + bias + LayerNorm3D(64, eps 1e-6).
 * volume_cl [B,D,H,W,8], pe_w = pe_proj.weight [8][24], pe_div_host = the 4 frequencies exp(2k * -ln(1e4)/8) (HOST
 * pointer), w_packed = pack_linear_bf16x3(down.0.weight as [64][patch_voxel*8 + c]) -> tokens [B,n,64].           */
int mvs_tr_embed_fwd(const float* volume_cl, const float* position3d, const float* pe_w, const float* pe_div_host,
                     const void* w_packed, const float* bias, const float* ln_w, const float* ln_b, float* tokens,
                     int B, int D, int H, int W, int rd, int rh, int rw, int precision, void* stream);

/* y = epilogue(x @ W^T): x [B,n,K], W [N,K] packed, y [B,n,N].
 *   MVS_TR_EPI_BIAS    y = x W^T (+ bias)                                                K = 64
 *   MVS_TR_EPI_GELU    y = gelu(x W^T + bias), exact erf form (FFN.linear1 + act)         K = 64
 *   MVS_TR_EPI_RES_LN  y = LayerNorm(residual + gamma[0] * (x W^T + bias)), N = 64        K = 64 | 256
 *                      (attn.proj / ffn.linear2 + layer scale + post-norm, module.py:575-576)                      */
int mvs_tr_linear_fwd(const float* x, const void* w_packed, const float* bias, int epilogue, const float* residual,
                      const float* gamma, const float* ln_w, const float* ln_b, float ln_eps, float* y, int B, int n,
                      int K, int N, int precision, void* stream);

/* attn.qkv (no bias) written as the operands of the attention kernel, q pre-scaled by softmax_scale * log2 e.  softmax_scale =
 * head_dim^-0.5 * log(n) / log(train_avg_length) for "entropy_invariance" (attention.py:158-161).  heads = 4, head_dim = 16.
 * `precision` = contraction of the projection itself (MVS_PREC_BF16X3); `operand_format` = what the attention kernel will read:
 *   MVS_PREC_BF16X3 (or _BF16P)  q, k as [B,heads,npad,32] bf16 = [hi16 | lo16], v transposed as [B,heads,2,16,npad] bf16, npad = n
 *                                rounded up to 64 (rounds 1-3: fp32-equivalent attention)
 *   MVS_PREC_ATTN16              ONE 16-bit term per operand like the reference's flash-attn (dino/layers/attention.py:141-170 runs q, k, v,
 *                                p in bf16): q, k fp16 (clamped to +-65504), v bf16 (the probabilities are bf16 in the kernel); q
 *                                [B,heads,npad,16]; k as score-MFMA operand tiles [B,heads,npad/32,4,16,2,4]; v as p.v-MFMA operand tiles
 *                                [B,heads,npad/32,4,16,8] in the key order the scores leave the matrix core in
 *                                (csrc/attention_f16_kernels.hip); npad = n rounded up to 256
 * Each buffer holds mvs_tr_attention_operand_bytes(B, n, heads) bytes (enough for either format).                              */
size_t mvs_tr_attention_operand_bytes(int B, int n, int heads);
int mvs_tr_qkv_fwd(const float* x, const void* w_packed, void* q, void* k, void* vt, float softmax_scale, int B, int n,
                   int heads, int precision, int operand_format, void* stream);
/* softmax(q k^T) v over all n tokens -> out [B,n,heads*16] (scaled_dot_product_attention, attention.py:96);
 * precision = the operand format of mvs_tr_qkv_fwd: MVS_PREC_BF16X3 (four-term scores, three-term p.v), MVS_PREC_BF16P (the same with
 * one-term probabilities) or MVS_PREC_ATTN16 (fp16 q / k, bf16 p / v, one MFMA term, fp32 softmax statistics and accumulation)              */
int mvs_tr_attention_fwd(const void* q, const void* k, const void* vt, float* out, int B, int n, int heads,
                         int precision, void* stream);
/* Backward of the attention core (training path; the reference differentiates scaled_dot_product_attention / flash-attn,
 * dino/layers/attention.py:141-170): qkv [B,n,3,heads,16] fp32 = the projection's plain output (q | k | v), o [B,n,heads*16] the
 * forward result, d_o its gradient -> d_qkv [B,n,3,heads,16] (dq | dk | dv).  fp32 throughout, two launches, no atomics.
 * lse_ws / dsum_ws: B*heads*n floats each (log-sum-exp and sum_c d_o*o per query row, produced by the first launch).            */
int mvs_tr_attention_bwd(const float* qkv, const float* o, const float* d_o, float* d_qkv, float* lse_ws, float* dsum_ws,
                         int B, int n, int heads, float softmax_scale, void* stream);

/* `up` = ConvTranspose3d(64, 8, kernel = stride = (rd,rh,rw)) + bias + LayerNorm3D(8, eps 1e-6), then `prob` =
 * Conv3d(8, 1, 1) + bias (module.py:621-626, 643-644): tokens [B,n,64] -> logits [B,D,H,W].
 * w_packed = pack_linear_bf16x3(up.0.weight as [patch_voxel*8 + co][ci]).                                          */
int mvs_tr_up_prob_fwd(const float* tokens, const void* w_packed, const float* up_bias, const float* ln_w,
                       const float* ln_b, const float* prob_w, const float* prob_b, float* logits, int B, int D, int H,
                       int W, int rd, int rh, int rw, int precision, void* stream);

/* ==== section 8f #3: depth-map filtering after inference ============================================
 * misc/fusion.py (reprojection-consistency filters) as driven by test.py:388-409 ("pcd") and test.py:455-483 ("dpcd").
 * Cameras: [N,2,4,4] as everywhere (0 = extrinsic world->camera, 1 = intrinsic in the top-left 3x3), first packed once per
 * camera into mvs_fusion_campack_floats() floats {K, K^-1, E, E^-1} (the reference inverts them per call, fusion.py:24,32).
 * Depth / confidence maps are planar fp32 [n,h,w] / [n,v,h,w]; pixel centres sit at +0.5 (fusion.py:9-10).              */
size_t mvs_fusion_campack_floats(void);
int mvs_fusion_prepare_cams(const float* cams /*[N,2,4,4]*/, int N, float* packed /*[N,campack]*/, void* stream);
/* dynamic = 0: get_reproj (fusion.py:80-97) + vis_filter (:100-109) + ave_fusion (:112-114); p0 = img_dist_thresh,
 *              p1 = depth_thresh, vthresh = view threshold; srcs_conf (nullable) zeroes source depths with conf <=
 *              conf_thresh (test.py:389-392); vis_masks [n,v,h,w].
 * dynamic = 1: get_reproj_dynamic (:116-153) + vis_filter_dynamic (:156-168) + test.py:463-476; p0 = dist_base,
 *              p1 = rel_diff_base; vis_masks [n,v,v-1,h,w]; v >= 2.
 * Either computes the reprojection from depths + cameras (xyd_in = NULL; written to xyd_out [n,v,3,h,w] / in_range_out
 * [n,v,h,w] when non-NULL) or starts from a given one (xyd_in, in_range_in).  With depth != NULL the filter runs:
 * depth [n,h,w] = averaged depth, geo_mask / mask [n,h,w] (mask = geo & (ref_conf > conf_thresh), ref_conf nullable),
 * points [n,3,h,w] = world coordinates of the averaged depth (test.py:406-409); any of vis_masks / geo_mask / mask /
 * points may be NULL.  v <= 16.                                                                                          */
int mvs_fusion_filter_fwd(int dynamic, const float* ref_depth, const float* ref_conf, const float* srcs_depth,
                          const float* srcs_conf, const float* ref_cam_packed, const float* srcs_cam_packed,
                          const float* xyd_in, const float* in_range_in, float conf_thresh, float p0, float p1, float vthresh,
                          float* xyd_out, float* in_range_out, uint8_t* vis_masks, float* depth, uint8_t* geo_mask,
                          uint8_t* mask, float* points, int n, int v, int h, int w, void* stream);
/* ave_fusion (fusion.py:112-114) on its own: masks [n,v,h,w] fp32 */
int mvs_fusion_ave_fwd(const float* ref_depth, const float* reproj_xyd, const float* masks, float* out, int n, int v, int h,
                       int w, void* stream);

/* ---- layout helpers for the nn.Module-level API (NCDHW <-> channel-last) -------------------------*/
int mvs_ncdhw_to_cl(const float* x, float* y_cl, int B, int C, int D, int H, int W, void* stream);
int mvs_cl_to_ncdhw(const float* x_cl, float* y, int B, int C, int D, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
