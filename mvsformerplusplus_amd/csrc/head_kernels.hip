// Depth head, hypothesis scheduling and small layout helpers.
//
//   prob_regress     `prob` conv of the regulariser (1x1x1 + bias for CostRegNet3D module.py:486, 3x3x3 no bias for
//                    CostRegNet module.py:391) fused with softmax over depth, temperature depth regression /
//                    argmax / plain regression, and the confidence (cost_volume.py:105-128, module.py:649-671).
//                    One work-item per pixel keeps the D logits in registers: the [D,H,W] logits make no HBM round
//                    trip unless the caller asks for prob_volume / prob_volume_pre (training losses only).
//   init_range / schedule_inverse_range / schedule_range   module.py:674-741; the 2x trilinear upsample
//                    (align_corners=True, D unchanged => bilinear in H,W) is fused with the per-pixel linspace and
//                    the reciprocal, so the low-resolution hypothesis tensor is never materialised.
//   confidence_average   DINOv2_mvsformer_model.py:167-177 (nearest upsample + mean over stages).
// All of these are HBM-bound streaming kernels: bytes = inputs read once + outputs written once.
#include "mvs_common.h"

namespace mvs {

// softmax over depth + depth regression / argmax / plain regression + confidence of ONE pixel (cost_volume.py:105-128, module.py:649-671).
// l: the D logits in registers (DC > 0) or, DC == 0, read from lg[d * HW]; hp / pv: this pixel's hypothesis / probability columns
// (stride HW; pv may be null).  Shared by prob_regress_kernel and the schedule-fused head (round 5) so that both give the same bits.
template <int DC>
__device__ __forceinline__ void head_from_logits(const float* l, const float* lg, const float* hp, size_t HW, int D, float tmp, int mode, int conf_n,
                                                 float* pv, float& depth, float& conf) {
#define HL(d) (DC > 0 ? l[(DC > 0 ? (d) : 0)] : lg[(size_t)(d) * HW])
    // ---- softmax over depth (cost_volume.py:106) ----
    float m = -INFINITY;
#pragma unroll
    for (int d = 0; d < D; ++d) m = fmaxf(m, HL(d));
    float den = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) den += expf(HL(d) - m);

    depth = 0.0f;
    conf = 0.0f;
    if (mode == MVS_HEAD_CE_EVAL) {
        // depth_regression(softmax(pre * tmp), depth_values)  cost_volume.py:115
        float m2 = -INFINITY;
#pragma unroll
        for (int d = 0; d < D; ++d) m2 = fmaxf(m2, HL(d) * tmp);
        float den2 = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) den2 += expf(HL(d) * tmp - m2);
#pragma unroll
        for (int d = 0; d < D; ++d) depth += (expf(HL(d) * tmp - m2) / den2) * hp[(size_t)d * HW];
    }
    float best = -1.0f, idxf = 0.0f;
    int best_i = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pr = expf(HL(d) - m) / den;
        if (pv) pv[(size_t)d * HW] = pr;
        if (pr > best) { best = pr; best_i = d; }                 // first maximum, like torch.max
        if (mode == MVS_HEAD_REG) { depth += pr * hp[(size_t)d * HW]; idxf += pr * (float)d; }
    }
    conf = best;                                                  // prob_volume.max(1)[0]  cost_volume.py:117
    if (mode == MVS_HEAD_CE_TRAIN) depth = hp[(size_t)best_i * HW];   // cost_volume.py:109-112
    if (mode == MVS_HEAD_REG && conf_n > 0) {
        // conf_regression module.py:658-671: window sum of n probabilities around floor(sum p*idx)
        int idx = (int)idxf;
        idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
        const int lo = (conf_n & 1) ? idx - conf_n / 2 : idx - (conf_n / 2 - 1);
        const int hi = idx + conf_n / 2;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d >= lo && d <= hi) s += expf(HL(d) - m) / den;
        conf = s;
    }
#undef HL
}

// a16: mean over stages of nearest-upsampled confidences (DINOv2_mvsformer_model.py:167-177).  As a stand-alone kernel
// (confidence_average_kernel) and - round 5 - as the epilogue of the LAST stage's head (prob_regress_kernel: cavg.n = number of
// EARLIER stages, this stage's own confidence is the value the work-item just computed; one launch and one HW read less).
struct ConfPtrs {
    const float* p[8];
    int shift[8];
    int n;
};

// ------------------------------------------------------------------------------------------------
// logits -> softmax -> depth / confidence
//   DC  compile-time D (registers) or 0 (logits live in the prob_volume_pre buffer)
//   KS  1: 1x1x1 head (+bias) on channel-last features, 3: 3x3x3 head, 0: logits are given
// ------------------------------------------------------------------------------------------------
template <int DC, int KS>
__global__ __launch_bounds__(256) void prob_regress_kernel(const float* __restrict__ in, const float* __restrict__ prob_w,
                                                           const float* __restrict__ prob_b, const float* __restrict__ hyp, float tmp,
                                                           int mode, int conf_n, float* __restrict__ depth_out,
                                                           float* __restrict__ conf_out, float* __restrict__ prob_vol,
                                                           float* __restrict__ pre, int D_, int H, int W, ConfPtrs cavg = ConfPtrs{},
                                                           float* __restrict__ conf_avg_out = nullptr) {
    const int D = DC > 0 ? DC : D_;
    const int HW = H * W;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float l[DC > 0 ? DC : 1];
    float* prep = pre ? pre + (size_t)b * D * HW + p : nullptr;
    const float* hp = hyp + (size_t)b * D * HW + p;
#define MVS_LOGIT(d) (DC > 0 ? l[(DC > 0 ? (d) : 0)] : prep[(size_t)(d) * HW])

    // ---- logits ----
    if (KS == 3 && DC > 0) {
        // 3x3x3 head with the logits of the whole depth column in registers: every (kh, kw) neighbour column is read
        // ONCE (2 float4 per plane) and plane zz feeds the three outputs zz+1, zz, zz-1 (kd = 0, 1, 2): 9*D loads of
        // 32 bytes per pixel instead of 27*D.
#pragma unroll
        for (int d = 0; d < (DC > 0 ? DC : 1); ++d) l[d] = 0.0f;
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = y + kh - 1;
            if (yy < 0 || yy >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = x + kw - 1;
                if (xx < 0 || xx >= W) continue;
                const float* w0 = prob_w + ((0 * 3 + kh) * 3 + kw) * 8;      // kd = 0 -> output zz + 1
                const float* w1 = prob_w + ((1 * 3 + kh) * 3 + kw) * 8;      // kd = 1 -> output zz
                const float* w2 = prob_w + ((2 * 3 + kh) * 3 + kw) * 8;      // kd = 2 -> output zz - 1
                const float* col = in + (((size_t)b * D) * HW + (size_t)yy * W + xx) * 8;
#pragma unroll
                for (int zz = 0; zz < (DC > 0 ? DC : 1); ++zz) {
                    const float4* f = reinterpret_cast<const float4*>(col + (size_t)zz * HW * 8);
                    const float4 a = f[0], c = f[1];
                    const float xv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
                    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) { s0 += xv[ch] * w0[ch]; s1 += xv[ch] * w1[ch]; s2 += xv[ch] * w2[ch]; }
                    if (zz + 1 < DC) l[(zz + 1 < DC) ? zz + 1 : 0] += s0;
                    l[zz] += s1;
                    if (zz >= 1) l[(zz >= 1) ? zz - 1 : 0] += s2;
                }
            }
        }
        if (prep) {
#pragma unroll
            for (int d = 0; d < (DC > 0 ? DC : 1); ++d) prep[(size_t)d * HW] = l[d];
        }
    } else {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float v;
        if (KS == 0) {
            v = in[((size_t)b * D + d) * HW + p];
        } else if (KS == 1) {
            const float4* f = reinterpret_cast<const float4*>(in + (((size_t)b * D + d) * HW + p) * 8);
            const float4 a = f[0], c = f[1];
            v = a.x * prob_w[0];
            v += a.y * prob_w[1]; v += a.z * prob_w[2]; v += a.w * prob_w[3];
            v += c.x * prob_w[4]; v += c.y * prob_w[5]; v += c.z * prob_w[6]; v += c.w * prob_w[7];
            v += prob_b[0];
        } else {
            v = 0.0f;
            for (int kd = 0; kd < 3; ++kd) {
                const int zz = d + kd - 1;
                if (zz < 0 || zz >= D) continue;
                for (int kh = 0; kh < 3; ++kh) {
                    const int yy = y + kh - 1;
                    if (yy < 0 || yy >= H) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        const int xx = x + kw - 1;
                        if (xx < 0 || xx >= W) continue;
                        const float4* f = reinterpret_cast<const float4*>(in + (((size_t)b * D + zz) * HW + (size_t)yy * W + xx) * 8);
                        const float* w = prob_w + ((kd * 3 + kh) * 3 + kw) * 8;
                        const float4 a = f[0], c = f[1];
                        v += a.x * w[0]; v += a.y * w[1]; v += a.z * w[2]; v += a.w * w[3];
                        v += c.x * w[4]; v += c.y * w[5]; v += c.z * w[6]; v += c.w * w[7];
                    }
                }
            }
        }
        if (DC > 0) l[DC > 0 ? d : 0] = v;
        if (prep && KS != 0) prep[(size_t)d * HW] = v;
    }
    }

    float depth, conf;
    head_from_logits<DC>(l, prep, hp, (size_t)HW, D, tmp, mode, conf_n, prob_vol ? prob_vol + (size_t)b * D * HW + p : nullptr, depth, conf);
    depth_out[(size_t)b * HW + p] = depth;
    conf_out[(size_t)b * HW + p] = conf;
    if (conf_avg_out != nullptr) {                                // a16 fused: (sum of the earlier stages' nearest-upsampled confidences + this one) / n
        float s = 0.0f;
        for (int i = 0; i < cavg.n; ++i) {
            const int hs = H >> cavg.shift[i], ws = W >> cavg.shift[i];
            s += cavg.p[i][(size_t)b * hs * ws + (size_t)(y >> cavg.shift[i]) * ws + (x >> cavg.shift[i])];
        }
        conf_avg_out[(size_t)b * HW + p] = (s + conf) / (float)(cavg.n + 1);      // same summation order as confidence_average_kernel: stage 1 first
    }
#undef MVS_LOGIT
}

// free-function forms of module.py:649-671 on an existing probability volume
__global__ void depth_regression_kernel(const float* __restrict__ p, const float* __restrict__ dv, float* __restrict__ out, int D, int HW) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (i >= HW) return;
    float s = 0.0f;
    for (int d = 0; d < D; ++d) s += p[((size_t)b * D + d) * HW + i] * dv[((size_t)b * D + d) * HW + i];
    out[(size_t)b * HW + i] = s;
}

__global__ void conf_regression_kernel(const float* __restrict__ p, int n, float* __restrict__ out, int D, int HW) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (i >= HW) return;
    const float* pp = p + (size_t)b * D * HW + i;
    float idxf = 0.0f;
    for (int d = 0; d < D; ++d) idxf += pp[(size_t)d * HW] * (float)d;
    int idx = (int)idxf;
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    const int lo = (n & 1) ? idx - n / 2 : idx - (n / 2 - 1);
    const int hi = idx + n / 2;
    float s = 0.0f;
    for (int d = (lo < 0 ? 0 : lo); d <= hi && d < D; ++d) s += pp[(size_t)d * HW];
    out[(size_t)b * HW + i] = s;
}

// ------------------------------------------------------------------------------------------------
// a13 / init_range: [B,N] depth values -> [B,D,H,W] hypotheses              module.py:674-704
// ------------------------------------------------------------------------------------------------
__global__ void init_range_kernel(const float* __restrict__ dv, int N, int inverse, float* __restrict__ hyp, int D, int HW) {
    const int b = (int)blockIdx.z, d = (int)blockIdx.y;
    const float v = init_range_value(dv[(size_t)b * N], dv[(size_t)b * N + N - 1], inverse, d, D);
    float* o = hyp + ((size_t)b * D + d) * HW;
    for (int p = (int)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (int)gridDim.x * blockDim.x) o[p] = v;
}

// the per-pixel form (module.py:683-688, 698-703: cur_depth [B,H,W,N] - every pixel carries its own first / last depth)
__global__ void init_range_pixel_kernel(const float* __restrict__ dv, int N, int inverse, float* __restrict__ hyp, int D, int HW) {
    const int b = (int)blockIdx.z, d = (int)blockIdx.y;
    float* o = hyp + ((size_t)b * D + d) * HW;
    for (int p = (int)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (int)gridDim.x * blockDim.x) {
        const float* px = dv + ((size_t)b * HW + p) * N;
        o[p] = init_range_value(px[0], px[N - 1], inverse, d, D);
    }
}

// bilinear source coordinates of F.interpolate(..., align_corners=True) along one axis
__device__ __forceinline__ void lin_coord(int dst, int in_size, int out_size, int* i0, int* i1, float* l0, float* l1) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    const float src = scale * (float)dst;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    float lam = src - (float)a;
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    *i0 = a;
    *i1 = a + 1 < in_size ? a + 1 : in_size - 1;
    *l1 = lam;
    *l0 = 1.0f - lam;
}

// a14 (mode 0, inverse) / a15 (mode 1, linear): prev stage [h,w] -> [D,H,W] with H = 2h, W = 2w (any ratio works)
__global__ __launch_bounds__(256) void schedule_range_kernel(const float* __restrict__ prev_depth, const float* __restrict__ prev_hyp,
                                                             int Dprev, float ratio, const float* __restrict__ interval, int linear,
                                                             float* __restrict__ hyp, int D, int H, int W, int h, int w, int variant) {
    const int HW = H * W, hw = h * w;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_coord(y, h, H, &y0, &y1, &ly0, &ly1);
    lin_coord(x, w, W, &x0, &x1, &lx0, &lx1);
    const int cy[4] = {y0, y0, y1, y1}, cx[4] = {x0, x1, x0, x1};
    float lo[4], span[4];    // per corner: value at d = 0 and (value at d = D-1) - (value at d = 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = cy[k] * w + cx[k];
        const float dep = prev_depth[(size_t)b * hw + q];
        if (!linear) {
            const float h1 = prev_hyp[((size_t)b * Dprev + 1) * hw + q];
            const float h2 = prev_hyp[((size_t)b * Dprev + 2) * hw + q];
            const float last_itv = 1.0f / h2 - 1.0f / h1;                           // module.py:708
            float inv_min = 1.0f / dep + ratio * last_itv;
            float inv_max = 1.0f / dep - ratio * last_itv;
            if (variant) {
                // shift = True (module.py:712-715), statement by statement: the second line reads the ALREADY shifted inverse_max_depth,
                // so inverse_min_depth moves by the rounding residue of the first line only - the reference's behaviour, kept
                const float is_neg = inv_max < 0.002f ? 1.0f : 0.0f;
                inv_max = inv_max - (inv_max - 0.002f) * is_neg;
                inv_min = inv_min - (inv_max - 0.002f) * is_neg;
            }
            lo[k] = inv_max;
            span[k] = inv_min - inv_max;
        } else {
            const float itv = variant ? interval[(size_t)b * hw + q] : interval[b];          // per-pixel [B,h,w] intervals (module.py:731-732)
            float dmin = dep - (float)D / 2.0f * itv;                               // module.py:733-736
            dmin = dmin < 0.001f ? 0.001f : dmin;
            const float dmax = dep + (float)D / 2.0f * itv;
            lo[k] = dmin;
            span[k] = (dmax - dmin) / (float)(D - 1);
        }
    }
    for (int d = 0; d < D; ++d) {
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            c[k] = linear ? lo[k] + (float)d * span[k] : lo[k] + span[k] * ((float)d / (float)(D - 1));
        const float v = ly0 * (lx0 * c[0] + lx1 * c[1]) + ly1 * (lx0 * c[2] + lx1 * c[3]);
        hyp[((size_t)b * D + d) * HW + p] = linear ? v : 1.0f / v;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5: a stage's head FUSED with the next stage's inverse-depth schedule (a10-a12 + a14 in one launch instead of two).
// The schedule is a 2x bilinear (align_corners=True) upsample of per-pixel (1/depth -/+ ratio * itv) followed by a linspace and a
// reciprocal (module.py:707-724): an output pixel (Y, X) reads the source pixels floor(Y r), floor(Y r) + 1 with r = (H-1)/(2H-1) < 1/2,
// so the 2 SH x 2 SW outputs [2 y0, 2 y0 + 2 SH) x [2 x0, 2 x0 + 2 SW) of the source tile [y0, y0 + SH) x [x0, x0 + SW) need the source rows
// y0 - 1 .. y0 + SH and columns x0 - 1 .. x0 + SW: a one-pixel ring.  A workgroup therefore runs the head for its 8 x 32 tile (writing
// depth / confidence / probabilities for it) AND for the ring (84 pixels, nothing written: the neighbour tiles own them and compute the
// same bits), keeps (inv_max, inv_min - inv_max) of the 10 x 34 region in LDS, and writes the next stage's hypotheses of its 16 x 64
// outputs - the arithmetic of schedule_range_kernel (variant 0), statement for statement.  33 % more head work (logits and hypotheses of
// the ring come from L2) against a launch and a [B,H,W] + 2 [B,H,W] re-read less.  KS = 0 (logits given) only.
// ------------------------------------------------------------------------------------------------
constexpr int HS_SH = 8, HS_SW = 32, HS_RH = HS_SH + 2, HS_RW = HS_SW + 2, HS_NREG = HS_RH * HS_RW;

template <int DC>
__global__ __launch_bounds__(256) void prob_regress_sched_kernel(const float* __restrict__ logits, const float* __restrict__ hyp, float tmp, int mode,
                                                                 int conf_n, float* __restrict__ depth_out, float* __restrict__ conf_out,
                                                                 float* __restrict__ prob_vol, int D_, int H, int W, float ratio,
                                                                 float* __restrict__ next_hyp, int Dn, int tiles_x) {
    __shared__ float s_lo[HS_NREG], s_span[HS_NREG];
    const int D = DC > 0 ? DC : D_;
    const size_t HW = (size_t)H * W;
    const int b = (int)blockIdx.y;
    const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
    const int y0 = ty * HS_SH, x0 = tx * HS_SW;
    for (int task = (int)threadIdx.x; task < HS_NREG; task += 256) {
        const int ry = task / HS_RW, rx = task - ry * HS_RW;
        const int y = y0 - 1 + ry, x = x0 - 1 + rx;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;       // outside the image: never referenced (the bilinear taps are clamped)
        const bool owner = ry >= 1 && ry <= HS_SH && rx >= 1 && rx <= HS_SW;
        const size_t p = (size_t)y * W + x;
        const float* lg = logits + (size_t)b * D * HW + p;
        const float* hp = hyp + (size_t)b * D * HW + p;
        float l[DC > 0 ? DC : 1];
        if (DC > 0) {
#pragma unroll
            for (int d = 0; d < (DC > 0 ? DC : 1); ++d) l[d] = lg[(size_t)d * HW];
        }
        float depth, conf;
        head_from_logits<DC>(l, lg, hp, HW, D, tmp, mode, conf_n, (owner && prob_vol) ? prob_vol + (size_t)b * D * HW + p : nullptr, depth, conf);
        if (owner) {
            depth_out[(size_t)b * HW + p] = depth;
            conf_out[(size_t)b * HW + p] = conf;
        }
        const float h1 = hp[HW], h2 = hp[2 * HW];
        const float last_itv = 1.0f / h2 - 1.0f / h1;                                // module.py:708
        const float inv_min = 1.0f / depth + ratio * last_itv;
        const float inv_max = 1.0f / depth - ratio * last_itv;
        s_lo[task] = inv_max;
        s_span[task] = inv_min - inv_max;
    }
    __syncthreads();
    const int Hn = 2 * H, Wn = 2 * W;
    const size_t HWn = (size_t)Hn * Wn;
    for (int o = (int)threadIdx.x; o < 4 * HS_SH * HS_SW; o += 256) {
        const int oy = o / (2 * HS_SW), ox = o - oy * (2 * HS_SW);
        const int Y = 2 * y0 + oy, X = 2 * x0 + ox;
        if (Y >= Hn || X >= Wn) continue;
        int ya, yb, xa, xb;
        float ly0, ly1, lx0, lx1;
        lin_coord(Y, H, Hn, &ya, &yb, &ly0, &ly1);
        lin_coord(X, W, Wn, &xa, &xb, &lx0, &lx1);
        const int r0 = (ya - (y0 - 1)) * HS_RW, r1 = (yb - (y0 - 1)) * HS_RW, c0 = xa - (x0 - 1), c1 = xb - (x0 - 1);
        const int q[4] = {r0 + c0, r0 + c1, r1 + c0, r1 + c1};
        float lo[4], span[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = s_lo[q[k]]; span[k] = s_span[q[k]]; }
        float* dst = next_hyp + (size_t)b * Dn * HWn + (size_t)Y * Wn + X;
        for (int d = 0; d < Dn; ++d) {
            float c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = lo[k] + span[k] * ((float)d / (float)(Dn - 1));
            const float v = ly0 * (lx0 * c[0] + lx1 * c[1]) + ly1 * (lx0 * c[2] + lx1 * c[3]);
            dst[(size_t)d * HWn] = 1.0f / v;
        }
    }
}

// a16: mean over stages of nearest-upsampled confidences (ConfPtrs: above)
__global__ void confidence_average_kernel(ConfPtrs cp, float* __restrict__ out, int H, int W) {
    const int HW = H * W;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int b = (int)blockIdx.y;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float s = 0.0f;
    for (int i = 0; i < cp.n; ++i) {
        const int hs = H >> cp.shift[i], ws = W >> cp.shift[i];
        s += cp.p[i][(size_t)b * hs * ws + (size_t)(y >> cp.shift[i]) * ws + (x >> cp.shift[i])];
    }
    out[(size_t)b * HW + p] = s / (float)cp.n;
}

// NCDHW <-> channel-last (module-level API only; the fused stage never needs them)
__global__ void ncdhw_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t vox) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= vox) return;
    for (int c = 0; c < C; ++c) y[(b * vox + i) * C + c] = x[(b * C + c) * vox + i];
}

__global__ void cl_to_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t vox) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= vox) return;
    for (int c = 0; c < C; ++c) y[(b * C + c) * vox + i] = x[(b * vox + i) * C + c];
}

template <int KS>
static int launch_head(const float* in, const float* pw, const float* pb, const float* hyp, float tmp, int mode, int conf_n,
                       float* depth, float* conf, float* pv, float* pre, int B, int D, int H, int W, hipStream_t st,
                       ConfPtrs cavg = ConfPtrs{}, float* conf_avg = nullptr) {
    const dim3 grid(ceil_div((long long)H * W, 256), B), block(256);
#define MVS_HEAD_CASE(DC)                                                                                                     \
    case DC:                                                                                                                  \
        hipLaunchKernelGGL((prob_regress_kernel<DC, KS>), grid, block, 0, st, in, pw, pb, hyp, tmp, mode, conf_n, depth, conf, pv, pre, D, H, W, cavg, conf_avg); \
        break;
    switch (D) {
        MVS_HEAD_CASE(4)
        MVS_HEAD_CASE(8)
        MVS_HEAD_CASE(16)
        MVS_HEAD_CASE(32)
        MVS_HEAD_CASE(48)
        default:
            if (!pre) { set_error("prob_regress: D=%d has no register-resident variant; pass a prob_volume_pre buffer", D); return MVS_ERR_ARG; }
            hipLaunchKernelGGL((prob_regress_kernel<0, KS>), grid, block, 0, st, in, pw, pb, hyp, tmp, mode, conf_n, depth, conf, pv, pre, D, H, W, cavg, conf_avg);
    }
#undef MVS_HEAD_CASE
    return check_launch("prob_regress_kernel");
}

}  // namespace mvs

using namespace mvs;

static int check_head(const char* who, const void* in, const float* hyp, float* depth, float* conf, int mode, int B, int D, int H, int W) {
    if (!in || !hyp || !depth || !conf) { set_error("%s: null pointer", who); return MVS_ERR_ARG; }
    if (B < 1 || D < 1 || H < 1 || W < 1) { set_error("%s: bad shape", who); return MVS_ERR_ARG; }
    if (mode < MVS_HEAD_CE_EVAL || mode > MVS_HEAD_REG) { set_error("%s: bad mode %d", who, mode); return MVS_ERR_ARG; }
    return MVS_OK;
}

extern "C" int mvs_prob_regress_fwd(const float* feat_cl, const float* prob_w, const float* prob_b, int prob_ksize, const float* hyp,
                                    float tmp, int mode, int conf_n, float* depth, float* conf, float* prob_volume,
                                    float* prob_volume_pre, int B, int D, int H, int W, void* stream) {
    int rc = check_head("mvs_prob_regress_fwd", feat_cl, hyp, depth, conf, mode, B, D, H, W);
    if (rc != MVS_OK) return rc;
    if (!prob_w) { set_error("mvs_prob_regress_fwd: null weights"); return MVS_ERR_ARG; }
    if (prob_ksize == 1) {
        if (!prob_b) { set_error("mvs_prob_regress_fwd: the 1x1x1 head has a bias (module.py:486)"); return MVS_ERR_ARG; }
        return launch_head<1>(feat_cl, prob_w, prob_b, hyp, tmp, mode, conf_n, depth, conf, prob_volume, prob_volume_pre, B, D, H, W, (hipStream_t)stream);
    }
    if (prob_ksize == 3)
        return launch_head<3>(feat_cl, prob_w, prob_b, hyp, tmp, mode, conf_n, depth, conf, prob_volume, prob_volume_pre, B, D, H, W, (hipStream_t)stream);
    set_error("mvs_prob_regress_fwd: prob kernel size %d unsupported (1 or 3)", prob_ksize);
    return MVS_ERR_UNSUPPORTED;
}

extern "C" int mvs_softmax_regress_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth, float* conf,
                                       float* prob_volume, int B, int D, int H, int W, void* stream) {
    int rc = check_head("mvs_softmax_regress_fwd", logits, hyp, depth, conf, mode, B, D, H, W);
    if (rc != MVS_OK) return rc;
    // KS = 0 reads logits; for run-time D they are re-read from the same buffer (never written: KS == 0)
    return launch_head<0>(logits, nullptr, nullptr, hyp, tmp, mode, conf_n, depth, conf, prob_volume, const_cast<float*>(logits), B, D, H, W, (hipStream_t)stream);
}

extern "C" int mvs_softmax_regress_confavg_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth, float* conf,
                                              float* prob_volume, const float* const* prev_conf_host_ptrs, const int* prev_shifts_host, int n_prev,
                                              float* conf_avg, int B, int D, int H, int W, void* stream) {
    int rc = check_head("mvs_softmax_regress_confavg_fwd", logits, hyp, depth, conf, mode, B, D, H, W);
    if (rc != MVS_OK) return rc;
    if (!conf_avg || n_prev < 0 || n_prev > 7 || (n_prev > 0 && (!prev_conf_host_ptrs || !prev_shifts_host))) { set_error("mvs_softmax_regress_confavg_fwd: bad arguments (0..7 earlier stages)"); return MVS_ERR_ARG; }
    ConfPtrs cp;
    cp.n = n_prev;
    for (int i = 0; i < 8; ++i) { cp.p[i] = i < n_prev ? prev_conf_host_ptrs[i] : nullptr; cp.shift[i] = i < n_prev ? prev_shifts_host[i] : 0; }
    for (int i = 0; i < n_prev; ++i)
        if (!cp.p[i] || cp.shift[i] < 0 || cp.shift[i] > 16 || ((H >> cp.shift[i]) << cp.shift[i]) != H || ((W >> cp.shift[i]) << cp.shift[i]) != W) {
            set_error("mvs_softmax_regress_confavg_fwd: earlier stage %d must be a 2^shift-times smaller map of this %dx%d stage", i, H, W);
            return MVS_ERR_ARG;
        }
    return launch_head<0>(logits, nullptr, nullptr, hyp, tmp, mode, conf_n, depth, conf, prob_volume, const_cast<float*>(logits), B, D, H, W, (hipStream_t)stream, cp, conf_avg);
}

extern "C" int mvs_softmax_regress_schedule_fwd(const float* logits, const float* hyp, float tmp, int mode, int conf_n, float* depth, float* conf,
                                               float* prob_volume, float ratio, float* next_hyp, int next_D, int B, int D, int H, int W, void* stream) {
    int rc = check_head("mvs_softmax_regress_schedule_fwd", logits, hyp, depth, conf, mode, B, D, H, W);
    if (rc != MVS_OK) return rc;
    if (!next_hyp || next_D < 2 || D < 3 || H < 2 || W < 2) { set_error("mvs_softmax_regress_schedule_fwd: bad arguments (needs D >= 3 hypotheses, module.py:708, and next_D >= 2)"); return MVS_ERR_ARG; }
    const int tiles_x = (int)ceil_div(W, HS_SW), tiles_y = (int)ceil_div(H, HS_SH);
    const dim3 grid(tiles_x * tiles_y, B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MVS_HS_CASE(DC)                                                                                                                   \
    case DC:                                                                                                                              \
        hipLaunchKernelGGL((prob_regress_sched_kernel<DC>), grid, block, 0, st, logits, hyp, tmp, mode, conf_n, depth, conf, prob_volume, D, H, W, ratio, \
                           next_hyp, next_D, tiles_x);                                                                                    \
        break;
    switch (D) {
        MVS_HS_CASE(4)
        MVS_HS_CASE(8)
        MVS_HS_CASE(16)
        MVS_HS_CASE(32)
        MVS_HS_CASE(48)
        default:
            hipLaunchKernelGGL((prob_regress_sched_kernel<0>), grid, block, 0, st, logits, hyp, tmp, mode, conf_n, depth, conf, prob_volume, D, H, W, ratio,
                               next_hyp, next_D, tiles_x);
    }
#undef MVS_HS_CASE
    return check_launch("prob_regress_sched_kernel");
}

extern "C" int mvs_depth_regression_fwd(const float* p, const float* depth_values, float* out, int B, int D, int H, int W, void* stream) {
    if (!p || !depth_values || !out || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_depth_regression_fwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(depth_regression_kernel, dim3(ceil_div((long long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, p, depth_values, out, D, H * W);
    return check_launch("depth_regression_kernel");
}

extern "C" int mvs_conf_regression_fwd(const float* p, int n, float* out, int B, int D, int H, int W, void* stream) {
    if (!p || !out || n < 1 || B < 1 || D < 1 || H < 1 || W < 1) { set_error("mvs_conf_regression_fwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(conf_regression_kernel, dim3(ceil_div((long long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, p, n, out, D, H * W);
    return check_launch("conf_regression_kernel");
}

extern "C" int mvs_init_range_fwd(const float* depth_values, int N, int inverse, float* hyp, int B, int D, int H, int W, void* stream) {
    if (!depth_values || !hyp || N < 1 || B < 1 || D < 2 || H < 1 || W < 1) { set_error("mvs_init_range_fwd: bad arguments"); return MVS_ERR_ARG; }
    const int HW = H * W;
    const unsigned gx = ceil_div(HW, 256) > 64 ? 64 : ceil_div(HW, 256);
    hipLaunchKernelGGL(init_range_kernel, dim3(gx, D, B), dim3(256), 0, (hipStream_t)stream, depth_values, N, inverse, hyp, D, HW);
    return check_launch("init_range_kernel");
}

extern "C" int mvs_init_range_pixel_fwd(const float* depth_values, int N, int inverse, float* hyp, int B, int D, int H, int W, void* stream) {
    if (!depth_values || !hyp || N < 1 || B < 1 || D < 2 || H < 1 || W < 1) { set_error("mvs_init_range_pixel_fwd: bad arguments"); return MVS_ERR_ARG; }
    const int HW = H * W;
    const unsigned gx = ceil_div(HW, 256) > 256 ? 256 : ceil_div(HW, 256);
    hipLaunchKernelGGL(init_range_pixel_kernel, dim3(gx, D, B), dim3(256), 0, (hipStream_t)stream, depth_values, N, inverse, hyp, D, HW);
    return check_launch("init_range_pixel_kernel");
}

extern "C" int mvs_schedule_inverse_range_fwd(const float* prev_depth, const float* prev_hyp, int Dprev, float ratio, int shift, float* hyp,
                                              int B, int D, int H, int W, void* stream) {
    if (!prev_depth || !prev_hyp || !hyp || Dprev < 3 || B < 1 || D < 2 || H < 2 || W < 2) { set_error("mvs_schedule_inverse_range_fwd: bad arguments (needs Dprev >= 3, module.py:708)"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(schedule_range_kernel, dim3(ceil_div((long long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, prev_depth, prev_hyp,
                       Dprev, ratio, (const float*)nullptr, 0, hyp, D, H, W, H / 2, W / 2, shift ? 1 : 0);
    return check_launch("schedule_range_kernel");
}

extern "C" int mvs_schedule_range_fwd(const float* prev_depth, const float* interval, int interval_per_pixel, float* hyp, int B, int D, int H,
                                      int W, void* stream) {
    if (!prev_depth || !interval || !hyp || B < 1 || D < 2 || H < 2 || W < 2) { set_error("mvs_schedule_range_fwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(schedule_range_kernel, dim3(ceil_div((long long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, prev_depth,
                       (const float*)nullptr, 0, 0.0f, interval, 1, hyp, D, H, W, H / 2, W / 2, interval_per_pixel ? 1 : 0);
    return check_launch("schedule_range_kernel");
}

extern "C" int mvs_confidence_average(const float* const* conf_host_ptrs, const int* shifts_host, int n_stages, float* out, int B, int H,
                                      int W, void* stream) {
    if (!conf_host_ptrs || !shifts_host || !out || n_stages < 1 || n_stages > 8) { set_error("mvs_confidence_average: bad arguments (1..8 stages)"); return MVS_ERR_ARG; }
    ConfPtrs cp;
    cp.n = n_stages;
    for (int i = 0; i < 8; ++i) { cp.p[i] = i < n_stages ? conf_host_ptrs[i] : nullptr; cp.shift[i] = i < n_stages ? shifts_host[i] : 0; }
    hipLaunchKernelGGL(confidence_average_kernel, dim3(ceil_div((long long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, cp, out, H, W);
    return check_launch("confidence_average_kernel");
}

extern "C" int mvs_ncdhw_to_cl(const float* x, float* y_cl, int B, int C, int D, int H, int W, void* stream) {
    if (!x || !y_cl || B < 1 || C < 1) { set_error("mvs_ncdhw_to_cl: bad arguments"); return MVS_ERR_ARG; }
    const size_t vox = (size_t)D * H * W;
    hipLaunchKernelGGL(ncdhw_to_cl_kernel, dim3((unsigned)((vox + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, y_cl, C, vox);
    return check_launch("ncdhw_to_cl_kernel");
}

extern "C" int mvs_cl_to_ncdhw(const float* x_cl, float* y, int B, int C, int D, int H, int W, void* stream) {
    if (!x_cl || !y || B < 1 || C < 1) { set_error("mvs_cl_to_ncdhw: bad arguments"); return MVS_ERR_ARG; }
    const size_t vox = (size_t)D * H * W;
    hipLaunchKernelGGL(cl_to_ncdhw_kernel, dim3((unsigned)((vox + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x_cl, y, C, vox);
    return check_launch("cl_to_ncdhw_kernel");
}
