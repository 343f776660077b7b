/* ORACLE (test infrastructure, NOT the product): plain-C restatement, from their published definitions, of the
 * PyTorch operators the reference's depth hot path calls (torch is a third-party dependency of the reference,
 * requirements.txt:9 torch==2.1.2, not vendored under /root/reference):
 *
 *   F.grid_sample(bilinear, zeros, align_corners=True)   models/warping.py:105
 *   nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d            models/module.py:109,148,191,391,467-486
 *   BatchNorm (eval) + ReLU                              models/module.py:110-125,193-197
 *   softmax over depth, entropy, depth regression        models/cost_volume.py:90-117
 *   F.interpolate(trilinear, align_corners=True)         models/module.py:723
 *
 * plus the reference's own arithmetic around them (homography warp warping.py:80-103, group-wise correlation
 * cost_volume.py:79-87, weighted aggregation :97-101).  Direct loops, fp32 accumulation in the natural order,
 * no blocking, no SIMD tricks: slow but obviously what the definitions say.  oracle/c_path.py composes these
 * into a whole stage with numpy; tests/test_oracle_c.py checks it against the golden vectors and the torch
 * restatement (oracle/ref_path.py).  Layouts are PyTorch's: NCHW / NCDHW, row-major, fp32.
 *
 * Build: make -C oracle   ->  oracle/_build/libstage_ref.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX3(c, y, x, H, W) (((size_t)(c) * (H) + (y)) * (W) + (x))

/* warping.py:84-106.  hom = {R (9, row-major), t (3)} = (src_proj @ inverse(ref_proj))[:3,:4] split.
 * depth [D,H,W]; src [C,H,W] -> warped [C,D,H,W], mask [D,H,W] (1 = outside / behind camera). */
void ref_homo_warp(const float* src, const float* hom, const float* depth, float* warped, uint8_t* mask, int C, int D, int H, int W) {
    const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float fx = (float)x, fy = (float)y;
                const float qx = hom[0] * fx + hom[1] * fy + hom[2];
                const float qy = hom[3] * fx + hom[4] * fy + hom[5];
                const float qz = hom[6] * fx + hom[7] * fy + hom[8];
                const float dep = depth[IDX3(d, y, x, H, W)];
                const float px = qx * dep + hom[9], py = qy * dep + hom[10], pz = qz * dep + hom[11];
                const float u = px / (pz + 1e-6f), v = py / (pz + 1e-6f);
                const float xn = u / half_w - 1.0f, yn = v / half_h - 1.0f;
                if (mask) mask[IDX3(d, y, x, H, W)] = (xn > 1.0f) || (xn < -1.0f) || (yn > 1.0f) || (yn < -1.0f) || (pz <= 0.0f);
                /* grid_sample: un-normalise with align_corners=True, 4 taps, each tap zero outside the image */
                const float ix = ((xn + 1.0f) / 2.0f) * (float)(W - 1), iy = ((yn + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const int finite = (ix > -2.0f && ix < (float)W + 1.0f && iy > -2.0f && iy < (float)H + 1.0f);
                const int x0 = finite ? (int)fx0 : -5, y0 = finite ? (int)fy0 : -5;
                const float wx1 = ix - fx0, wx0 = fx0 + 1.0f - ix, wy1 = iy - fy0, wy0 = fy0 + 1.0f - iy;
                for (int c = 0; c < C; ++c) {
                    float acc = 0.0f;
                    for (int k = 0; k < 4; ++k) {
                        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                        const float w = ((k & 1) ? wx1 : wx0) * ((k >> 1) ? wy1 : wy0);
                        acc += src[IDX3(c, yy, xx, H, W)] * w;
                    }
                    warped[(((size_t)c * D + d) * H + y) * W + x] = acc;
                }
            }
}

/* cost_volume.py:79-87: ip[g,d,y,x] = mean_{c in group g} ref[c,y,x] * warped[c,d,y,x]  (G == C: plain product) */
void ref_group_corr(const float* ref, const float* warped, float* ip, int C, int G, int D, int H, int W) {
    const int cpg = C / G;
    const size_t HW = (size_t)H * W;
    for (int g = 0; g < G; ++g)
        for (int d = 0; d < D; ++d)
            for (size_t p = 0; p < HW; ++p) {
                float s = 0.0f;
                for (int cc = 0; cc < cpg; ++cc) {
                    const int c = g * cpg + cc;
                    s += ref[(size_t)c * HW + p] * warped[((size_t)c * D + d) * HW + p];
                }
                ip[((size_t)g * D + d) * HW + p] = s / (float)cpg;
            }
}

/* cost_volume.py:90-92: entropy of softmax_D(sum_g ip) */
void ref_entropy(const float* ip, float* ent, int G, int D, int HW) {
    for (int p = 0; p < HW; ++p) {
        float sim[1024];
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) {
            float s = 0.0f;
            for (int g = 0; g < G; ++g) s += ip[((size_t)g * D + d) * HW + p];
            sim[d] = s;
            if (s > m) m = s;
        }
        float den = 0.0f;
        for (int d = 0; d < D; ++d) den += expf(sim[d] - m);
        float e = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pr = expf(sim[d] - m) / den;
            e += -pr * logf(pr + 1e-7f);
        }
        ent[p] = e;
    }
}

/* nn.Conv2d: x [Ci,H,W], w [Co,Ci,k,k], stride 1, zero padding k/2, optional bias */
void ref_conv2d(const float* x, const float* w, const float* bias, float* y, int Ci, int Co, int H, int W, int k) {
    const int pad = k / 2;
    for (int co = 0; co < Co; ++co)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                float s = bias ? bias[co] : 0.0f;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int a = 0; a < k; ++a)
                        for (int b = 0; b < k; ++b) {
                            const int iy = yy + a - pad, ix = xx + b - pad;
                            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                            s += x[IDX3(ci, iy, ix, H, W)] * w[(((size_t)co * Ci + ci) * k + a) * k + b];
                        }
                y[IDX3(co, yy, xx, H, W)] = s;
            }
}

/* nn.Conv3d: x [Ci,D,H,W], w [Co,Ci,k,k,k], stride (sd,sh,sw), zero padding k/2 -> y [Co,OD,OH,OW] */
void ref_conv3d(const float* x, const float* w, const float* bias, float* y, int Ci, int Co, int D, int H, int W, int k, int sd, int sh, int sw) {
    const int pad = k / 2;
    const int OD = (D + 2 * pad - k) / sd + 1, OH = (H + 2 * pad - k) / sh + 1, OW = (W + 2 * pad - k) / sw + 1;
    for (int co = 0; co < Co; ++co)
        for (int oz = 0; oz < OD; ++oz)
            for (int oy = 0; oy < OH; ++oy)
                for (int ox = 0; ox < OW; ++ox) {
                    float s = bias ? bias[co] : 0.0f;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int a = 0; a < k; ++a) {
                            const int iz = oz * sd + a - pad;
                            if (iz < 0 || iz >= D) continue;
                            for (int b = 0; b < k; ++b) {
                                const int iy = oy * sh + b - pad;
                                if (iy < 0 || iy >= H) continue;
                                for (int c = 0; c < k; ++c) {
                                    const int ix = ox * sw + c - pad;
                                    if (ix < 0 || ix >= W) continue;
                                    s += x[(((size_t)ci * D + iz) * H + iy) * W + ix] * w[((((size_t)co * Ci + ci) * k + a) * k + b) * k + c];
                                }
                            }
                        }
                    y[(((size_t)co * OD + oz) * OH + oy) * OW + ox] = s;
                }
}

/* nn.ConvTranspose3d, k = 3, padding 1, stride (sd,sh,sw), output_padding (sd-1,sh-1,sw-1), w [Ci,Co,3,3,3]:
 * the defining scatter  y[co, i*s - p + k] += x[ci, i] * w[ci, co, k]  ->  y [Co, D*sd, H*sh, W*sw] */
void ref_conv_transpose3d(const float* x, const float* w, float* y, int Ci, int Co, int D, int H, int W, int sd, int sh, int sw) {
    const int OD = D * sd, OH = H * sh, OW = W * sw;
    memset(y, 0, sizeof(float) * (size_t)Co * OD * OH * OW);
    for (int ci = 0; ci < Ci; ++ci)
        for (int z = 0; z < D; ++z)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    const float v = x[(((size_t)ci * D + z) * H + yy) * W + xx];
                    for (int co = 0; co < Co; ++co)
                        for (int a = 0; a < 3; ++a) {
                            const int oz = z * sd - 1 + a;
                            if (oz < 0 || oz >= OD) continue;
                            for (int b = 0; b < 3; ++b) {
                                const int oy = yy * sh - 1 + b;
                                if (oy < 0 || oy >= OH) continue;
                                for (int c = 0; c < 3; ++c) {
                                    const int ox = xx * sw - 1 + c;
                                    if (ox < 0 || ox >= OW) continue;
                                    y[(((size_t)co * OD + oz) * OH + oy) * OW + ox] += v * w[((((size_t)ci * Co + co) * 3 + a) * 3 + b) * 3 + c];
                                }
                            }
                        }
                }
}

/* eval-mode BatchNorm (+ optional ReLU), in place over x [C, n] */
void ref_bn_relu(float* x, const float* gamma, const float* beta, const float* mean, const float* var, int C, size_t n, int relu) {
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + 1e-5f);
        for (size_t i = 0; i < n; ++i) {
            float v = (x[(size_t)c * n + i] - mean[c]) * inv * gamma[c] + beta[c];
            x[(size_t)c * n + i] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
}

/* cost_volume.py:105-117 ('ce', eval): prob = softmax_D(logit); depth = sum softmax_D(logit*tmp) * hyp; conf = max prob */
void ref_softmax_regress(const float* logit, const float* hyp, float tmp, float* prob, float* depth, float* conf, int D, int HW) {
    for (int p = 0; p < HW; ++p) {
        float m = -INFINITY, m2 = -INFINITY;
        for (int d = 0; d < D; ++d) {
            const float l = logit[(size_t)d * HW + p];
            if (l > m) m = l;
            if (l * tmp > m2) m2 = l * tmp;
        }
        float den = 0.0f, den2 = 0.0f;
        for (int d = 0; d < D; ++d) {
            den += expf(logit[(size_t)d * HW + p] - m);
            den2 += expf(logit[(size_t)d * HW + p] * tmp - m2);
        }
        float best = 0.0f, dep = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pr = expf(logit[(size_t)d * HW + p] - m) / den;
            prob[(size_t)d * HW + p] = pr;
            if (pr > best) best = pr;
            dep += (expf(logit[(size_t)d * HW + p] * tmp - m2) / den2) * hyp[(size_t)d * HW + p];
        }
        depth[p] = dep;
        conf[p] = best;
    }
}

/* F.interpolate(mode='trilinear', align_corners=True) with unchanged depth: x [D,h,w] -> y [D,H,W] */
void ref_upsample_bilinear_ac(const float* x, float* y, int D, int h, int w, int H, int W) {
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    for (int d = 0; d < D; ++d)
        for (int yy = 0; yy < H; ++yy) {
            const float fy = sh * (float)yy;
            int y0 = (int)fy;
            if (y0 > h - 1) y0 = h - 1;
            const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            for (int xx = 0; xx < W; ++xx) {
                const float fx = sw * (float)xx;
                int x0 = (int)fx;
                if (x0 > w - 1) x0 = w - 1;
                const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const float* s = x + (size_t)d * h * w;
                y[((size_t)d * H + yy) * W + xx] = ly0 * (lx0 * s[y0 * w + x0] + lx1 * s[y0 * w + x1]) + ly1 * (lx0 * s[y1 * w + x0] + lx1 * s[y1 * w + x1]);
            }
        }
}
