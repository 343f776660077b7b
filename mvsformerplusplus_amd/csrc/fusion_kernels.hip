// Depth-map filtering after inference: reprojection-consistency checks between the depth map of a
// reference view and those of its source views, misc/fusion.py (Vis-MVSNet's filters as vendored by the reference) as driven
// by test.py:388-409 ("pcd", static thresholds) and test.py:455-483 ("dpcd", dynamic consistency).
//
// The reference materialises ~25 [N,h,w,4,1] intermediates per call.  Here one work-item owns one reference pixel and
// walks the source views: project with the reference depth, take the bilinear sample in the source view (one gather
// of 4 depth taps - the same access pattern as the warp kernels), project back, apply the thresholds, accumulate the
// masked depths, and finish with the averaged depth, the geometric mask, the final mask and the world point.  The
// [n,v,3,h,w] reprojection tensor is only written when the caller asks for it (API parity with get_reproj*); the filters
// can also start from such a tensor (vis_filter / vis_filter_dynamic called on their own).
// HBM-bound: algorithmic bytes per pixel = 4 (ref depth) + 4 (conf) + v * 4 (source depth taps, L2-resident re-use) + 22 out.
#include "mvs_common.h"

namespace mvs {

struct CamPack {          // 50 floats per camera, produced by fusion_prepare_cams_kernel
    float K[9], Kinv[9], E[16], Einv[16];
};

struct FusionArgs {
    const float* ref_depth;    // [n,h,w]
    const float* ref_conf;     // [n,h,w] or nullptr (no photometric mask)
    const float* srcs_depth;   // [n,v,h,w]
    const float* srcs_conf;    // [n,v,h,w] or nullptr (static mode: source depths with conf <= thresh count as holes, test.py:389-392)
    const CamPack* ref_cam;    // [n]
    const CamPack* srcs_cam;   // [n,v]
    const float* xyd_in;       // [n,v,3,h,w] or nullptr: start from an existing reprojection
    const float* in_range_in;  // [n,v,h,w] or nullptr (static, with xyd_in)
    float* xyd_out;            // [n,v,3,h,w] or nullptr
    float* in_range_out;       // [n,v,h,w] or nullptr (static)
    uint8_t* vis_masks;        // static [n,v,h,w]; dynamic [n,v,v-1,h,w]; or nullptr
    float* depth;              // [n,h,w] averaged depth, or nullptr (no filtering requested)
    uint8_t* geo_mask;         // [n,h,w]
    uint8_t* mask;             // [n,h,w]   photometric & geometric
    float* points;             // [n,3,h,w] world coordinates of the averaged depth, or nullptr
    float conf_thresh, p0, p1; // static: p0 = img_dist_thresh, p1 = depth_thresh; dynamic: p0 = dist_base, p1 = rel_diff_base
    float vthresh;             // static
    int n, v, h, w;
};

__device__ __forceinline__ void mat3(const float* m, float x, float y, float z, float* o) {
    o[0] = m[0] * x + m[1] * y + m[2] * z;
    o[1] = m[3] * x + m[4] * y + m[5] * z;
    o[2] = m[6] * x + m[7] * y + m[8] * z;
}
__device__ __forceinline__ void mat4(const float* m, const float* p, float* o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = m[i * 4] * p[0] + m[i * 4 + 1] * p[1] + m[i * 4 + 2] * p[2] + m[i * 4 + 3] * p[3];
}

// image point (homogeneous, third component 1) + depth -> world, fusion.py:23-34
__device__ __forceinline__ void img2world(const CamPack& c, float u, float v, float depth, float* world) {
    float cam[4];
    mat3(c.Kinv, u, v, 1.0f, cam);
    const float d = cam[2] + 1e-9f;
    cam[0] = cam[0] / d * depth; cam[1] = cam[1] / d * depth; cam[2] = cam[2] / d * depth; cam[3] = 1.0f;
    mat4(c.Einv, cam, world);
    const float w = world[3] + 1e-9f;
#pragma unroll
    for (int i = 0; i < 4; ++i) world[i] /= w;
}
// world -> camera (homogeneous, normalised) -> image, fusion.py:37-47; returns camera z
__device__ __forceinline__ float world2img(const CamPack& c, const float* world, float* uv) {
    float cam[4];
    mat4(c.E, world, cam);
    const float w = cam[3] + 1e-9f;
#pragma unroll
    for (int i = 0; i < 4; ++i) cam[i] /= w;
    const float w2 = cam[3] + 1e-9f;
    float img[3];
    mat3(c.K, cam[0] / w2, cam[1] / w2, cam[2] / w2, img);
    const float z = img[2] + 1e-9f;
    uv[0] = img[0] / z;
    uv[1] = img[1] / z;
    return cam[2];
}

// bilinear taps of grid_sample(align_corners=True, zeros padding) at normalised coordinates (gx, gy)
struct Bilin { int x0, y0; float wx1, wy1; };
__device__ __forceinline__ Bilin bilin(float gx, float gy, int w, int h) {
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    Bilin b;
    b.x0 = (int)fx; b.y0 = (int)fy; b.wx1 = ix - fx; b.wy1 = iy - fy;
    return b;
}

template <bool DYN>
__global__ __launch_bounds__(256) void fusion_kernel(const FusionArgs a) {
    const int HW = a.h * a.w;
    const int p = (int)(blockIdx.x * 256 + threadIdx.x);
    const int b = (int)blockIdx.y;
    if (p >= HW) return;
    const int y = p / a.w, x = p - y * a.w;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;                      // pixel centres, fusion.py:9-10
    const CamPack& rc = a.ref_cam[b];
    const float rd = a.ref_depth[(size_t)b * HW + p];
    float world_ref[4];
    if (a.xyd_in == nullptr) img2world(rc, px, py, rd, world_ref);

    float sum_depth = 0.0f, sum_mask = 0.0f;
    int counts[kMaxSrcViews];                                                    // dynamic: #views passing threshold i = k + 2
#pragma unroll
    for (int k = 0; k < kMaxSrcViews; ++k) counts[k] = 0;

    for (int v = 0; v < a.v; ++v) {
        float rx, ry, rz, inr = 1.0f;
        if (a.xyd_in != nullptr) {
            const float* q = a.xyd_in + ((size_t)(b * a.v + v) * 3) * HW + p;
            rx = q[0]; ry = q[HW]; rz = q[2 * HW];
            if (!DYN && a.in_range_in != nullptr) inr = a.in_range_in[(size_t)(b * a.v + v) * HW + p];
        } else {
            const CamPack& sc = a.srcs_cam[b * a.v + v];
            const float* sd = a.srcs_depth + (size_t)(b * a.v + v) * HW;
            const float* scf = (!DYN && a.srcs_conf != nullptr) ? a.srcs_conf + (size_t)(b * a.v + v) * HW : nullptr;
            float uv[2];
            world2img(sc, world_ref, uv);
            if (DYN) {
                // fusion.py:133-139: sample the source depth at the projected position, lift it, bring it back
                const Bilin t = bilin(uv[0] / ((float)(a.w - 1) / 2.0f) - 1.0f, uv[1] / ((float)(a.h - 1) / 2.0f) - 1.0f, a.w, a.h);
                float wd = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = t.x0 + (j & 1), yy = t.y0 + (j >> 1);
                    const float wgt = ((j & 1) ? t.wx1 : 1.0f - t.wx1) * ((j >> 1) ? t.wy1 : 1.0f - t.wy1);
                    if (xx >= 0 && xx < a.w && yy >= 0 && yy < a.h) wd += wgt * sd[yy * a.w + xx];
                }
                float ws[4], ruv[2];
                img2world(sc, uv[0], uv[1], wd, ws);
                rz = world2img(rc, ws, ruv);
                rx = ruv[0]; ry = ruv[1];
            } else {
                // fusion.py:50-97: the source view's (x, y, depth)-in-reference map, bilinearly sampled at the projected position
                float gx = uv[0] / (float)a.w * 2.0f - 1.0f, gy = uv[1] / (float)a.h * 2.0f - 1.0f;
                gx = fminf(fmaxf(gx, -1.1f), 1.1f);
                gy = fminf(fmaxf(gy, -1.1f), 1.1f);
                inr = (gx >= -1.0f && gx <= 1.0f && gy >= -1.0f && gy <= 1.0f) ? 1.0f : 0.0f;
                const Bilin t = bilin(gx, gy, a.w, a.h);
                rx = ry = rz = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = t.x0 + (j & 1), yy = t.y0 + (j >> 1);
                    const float wgt = ((j & 1) ? t.wx1 : 1.0f - t.wx1) * ((j >> 1) ? t.wy1 : 1.0f - t.wy1);
                    if (xx < 0 || xx >= a.w || yy < 0 || yy >= a.h) continue;
                    float d = sd[yy * a.w + xx];
                    if (scf != nullptr && !(scf[yy * a.w + xx] > a.conf_thresh)) d = 0.0f;        // test.py:389-392
                    float ws[4], suv[2];
                    img2world(sc, (float)xx + 0.5f, (float)yy + 0.5f, d, ws);
                    const float z = world2img(rc, ws, suv);
                    rx += wgt * suv[0]; ry += wgt * suv[1]; rz += wgt * z;
                }
            }
            if (a.xyd_out != nullptr) {
                float* q = a.xyd_out + ((size_t)(b * a.v + v) * 3) * HW + p;
                q[0] = rx; q[HW] = ry; q[2 * HW] = rz;
            }
            if (!DYN && a.in_range_out != nullptr) a.in_range_out[(size_t)(b * a.v + v) * HW + p] = inr;
        }
        if (a.depth == nullptr) continue;                                          // reprojection only
        const float dx = rx - px, dy = ry - py;
        const float dist = sqrtf(dx * dx + dy * dy);
        if (DYN) {
            // fusion.py:156-168: thresholds i / dist_base and i / rel_diff_base for i = 2 .. v; the last one gates the average
            const float ddiff = fabsf(rd - rz) / rd;
            bool last = false;
#pragma unroll
            for (int k = 0; k < kMaxSrcViews - 1; ++k) {                           // static trip count: counts[] stays in registers
                if (k + 2 > a.v) break;
                const float i = (float)(k + 2);
                const bool m = (dist < i / a.p0) && (ddiff < i / a.p1);
                counts[k] += m ? 1 : 0;
                last = m;
                if (a.vis_masks != nullptr) a.vis_masks[((size_t)(b * a.v + v) * (a.v - 1) + k) * HW + p] = m ? 1 : 0;
            }
            if (last) { sum_depth += rz; sum_mask += 1.0f; }
        } else {
            // fusion.py:100-114
            const bool m = (inr > 0.0f) && (dist < a.p0) && (fabsf(rd - rz) < fmaxf(rd, rz) * a.p1);
            if (m) { sum_depth += rz; sum_mask += 1.0f; }
            if (a.vis_masks != nullptr) a.vis_masks[(size_t)(b * a.v + v) * HW + p] = m ? 1 : 0;
        }
    }
    if (a.depth == nullptr) return;
    const float ave = (sum_depth + rd) / (sum_mask + 1.0f);
    bool geo;
    if (DYN) {
        geo = sum_mask >= (float)(a.v + 1);                                        // test.py:472 (never true), kept for fidelity
#pragma unroll
        for (int k = 0; k < kMaxSrcViews - 1; ++k)
            if (k + 2 <= a.v) geo = geo || (counts[k] >= k + 2);                   // test.py:473-474
    } else {
        geo = sum_mask >= a.vthresh - 1.1f;                                        // fusion.py:108
    }
    const bool prob = a.ref_conf == nullptr || a.ref_conf[(size_t)b * HW + p] > a.conf_thresh;
    a.depth[(size_t)b * HW + p] = ave;
    if (a.geo_mask != nullptr) a.geo_mask[(size_t)b * HW + p] = geo ? 1 : 0;
    if (a.mask != nullptr) a.mask[(size_t)b * HW + p] = (geo && prob) ? 1 : 0;
    if (a.points != nullptr) {
        float wp[4];
        img2world(rc, px, py, ave, wp);
        float* o = a.points + (size_t)b * 3 * HW + p;
        o[0] = wp[0]; o[HW] = wp[1]; o[2 * HW] = wp[2];
    }
}

// K, K^-1, E, E^-1 of every camera [N,2,4,4] (0 = extrinsic, 1 = intrinsic in the top-left 3x3); inverses in fp64
__global__ void fusion_prepare_cams_kernel(const float* __restrict__ cams, int N, CamPack* __restrict__ out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= N) return;
    const float* E = cams + (size_t)i * 32;
    const float* K4 = E + 16;
    CamPack c;
    double k[9];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) { c.K[r * 3 + q] = K4[r * 4 + q]; k[r * 3 + q] = (double)K4[r * 4 + q]; }
    const double c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
    const double det = k[0] * c00 + k[1] * c01 + k[2] * c02;
    const double ki[9] = {c00 / det, (k[2] * k[7] - k[1] * k[8]) / det, (k[1] * k[5] - k[2] * k[4]) / det,
                          c01 / det, (k[0] * k[8] - k[2] * k[6]) / det, (k[2] * k[3] - k[0] * k[5]) / det,
                          c02 / det, (k[1] * k[6] - k[0] * k[7]) / det, (k[0] * k[4] - k[1] * k[3]) / det};
    for (int j = 0; j < 9; ++j) c.Kinv[j] = (float)ki[j];
    // 4x4 Gauss-Jordan with partial pivoting
    double m[4][8];
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 4; ++q) { c.E[r * 4 + q] = E[r * 4 + q]; m[r][q] = (double)E[r * 4 + q]; m[r][4 + q] = r == q ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(m[r][col]) > fabs(m[piv][col])) piv = r;
        if (piv != col) for (int q = 0; q < 8; ++q) { const double t = m[col][q]; m[col][q] = m[piv][q]; m[piv][q] = t; }
        const double inv = 1.0 / m[col][col];
        for (int q = 0; q < 8; ++q) m[col][q] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = m[r][col];
            for (int q = 0; q < 8; ++q) m[r][q] -= f * m[col][q];
        }
    }
    for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) c.Einv[r * 4 + q] = (float)m[r][4 + q];
    out[i] = c;
}

__global__ __launch_bounds__(256) void fusion_ave_kernel(const float* __restrict__ ref_depth, const float* __restrict__ xyd,
                                                         const float* __restrict__ masks, float* __restrict__ out, int v, int HW) {
    const int p = (int)(blockIdx.x * 256 + threadIdx.x), b = (int)blockIdx.y;
    if (p >= HW) return;
    float s = 0.0f, c = 0.0f;
    for (int i = 0; i < v; ++i) {
        const float m = masks[(size_t)(b * v + i) * HW + p];
        s += xyd[((size_t)(b * v + i) * 3 + 2) * HW + p] * m;
        c += m;
    }
    out[(size_t)b * HW + p] = (s + ref_depth[(size_t)b * HW + p]) / (c + 1.0f);                       // fusion.py:113
}

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_fusion_campack_floats(void) { return sizeof(CamPack) / sizeof(float); }

extern "C" int mvs_fusion_prepare_cams(const float* cams, int N, float* packed, void* stream) {
    if (!cams || !packed || N < 1) { set_error("mvs_fusion_prepare_cams: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(fusion_prepare_cams_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, cams, N, reinterpret_cast<CamPack*>(packed));
    return check_launch("fusion_prepare_cams_kernel");
}

extern "C" int mvs_fusion_filter_fwd(int dynamic, const float* ref_depth, const float* ref_conf, const float* srcs_depth, const float* srcs_conf,
                                     const float* ref_cam_packed, const float* srcs_cam_packed, const float* xyd_in, const float* in_range_in,
                                     float conf_thresh, float p0, float p1, float vthresh, float* xyd_out, float* in_range_out,
                                     uint8_t* vis_masks, float* depth, uint8_t* geo_mask, uint8_t* mask, float* points, int n, int v, int h, int w,
                                     void* stream) {
    if (!ref_depth || n < 1 || v < 1 || h < 1 || w < 1) { set_error("mvs_fusion_filter_fwd: bad arguments"); return MVS_ERR_ARG; }
    if (v > kMaxSrcViews) { set_error("mvs_fusion_filter_fwd: %d source views > %d", v, kMaxSrcViews); return MVS_ERR_UNSUPPORTED; }
    if (!xyd_in && (!srcs_depth || !ref_cam_packed || !srcs_cam_packed)) { set_error("mvs_fusion_filter_fwd: depths and cameras are needed without xyd_in"); return MVS_ERR_ARG; }
    if (points && !ref_cam_packed) { set_error("mvs_fusion_filter_fwd: points need the reference camera"); return MVS_ERR_ARG; }
    if (!depth && (vis_masks || geo_mask || mask || points)) { set_error("mvs_fusion_filter_fwd: filter outputs need `depth`"); return MVS_ERR_ARG; }
    if (dynamic && v < 2) { set_error("mvs_fusion_filter_fwd: the dynamic filter needs >= 2 source views"); return MVS_ERR_ARG; }
    FusionArgs a;
    a.ref_depth = ref_depth; a.ref_conf = ref_conf; a.srcs_depth = srcs_depth; a.srcs_conf = srcs_conf;
    a.ref_cam = reinterpret_cast<const CamPack*>(ref_cam_packed); a.srcs_cam = reinterpret_cast<const CamPack*>(srcs_cam_packed);
    a.xyd_in = xyd_in; a.in_range_in = in_range_in; a.xyd_out = xyd_out; a.in_range_out = in_range_out; a.vis_masks = vis_masks;
    a.depth = depth; a.geo_mask = geo_mask; a.mask = mask; a.points = points;
    a.conf_thresh = conf_thresh; a.p0 = p0; a.p1 = p1; a.vthresh = vthresh; a.n = n; a.v = v; a.h = h; a.w = w;
    const dim3 grid((unsigned)(((size_t)h * w + 255) / 256), n);
    if (dynamic) hipLaunchKernelGGL(fusion_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(fusion_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("fusion_kernel");
}

extern "C" int mvs_fusion_ave_fwd(const float* ref_depth, const float* reproj_xyd, const float* masks, float* out, int n, int v, int h, int w,
                                  void* stream) {
    if (!ref_depth || !reproj_xyd || !masks || !out || n < 1 || v < 1 || h < 1 || w < 1) { set_error("mvs_fusion_ave_fwd: bad arguments"); return MVS_ERR_ARG; }
    hipLaunchKernelGGL(fusion_ave_kernel, dim3((unsigned)(((size_t)h * w + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, ref_depth, reproj_xyd,
                       masks, out, v, h * w);
    return check_launch("fusion_ave_kernel");
}
