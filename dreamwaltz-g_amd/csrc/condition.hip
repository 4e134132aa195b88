// condition.hip -- OpenPose-style condition image of the posed body, on the GPU (include/dwg_condition.h).
//
// Restates, for ONE person and hand_dist_thres = None (what export_pose passes):
//   /root/reference/core/human/smpl_condition.py:82-143,191-235 (projection, invisibility, per-group occlusion thresholds),
//   /root/reference/core/human/open_pose.py:48-333 (draw order, colours, size rule, the 0.4 / 0.6 blend of the limbs).
// The reference does this on the CPU every step: open3d BVH build over the 20 908-triangle body + 128 rays, then ~185 cv2 calls and a
// PIL round trip.  Here: one workgroup per keypoint casts its ray against ALL triangles (2.7 M Moeller-Trumbore tests per image
// in fp64 -- tens of microseconds, no acceleration structure to build for a mesh that moves every step), and the image is formed
// by one thread per pixel that walks the 185 primitives IN THE REFERENCE'S DRAW ORDER (later primitives overwrite / blend over
// earlier ones exactly as sequential cv2 calls do), so no per-primitive pass over the canvas exists.
//
// Rasterisation rules (OpenCV is not in this image -- these restate its published algorithms, see oracle/condition.py):
//   filled circle  : row spans of drawing.cpp's midpoint Circle()            (table c_circle_hw, generated)
//   limb           : ellipse2Poly (integer-degree sine table, cvRound) then the polygon's row span [round(x_left), round(x_right)]
//                    with exact rational edge intersections (fillConvexPoly tracks a left and a right edge per row)
//   hand edge      : pixels within thickness / 2 of the segment (integer arithmetic)
//   blend          : rint(0.4f * old + 0.6f * colour) in fp32, round half to even (cv2.addWeighted on uint8)
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "condition_tables.h"
#include "../../include/dwg_condition.h"

namespace {

constexpr int NB = 18, NH = 21, NF = 68, NLIMB = 17, NEDGE = 20, NVERT = 361;

__device__ const unsigned char c_body_rgb[18][3] = {
    {255, 0, 0}, {255, 85, 0}, {255, 170, 0}, {255, 255, 0}, {170, 255, 0}, {85, 255, 0}, {0, 255, 0}, {0, 255, 85}, {0, 255, 170},
    {0, 255, 255}, {0, 170, 255}, {0, 85, 255}, {0, 0, 255}, {85, 0, 255}, {170, 0, 255}, {255, 0, 255}, {255, 0, 170}, {255, 0, 85}};
__device__ const unsigned char c_limb[17][2] = {{2, 3}, {2, 6}, {3, 4}, {4, 5}, {6, 7}, {7, 8}, {2, 9}, {9, 10}, {10, 11}, {2, 12},
                                                {12, 13}, {13, 14}, {2, 1}, {1, 15}, {15, 17}, {1, 16}, {16, 18}};      // 1-based
__device__ const unsigned char c_flip[18] = {0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16};
__device__ const unsigned char c_hand_edge[20][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {0, 5}, {5, 6}, {6, 7}, {7, 8}, {0, 9}, {9, 10},
                                                     {10, 11}, {11, 12}, {0, 13}, {13, 14}, {14, 15}, {15, 16}, {0, 17}, {17, 18},
                                                     {18, 19}, {19, 20}};

// ---------------------------------------------------------------------------------------------------------------------
// keypoints: one workgroup per keypoint
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cond_keypoints(int K, const float* __restrict__ kps, const float* __restrict__ ext,
                                                        const float* __restrict__ intr, int F, const float* __restrict__ verts,
                                                        const int* __restrict__ tris, const unsigned char* __restrict__ groups,
                                                        float thr_body, float thr_hand, float thr_face, int cull,
                                                        double* __restrict__ rows) {
    __shared__ double s_min[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    double R[9], T[3], c[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = (double)ext[4 * i + j];
        T[i] = (double)ext[4 * i + 3];
    }
    {   // centre = inv(R) (-T)   (smpl_condition.py:213)
        const double a = R[0], b = R[1], cc = R[2], d = R[3], e = R[4], f = R[5], g = R[6], h = R[7], i = R[8];
        const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
        const double inv = 1.0 / (a * A + b * B + cc * C);
        const double I[9] = {A * inv, -(b * i - cc * h) * inv, (b * f - cc * e) * inv, B * inv, (a * i - cc * g) * inv, -(a * f - cc * d) * inv,
                             C * inv, -(a * h - b * g) * inv, (a * e - b * d) * inv};
        for (int r = 0; r < 3; r++) c[r] = -(I[3 * r] * T[0] + I[3 * r + 1] * T[1] + I[3 * r + 2] * T[2]);
    }
    const double p[3] = {(double)kps[3 * k], (double)kps[3 * k + 1], (double)kps[3 * k + 2]};
    double dir[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const double t_far = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    for (int i = 0; i < 3; i++) dir[i] = (double)(float)(dir[i] / t_far);     // the reference hands float32 rays to open3d
    double best = INFINITY;
    if (cull) {
        for (int f = tid; f < F; f += 256) {
            const float* a = verts + 3 * (size_t)tris[3 * f];
            const float* b = verts + 3 * (size_t)tris[3 * f + 1];
            const float* cc = verts + 3 * (size_t)tris[3 * f + 2];
            const double v0[3] = {(double)a[0], (double)a[1], (double)a[2]};
            const double e1[3] = {(double)b[0] - v0[0], (double)b[1] - v0[1], (double)b[2] - v0[2]};
            const double e2[3] = {(double)cc[0] - v0[0], (double)cc[1] - v0[1], (double)cc[2] - v0[2]};
            const double pv[3] = {dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0]};
            const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
            if (!(fabs(det) > 1e-12)) continue;
            const double inv = 1.0 / det;
            const double tv[3] = {c[0] - v0[0], c[1] - v0[1], c[2] - v0[2]};
            const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
            const double q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
            const double w = (q[0] * dir[0] + q[1] * dir[1] + q[2] * dir[2]) * inv;
            const double t = (q[0] * e2[0] + q[1] * e2[1] + q[2] * e2[2]) * inv;
            if (u >= 0.0 && w >= 0.0 && u + w <= 1.0 && t > 0.0 && t < best) best = t;
        }
    }
    s_min[tid] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_min[tid] = fmin(s_min[tid], s_min[tid + s]);
        __syncthreads();
    }
    if (tid != 0) return;
    const double t_hit = s_min[0];
    // world -> camera -> image (smpl_condition.py:205-212)
    double cam[3];
    for (int r = 0; r < 3; r++) cam[r] = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2] + T[r];
    bool valid = !(cam[2] < 0.0);
    double h[3];
    for (int r = 0; r < 3; r++) h[r] = (double)intr[3 * r] * cam[0] + (double)intr[3 * r + 1] * cam[1] + (double)intr[3 * r + 2] * cam[2];
    const double x = h[0] / h[2], y = h[1] / h[2];
    double dist = -1.0;
    if (cull) {
        const int g = groups[k];
        const double thr = g == 0 ? (double)thr_body : (g == 1 ? (double)thr_hand : (double)thr_face);
        if ((t_far - t_hit) > thr) valid = false;
        dist = t_far;
    }
    const double W = (double)intr[2] * 2.0, H = (double)intr[5] * 2.0;       // to_controlnet_pose :27
    const double xn = x / W, yn = y / H;
    if (!(isfinite(xn) && isfinite(yn))) valid = false;
    rows[4 * k] = valid ? xn : 0.0;
    rows[4 * k + 1] = valid ? yn : 0.0;
    rows[4 * k + 2] = dist;
    rows[4 * k + 3] = valid ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// limb polygons -> per-row spans
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long floor_div(long long a, long long b) {      // b > 0
    long long q = a / b;
    return (a % b < 0) ? q - 1 : q;
}
__device__ __forceinline__ double rotate_x(double cx, double x, double alpha, double y, double beta) {
#pragma clang fp contract(off)
    return cx + x * alpha - y * beta;
}
__device__ __forceinline__ double rotate_y(double cy, double x, double beta, double y, double alpha) {
#pragma clang fp contract(off)
    return cy + x * beta + y * alpha;
}
__device__ __forceinline__ int trunc_to_int(double v) {        // Python int(): toward zero; far-away values clamped (cannot touch the canvas)
    if (!(v == v)) return 0;
    if (v > 1048576.0) return 1048576;
    if (v < -1048576.0) return -1048576;
    return (int)v;
}

__global__ __launch_bounds__(256) void k_cond_spans(int H, int W, const double* __restrict__ rows, int flip, int stickwidth,
                                                    int* __restrict__ spans /*[17][H][2]*/) {
    __shared__ int vx[NVERT], vy[NVERT];
    __shared__ int s_ok;
    const int l = blockIdx.x, tid = threadIdx.x;
    int i1 = c_limb[l][0] - 1, i2 = c_limb[l][1] - 1;
    if (flip) { i1 = c_flip[i1]; i2 = c_flip[i2]; }
    if (tid == 0) s_ok = (rows[4 * i1 + 3] > 0.5 && rows[4 * i2 + 3] > 0.5) ? 1 : 0;
    __syncthreads();
    if (!s_ok) {
        for (int y = tid; y < H; y += 256) { spans[((size_t)l * H + y) * 2] = 1; spans[((size_t)l * H + y) * 2 + 1] = 0; }
        return;
    }
    {   // open_pose.py:125-133 in fp64
        const double Y0 = rows[4 * i1] * (double)W, Y1 = rows[4 * i2] * (double)W;
        const double X0 = rows[4 * i1 + 1] * (double)H, X1 = rows[4 * i2 + 1] * (double)H;
        const double mX = (X0 + X1) / 2.0, mY = (Y0 + Y1) / 2.0;
        const double length = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1));
        const double angle_deg = atan2(X0 - X1, Y0 - Y1) * (180.0 / 3.14159265358979323846);
        const int cx = trunc_to_int(mY), cy = trunc_to_int(mX), a = trunc_to_int(length / 2.0), b = stickwidth;
        int ang = trunc_to_int(angle_deg);
        while (ang < 0) ang += 360;
        while (ang > 360) ang -= 360;
        const double alpha = (double)c_sin_deg[450 - ang], beta = (double)c_sin_deg[ang];
        for (int i = tid; i < NVERT; i += 256) {
            const double x = (double)a * (double)c_sin_deg[450 - i], y = (double)b * (double)c_sin_deg[i];
            vx[i] = (int)__builtin_rint(rotate_x((double)cx, x, alpha, y, beta));
            vy[i] = (int)__builtin_rint(rotate_y((double)cy, x, beta, y, alpha));
        }
    }
    __syncthreads();
    for (int y = tid; y < H; y += 256) {
        long long lo = 1, hi = 0;
        bool any = false;
        for (int i = 0; i < NVERT; i++) {
            const int j = (i + 1 == NVERT) ? 0 : i + 1;
            const long long x1 = vx[i], y1 = vy[i], x2 = vx[j], y2 = vy[j];
            long long ca, cb;
            if (y1 == y2) {
                if (y1 != y) continue;
                ca = x1 < x2 ? x1 : x2; cb = x1 < x2 ? x2 : x1;
            } else {
                if ((y < y1 && y < y2) || (y > y1 && y > y2)) continue;
                long long den = y2 - y1, num = x1 * (y2 - y1) + (x2 - x1) * ((long long)y - y1);
                if (den < 0) { den = -den; num = -num; }
                ca = cb = floor_div(2 * num + den, 2 * den);          // round(x) = floor(x + 1/2)
            }
            if (!any) { lo = ca; hi = cb; any = true; }
            else { lo = ca < lo ? ca : lo; hi = cb > hi ? cb : hi; }
        }
        if (!any) { lo = 1; hi = 0; }
        if (lo < -2147483000LL) lo = -2147483000LL;
        if (hi > 2147483000LL) hi = 2147483000LL;
        spans[((size_t)l * H + y) * 2] = (int)lo; spans[((size_t)l * H + y) * 2 + 1] = (int)hi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the image: one thread per pixel, primitives in the reference's draw order
// ---------------------------------------------------------------------------------------------------------------------
struct DrawSizes { int body_radius, stickwidth, hand_radius, hand_thickness, face_radius; };

__device__ __forceinline__ bool in_circle(int dx, int dy, int radius) {
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    if (dy > radius) return false;
    return dx <= (int)c_circle_hw[radius][dy];
}
__device__ __forceinline__ bool near_segment(int px, int py, int x1, int y1, int x2, int y2, int thickness) {
    const long long abx = x2 - x1, aby = y2 - y1, apx = px - x1, apy = py - y1;
    const long long L2 = abx * abx + aby * aby, tn = apx * abx + apy * aby, t2 = (long long)thickness * thickness;
    if (tn <= 0) return 4 * (apx * apx + apy * apy) <= t2;
    if (tn >= L2) { const long long bx = px - x2, by = py - y2; return 4 * (bx * bx + by * by) <= t2; }
    return 4 * ((apx * apx + apy * apy) * L2 - tn * tn) <= t2 * L2;
}
__device__ __forceinline__ int blend(int old, int col) {
#pragma clang fp contract(off)
    const float v = (float)old * 0.4f + (float)col * 0.6f;
    int r = (int)__builtin_rintf(v);
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}

__global__ __launch_bounds__(256) void k_cond_draw(int H, int W, const double* __restrict__ rows, int flags, DrawSizes sz,
                                                   const int* __restrict__ spans, unsigned char* __restrict__ out_u8,
                                                   float* __restrict__ out_chw) {
    __shared__ int kx[DWG_COND_KEYPOINTS], ky[DWG_COND_KEYPOINTS];
    __shared__ unsigned char kv[DWG_COND_KEYPOINTS];
    const int tid = threadIdx.x;
    if (tid < DWG_COND_KEYPOINTS) {
        const bool v = rows[4 * tid + 3] > 0.5;
        kv[tid] = v ? 1 : 0;
        kx[tid] = v ? trunc_to_int(rows[4 * tid] * (double)W) : 0;
        ky[tid] = v ? trunc_to_int(rows[4 * tid + 1] * (double)H) : 0;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= H * W) return;
    const int py = pix / W, px = pix - py * W;
    int r = 0, g = 0, b = 0;
    if (flags & DWG_COND_DRAW_BODY) {
        const bool flip = flags & DWG_COND_FLIP_LR;
        for (int i = 0; i < NB; i++) {
            const int k = flip ? c_flip[i] : i;
            if (kv[k] && kx[k] >= 1 && ky[k] >= 1 && in_circle(px - kx[k], py - ky[k], sz.body_radius)) {
                r = c_body_rgb[i][0]; g = c_body_rgb[i][1]; b = c_body_rgb[i][2];
            }
        }
        for (int l = 0; l < NLIMB; l++) {
            const int lo = spans[((size_t)l * H + py) * 2], hi = spans[((size_t)l * H + py) * 2 + 1];
            if (px >= lo && px <= hi) { r = blend(r, c_body_rgb[l][0]); g = blend(g, c_body_rgb[l][1]); b = blend(b, c_body_rgb[l][2]); }
        }
    }
    if (flags & DWG_COND_DRAW_HAND) {
        for (int hnd = 0; hnd < 2; hnd++) {
            const int base = NB + hnd * NH;
            for (int i = 0; i < NH; i++) {
                const int k = base + i;
                if (kv[k] && kx[k] >= 1 && ky[k] >= 1 && in_circle(px - kx[k], py - ky[k], sz.hand_radius)) { r = 0; g = 0; b = 255; }
            }
            for (int e = 0; e < NEDGE; e++) {
                const int k1 = base + c_hand_edge[e][0], k2 = base + c_hand_edge[e][1];
                if (!(kv[k1] && kv[k2])) continue;
                if (!(kx[k1] >= 1 && ky[k1] >= 1 && kx[k2] >= 1 && ky[k2] >= 1)) continue;
                if (near_segment(px, py, kx[k1], ky[k1], kx[k2], ky[k2], sz.hand_thickness)) {
                    r = c_hand_edge_rgb[e][0]; g = c_hand_edge_rgb[e][1]; b = c_hand_edge_rgb[e][2];
                }
            }
        }
    }
    if (flags & DWG_COND_DRAW_FACE) {
        for (int i = 0; i < NF; i++) {
            const int k = NB + 2 * NH + i;
            if (kv[k] && kx[k] >= 1 && ky[k] >= 1 && in_circle(px - kx[k], py - ky[k], sz.face_radius)) { r = 255; g = 255; b = 255; }
        }
    }
    if (out_u8) { out_u8[3 * (size_t)pix] = (unsigned char)r; out_u8[3 * (size_t)pix + 1] = (unsigned char)g; out_u8[3 * (size_t)pix + 2] = (unsigned char)b; }
    if (out_chw) {
        const size_t P = (size_t)H * W;
        out_chw[pix] = (float)r / 255.f; out_chw[P + pix] = (float)g / 255.f; out_chw[2 * P + pix] = (float)b / 255.f;
    }
}

DrawSizes draw_sizes(int H, int W) {       // open_pose.py:303-315
    int s[5] = {4, 4, 4, 2, 3};
    if (H != 512 || W != 512) {
        const double r = (H + W) / 2.0 / 512.0;
        for (int i = 0; i < 5; i++) { int v = (int)(s[i] * r); s[i] = v > 1 ? v : 1; }
    }
    return DrawSizes{s[0], s[1], s[2], s[3], s[4]};
}

}  // namespace

extern "C" {

int dwg_condition_keypoints(int32_t K, const float* keypoints, const float* extrinsic, const float* intrinsics, int32_t V,
                            const float* vertices, int32_t F, const int32_t* triangles, const uint8_t* groups, float thres_body,
                            float thres_hand, float thres_face, int32_t use_occlusion_culling, double* rows, dwg_stream_t stream) {
    if (K <= 0 || !keypoints || !extrinsic || !intrinsics || !rows) return DWG_E_ARG;
    if (use_occlusion_culling && (V <= 0 || F <= 0 || !vertices || !triangles || !groups)) return DWG_E_ARG;
    DWG_LAUNCH("cond_keypoints", k_cond_keypoints, dim3(K), dim3(256), 0, (hipStream_t)stream, K, keypoints, extrinsic, intrinsics,
               use_occlusion_culling ? F : 0, vertices, triangles, groups, thres_body, thres_hand, thres_face,
               use_occlusion_culling ? 1 : 0, rows);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

size_t dwg_condition_workspace_bytes(int32_t H, int32_t W) {
    (void)W;
    return H > 0 ? (size_t)NLIMB * H * 2 * sizeof(int) : 0;
}

int dwg_condition_draw(int32_t H, int32_t W, const double* rows, int32_t flags, uint8_t* out_u8, float* out_chw, void* workspace,
                       dwg_stream_t stream) {
    if (H <= 0 || W <= 0 || H > 16384 || W > 16384 || !rows || !workspace || (!out_u8 && !out_chw)) return DWG_E_ARG;
    const DrawSizes sz = draw_sizes(H, W);
    if (sz.body_radius > DWG_COND_MAX_RADIUS || sz.hand_radius > DWG_COND_MAX_RADIUS || sz.face_radius > DWG_COND_MAX_RADIUS) return DWG_E_ARG;
    int* spans = reinterpret_cast<int*>(workspace);
    if (flags & DWG_COND_DRAW_BODY)
        DWG_LAUNCH("cond_spans", k_cond_spans, dim3(NLIMB), dim3(256), 0, (hipStream_t)stream, H, W, rows, (flags & DWG_COND_FLIP_LR) ? 1 : 0,
                   sz.stickwidth, spans);
    DWG_LAUNCH("cond_draw", k_cond_draw, dim3(dwg_cdiv(H * W, 256)), dim3(256), 0, (hipStream_t)stream, H, W, rows, flags, sz,
               (const int*)spans, out_u8, out_chw);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
