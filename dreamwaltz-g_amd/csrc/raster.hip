// raster.hip -- differentiable tile-based 3D-Gaussian-splat rasterizer for gfx950 (MI355X).
//
// Replaces the third-party CUDA extension the reference calls at
//   /root/reference/core/gaussian/gaussian_renderer.py:186-195 (forward) and its autograd backward
// (boundary B1).  Written from the algorithm, not from the CUDA sources:
//
//   stage A  k_preprocess      1 thread / Gaussian: project, EWA covariance, 3-sigma radius, tile rect,
//                              48-byte splat record (3 x float4, 16-B aligned gathers), per-tile histogram
//            k_scan_tiles      one workgroup: exclusive scan of the <= few-thousand tile counters
//   stage B  k_scatter         1 thread / Gaussian: (depth|id) 64-bit key into its tiles' segments
//            k_tile_sort       1 workgroup / tile: ascending-only bitonic network on the tile's keys in LDS
//                              (no global radix sort: keys never leave the chip between read and write)
//            k_render_fwd      1 workgroup (4 x wave64) / 16x16 tile, splat records staged through LDS
//   backward k_render_bwd      back-to-front replay; per-splat partial gradients are reduced across the
//                              wavefront with DPP adds, across the 4 waves in LDS, then ONE global atomic per
//                              (splat,tile,component)
//            k_preprocess_bwd  1 thread / Gaussian: conic -> cov2D -> cov3D -> (scale, quaternion), mean chain
//
// Compositing order inside a tile: ascending (depth bits, Gaussian id) -- same tie-break as a stable sort.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_raster.h"

namespace {

struct Params {
    int G, H, W, tiles_x, tiles_y;
    float tanfovx, tanfovy, focal_x, focal_y, scale_mod;
    int sh_degree, sh_coeffs;
    const float* bg;
    const float* view;
    const float* proj;
    const float* campos;
};

struct GeomLayout {
    size_t header, rec0, rec1, rec2, rect, tile_count, tile_start, tile_cursor, total;
};

static GeomLayout geom_layout(int G, int H, int W) {
    GeomLayout L;
    size_t T = (size_t)dwg_cdiv(W, DWG_TILE) * dwg_cdiv(H, DWG_TILE);
    size_t g = (size_t)(G > 0 ? G : 1);
    size_t o = 0;
    L.header = o; o += 256;
    L.rec0 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rec1 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rec2 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rect = o; o = dwg_align_up(o + g * sizeof(uint2), 256);
    L.tile_count = o; o = dwg_align_up(o + T * 4, 256);
    L.tile_cursor = o; o = dwg_align_up(o + T * 4, 256);
    L.tile_start = o; o = dwg_align_up(o + (T + 1) * 4, 256);
    L.total = o;
    return L;
}

struct PairLayout { size_t keys, sorted, total; };
static PairLayout pair_layout(int64_t cap) {
    PairLayout L; size_t c = (size_t)(cap > 0 ? cap : 1);
    L.keys = 0; L.sorted = dwg_align_up(c * 8, 256); L.total = dwg_align_up(L.sorted + c * 4, 256);
    return L;
}
struct ImageLayout { size_t final_T, n_contrib, total; };
static ImageLayout image_layout(int H, int W) {
    ImageLayout L; size_t P = (size_t)H * W;
    L.final_T = 0; L.n_contrib = dwg_align_up(P * 4, 256); L.total = dwg_align_up(L.n_contrib + P * 4, 256);
    return L;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float3 xform43(const float* m, float3 p) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform44(const float* m, float3 p) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// rotation of an UN-normalised real-first quaternion (SURVEY checklist Q2), row-major
__device__ __forceinline__ void quat_to_R(float4 q, float R[9]) {
    float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}
__device__ __forceinline__ void cov3d_of(const float* scales, const float* rots, const float* cov3Dp, int i,
                                         float mod, float c6[6]) {
    if (cov3Dp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov3Dp[6 * i + k];
        return;
    }
    float R[9];
    quat_to_R(make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]), R);
    float s0 = mod * scales[3 * i], s1 = mod * scales[3 * i + 1], s2 = mod * scales[3 * i + 2];
    float M[9] = {R[0] * s0, R[1] * s1, R[2] * s2, R[3] * s0, R[4] * s1, R[5] * s2, R[6] * s0, R[7] * s1, R[8] * s2};
    c6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    c6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    c6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    c6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    c6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    c6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// EWA projection pieces shared by forward and backward
struct Ewa {
    float M[6];   // J * Wr (2x3)
    float MS[6];  // M * Sigma3D
    float a, b, c;  // cov2D (with +0.3 low-pass)
    float tx, ty, tz, xmul, ymul;
};
__device__ __forceinline__ Ewa ewa_project(const Params& p, const float* view, float3 pv, const float c6[6]) {
    Ewa e;
    float limx = 1.3f * p.tanfovx, limy = 1.3f * p.tanfovy;
    float txtz = pv.x / pv.z, tytz = pv.y / pv.z;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv.z;
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv.z;
    e.tz = pv.z;
    float J0 = p.focal_x / e.tz, J2 = -(p.focal_x * e.tx) / (e.tz * e.tz);
    float J4 = p.focal_y / e.tz, J5 = -(p.focal_y * e.ty) / (e.tz * e.tz);
    // Wr[k][c] = view[4*c + k]
#pragma unroll
    for (int c = 0; c < 3; c++) {
        e.M[c] = J0 * view[4 * c + 0] + J2 * view[4 * c + 2];
        e.M[3 + c] = J4 * view[4 * c + 1] + J5 * view[4 * c + 2];
    }
    float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            e.MS[r * 3 + c] = e.M[r * 3] * S[c] + e.M[r * 3 + 1] * S[3 + c] + e.M[r * 3 + 2] * S[6 + c];
    e.a = e.MS[0] * e.M[0] + e.MS[1] * e.M[1] + e.MS[2] * e.M[2] + 0.3f;
    e.b = e.MS[0] * e.M[3] + e.MS[1] * e.M[4] + e.MS[2] * e.M[5];
    e.c = e.MS[3] * e.M[3] + e.MS[4] * e.M[4] + e.MS[5] * e.M[5] + 0.3f;
    return e;
}

// Bit-identical evaluation of the splat exponent in forward and backward (explicit roundings).
__device__ __forceinline__ float splat_power(float ca, float cb, float cc, float dx, float dy) {
    float q = __fmul_rn(ca, __fmul_rn(dx, dx));
    q = __fmaf_rn(cc, __fmul_rn(dy, dy), q);
    float b = __fmul_rn(cb, __fmul_rn(dx, dy));
    return __fmaf_rn(-0.5f, q, -b);
}

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f};

// SH colour (reference formula: core/gaussian/spherical_harmonics.py:117-172, clamp gaussian_utils.py:16)
__device__ float3 sh_color(int deg, int M, const float* sh, float3 pos, const float* campos, unsigned* clampbits) {
    float3 d = make_float3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    float inv = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    float x = d.x * inv, y = d.y * inv, z = d.z * inv;
    float out[3];
    unsigned bits = 0;
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float r = kSH_C0 * SHC(0);
        if (deg > 0) {
            r = r - kSH_C1 * y * SHC(1) + kSH_C1 * z * SHC(2) - kSH_C1 * x * SHC(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + kSH_C2[0] * xy * SHC(4) + kSH_C2[1] * yz * SHC(5) + kSH_C2[2] * (2.f * zz - xx - yy) * SHC(6) +
                    kSH_C2[3] * xz * SHC(7) + kSH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + kSH_C3[0] * y * (3.f * xx - yy) * SHC(9) + kSH_C3[1] * xy * z * SHC(10) +
                        kSH_C3[2] * y * (4.f * zz - xx - yy) * SHC(11) + kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SHC(12) +
                        kSH_C3[4] * x * (4.f * zz - xx - yy) * SHC(13) + kSH_C3[5] * z * (xx - yy) * SHC(14) +
                        kSH_C3[6] * x * (xx - 3.f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        if (r < 0.f) { bits |= 1u << c; r = 0.f; }
        out[c] = r;
    }
    (void)M;
    *clampbits = bits;
    return make_float3(out[0], out[1], out[2]);
}

// ------------------------------------------------------------------------------------------------
// stage A
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_preprocess(Params p, const float* __restrict__ means3D,
                                                    const float* __restrict__ shs, const float* __restrict__ colors,
                                                    const float* __restrict__ opac, const float* __restrict__ scales,
                                                    const float* __restrict__ rots, const float* __restrict__ cov3Dp,
                                                    int* __restrict__ radii, float4* __restrict__ rec0,
                                                    float4* __restrict__ rec1, float4* __restrict__ rec2,
                                                    uint2* __restrict__ rect, uint32_t* __restrict__ tile_count, int use_lds_hist) {
    __shared__ float cam[32];
    extern __shared__ uint32_t hist[];      // [T] block-private tile histogram (hot tiles: one global atomic per block, not per splat)
    const int T = p.tiles_x * p.tiles_y;
    if (threadIdx.x < 16) cam[threadIdx.x] = p.view[threadIdx.x];
    else if (threadIdx.x < 32) cam[threadIdx.x] = p.proj[threadIdx.x - 16];
    if (use_lds_hist) for (int t = threadIdx.x; t < T; t += 256) hist[t] = 0u;
    __syncthreads();
    int i = blockIdx.x * 256 + threadIdx.x;
    const bool live_thread = i < p.G;
    if (!live_thread) i = 0;
    const float* view = cam; const float* proj = cam + 16;
    float3 pos = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    int radius = 0;
    uint2 rc = make_uint2(0u, 0u);
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    float3 pv = xform43(view, pos);
    if (live_thread && pv.z > 0.2f) {
        float4 ph = xform44(proj, pos);
        float pw = 1.f / (ph.w + 1e-7f);
        float ndcx = ph.x * pw, ndcy = ph.y * pw;
        float c6[6];
        cov3d_of(scales, rots, cov3Dp, i, p.scale_mod, c6);
        Ewa e = ewa_project(p, view, pv, c6);
        float det = e.a * e.c - e.b * e.b;
        if (det != 0.f) {
            float di = 1.f / det;
            float mid = 0.5f * (e.a + e.c);
            float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            float lm = fmaxf(mid + sq, mid - sq);
            int rad = (int)ceilf(3.f * sqrtf(lm));
            float px = ((ndcx + 1.f) * p.W - 1.f) * 0.5f;
            float py = ((ndcy + 1.f) * p.H - 1.f) * 0.5f;
            int tx0 = min(p.tiles_x, max(0, (int)((px - rad) / DWG_TILE)));
            int ty0 = min(p.tiles_y, max(0, (int)((py - rad) / DWG_TILE)));
            int tx1 = min(p.tiles_x, max(0, (int)((px + rad + DWG_TILE - 1) / DWG_TILE)));
            int ty1 = min(p.tiles_y, max(0, (int)((py + rad + DWG_TILE - 1) / DWG_TILE)));
            if ((tx1 - tx0) * (ty1 - ty0) > 0) {
                radius = rad;
                rc = make_uint2((unsigned)tx0 | ((unsigned)ty0 << 16), (unsigned)tx1 | ((unsigned)ty1 << 16));
                float3 col; unsigned cb = 0;
                if (colors) col = make_float3(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2]);
                else col = sh_color(p.sh_degree, p.sh_coeffs, shs + (size_t)i * p.sh_coeffs * 3, pos, p.campos, &cb);
                r0 = make_float4(px, py, pv.z, opac[i]);
                r1 = make_float4(e.c * di, -e.b * di, e.a * di, 0.f);
                r2 = make_float4(col.x, col.y, col.z, __uint_as_float(cb));
                for (int ty = ty0; ty < ty1; ty++)
                    for (int tx = tx0; tx < tx1; tx++) {
                        if (use_lds_hist) atomicAdd(&hist[ty * p.tiles_x + tx], 1u);
                        else atomicAdd(&tile_count[ty * p.tiles_x + tx], 1u);
                    }
            }
        }
    }
    if (live_thread) {
        radii[i] = radius;
        rect[i] = rc;
        rec0[i] = r0; rec1[i] = r1; rec2[i] = r2;
    }
    if (use_lds_hist) {
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += 256) { uint32_t c = hist[t]; if (c) atomicAdd(&tile_count[t], c); }
    }
}

// one workgroup of 1024 threads: exclusive scan of T tile counters
__global__ __launch_bounds__(1024) void k_scan_tiles(int T, const uint32_t* __restrict__ tile_count,
                                                     uint32_t* __restrict__ tile_start, int32_t* __restrict__ header) {
    __shared__ uint32_t part[1024];
    int tid = threadIdx.x;
    int chunk = (T + 1023) / 1024;
    int lo = tid * chunk, hi = min(T, lo + chunk);
    uint32_t s = 0;
    for (int t = lo; t < hi; t++) s += tile_count[t];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;  // exclusive prefix of this chunk
    for (int t = lo; t < hi; t++) { tile_start[t] = run; run += tile_count[t]; }
    if (tid == 1023) { tile_start[T] = part[1023]; header[0] = (int32_t)part[1023]; header[1] = 0; }
}

// ------------------------------------------------------------------------------------------------
// stage B
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter(int G, int tiles_x, int T, const float4* __restrict__ rec0,
                                                 const uint2* __restrict__ rect, const uint32_t* __restrict__ tile_start,
                                                 uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys,
                                                 int64_t cap, int32_t* __restrict__ header, int use_lds) {
    // Two sweeps over this block's splats: (1) count per tile in LDS, reserve ONE contiguous range per (block, tile) with a
    // single returning global atomic; (2) hand out slots inside the reserved ranges with LDS atomics.
    extern __shared__ uint32_t sm[];        // [T] counts / running local rank, [T] reserved base
    uint32_t* cnt = sm; uint32_t* base = sm + T;
    int i = blockIdx.x * 256 + threadIdx.x;
    int tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
    uint64_t key = 0;
    if (i < G) {
        uint2 rc = rect[i];
        tx0 = rc.x & 0xffff; ty0 = rc.x >> 16; tx1 = rc.y & 0xffff; ty1 = rc.y >> 16;
        key = ((uint64_t)__float_as_uint(rec0[i].z) << 32) | (uint32_t)i;
    }
    if (!use_lds) {
        for (int ty = ty0; ty < ty1; ty++)
            for (int tx = tx0; tx < tx1; tx++) {
                int t = ty * tiles_x + tx;
                int64_t slot = (int64_t)tile_start[t] + atomicAdd(&tile_cursor[t], 1u);
                if (slot < cap) keys[slot] = key; else header[1] = 1;
            }
        return;
    }
    for (int t = threadIdx.x; t < T; t += 256) cnt[t] = 0u;
    __syncthreads();
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = tx0; tx < tx1; tx++) atomicAdd(&cnt[ty * tiles_x + tx], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        uint32_t c = cnt[t];
        if (c) { base[t] = tile_start[t] + atomicAdd(&tile_cursor[t], c); cnt[t] = 0u; }
    }
    __syncthreads();
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = tx0; tx < tx1; tx++) {
            int t = ty * tiles_x + tx;
            int64_t slot = (int64_t)base[t] + atomicAdd(&cnt[t], 1u);
            if (slot < cap) keys[slot] = key; else header[1] = 1;
        }
}

// Ascending-only bitonic network (mirror step + half-cleaners) so that virtual +inf padding above n is legal.
template <typename Mem>
__device__ __forceinline__ void bitonic_network(Mem& m, int n, int npad) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int size = 2; size <= npad; size <<= 1) {
        int half = size >> 1;
        for (int t = tid; t < (npad >> 1); t += nthr) {
            int blk = t / half, r = t - blk * half;
            int i = blk * size + r, j = blk * size + (size - 1 - r);
            if (j < n) { uint64_t a = m.get(i), b = m.get(j); if (b < a) { m.set(i, b); m.set(j, a); } }
        }
        __syncthreads();
        for (int stride = half >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < (npad >> 1); t += nthr) {
                int blk = t / stride, r = t - blk * stride;
                int i = blk * 2 * stride + r, j = i + stride;
                if (j < n) { uint64_t a = m.get(i), b = m.get(j); if (b < a) { m.set(i, b); m.set(j, a); } }
            }
            __syncthreads();
        }
    }
}
struct LdsMem { uint64_t* p; __device__ uint64_t get(int i) const { return p[i]; } __device__ void set(int i, uint64_t v) { p[i] = v; } };
struct GlbMem { volatile uint64_t* p; __device__ uint64_t get(int i) const { return p[i]; } __device__ void set(int i, uint64_t v) { p[i] = v; } };

// One workgroup per tile. Handles tiles whose pair count n satisfies lo < n <= CAP (LDS) or n > lo (GLOBAL).
template <int CAP, bool GLOBAL>
__global__ __launch_bounds__(256) void k_tile_sort(const uint32_t* __restrict__ tile_start, uint64_t* __restrict__ keys,
                                                   uint32_t* __restrict__ sorted, int lo, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int tile = blockIdx.x;
    int64_t s = tile_start[tile], e = tile_start[tile + 1];
    if (s > cap) s = cap; if (e > cap) e = cap;
    int n = (int)(e - s);
    if (n <= lo) return;
    if (!GLOBAL && n > CAP) return;
    int npad = 2; while (npad < n) npad <<= 1;
    if (!GLOBAL) {
        uint64_t* sk = reinterpret_cast<uint64_t*>(smem_raw);
        for (int i = threadIdx.x; i < n; i += 256) sk[i] = keys[s + i];
        __syncthreads();
        LdsMem m{sk};
        bitonic_network(m, n, npad);
        for (int i = threadIdx.x; i < n; i += 256) sorted[s + i] = (uint32_t)sk[i];
    } else {
        GlbMem m{keys + s};
        __syncthreads();
        bitonic_network(m, n, npad);
        for (int i = threadIdx.x; i < n; i += 256) sorted[s + i] = (uint32_t)m.get(i);
    }
}

__global__ __launch_bounds__(256) void k_render_fwd(Params p, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ sorted, const float4* __restrict__ rec0,
                                                    const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                                                    int64_t cap, float* __restrict__ final_T, int* __restrict__ n_contrib,
                                                    float* __restrict__ out_color, float* __restrict__ out_depth,
                                                    float* __restrict__ out_alpha) {
    __shared__ float4 s0[256], s1[256], s2[256];
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int tid = threadIdx.x;
    const int px = tx * DWG_TILE + (tid & 15), py = ty * DWG_TILE + (tid >> 4);
    const bool inside = px < p.W && py < p.H;
    int64_t rs = tile_start[tile], re = tile_start[tile + 1];
    if (rs > cap) rs = cap; if (re > cap) re = cap;
    const int n = (int)(re - rs);
    const float fx = (float)px, fy = (float)py;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    int last = 0;
    bool done = !inside;
    for (int base = 0; base < n; base += 256) {
        if (__syncthreads_and(done)) break;
        int k = base + tid;
        if (k < n) {
            uint32_t g = sorted[rs + k];
            s0[tid] = rec0[g]; s1[tid] = rec1[g]; s2[tid] = rec2[g];
        }
        __syncthreads();
        int cnt = min(256, n - base);
        if (!done) {
            for (int j = 0; j < cnt; j++) {
                float4 a = s0[j]; float4 b = s1[j];
                float dx = a.x - fx, dy = a.y - fy;
                float power = splat_power(b.x, b.y, b.z, dx, dy);
                if (power > 0.f) continue;
                float alpha = fminf(0.99f, a.w * expf(power));
                if (alpha < (1.f / 255.f)) continue;
                float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) { done = true; break; }
                float w = alpha * T;
                float4 c = s2[j];
                C0 += c.x * w; C1 += c.y * w; C2 += c.z * w; D += a.z * w; A += w;
                T = test_T; last = base + j + 1;
            }
        }
    }
    if (inside) {
        size_t P = (size_t)p.H * p.W, pix = (size_t)py * p.W + px;
        final_T[pix] = T; n_contrib[pix] = last;
        out_color[pix] = C0 + T * p.bg[0];
        out_color[P + pix] = C1 + T * p.bg[1];
        out_color[2 * P + pix] = C2 + T * p.bg[2];
        out_depth[pix] = D; out_alpha[pix] = A;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
#define NGRAD 10  // g2d.x g2d.y | conic a, b(half), c | opacity | r g b | depth   (row stride 12 floats)
#define GSTRIDE 12

__global__ __launch_bounds__(256) void k_render_bwd(Params p, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ sorted, const float4* __restrict__ rec0,
                                                    const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                                                    int64_t cap, const float* __restrict__ final_T,
                                                    const int* __restrict__ n_contrib, const float* __restrict__ g_color,
                                                    const float* __restrict__ g_depth, const float* __restrict__ g_alpha,
                                                    float* __restrict__ gacc /* [G][GSTRIDE] */) {
    __shared__ float4 s0[256], s1[256], s2[256];
    __shared__ uint32_t sgid[256];
    __shared__ float acc[256 * NGRAD];
    __shared__ int smax;
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int px = tx * DWG_TILE + (tid & 15), py = ty * DWG_TILE + (tid >> 4);
    const bool inside = px < p.W && py < p.H;
    int64_t rs = tile_start[tile], re = tile_start[tile + 1];
    if (rs > cap) rs = cap; if (re > cap) re = cap;
    const float fx = (float)px, fy = (float)py;
    const size_t P = (size_t)p.H * p.W, pix = (size_t)py * p.W + px;
    const float T_final = inside ? final_T[pix] : 0.f;
    const int last = inside ? n_contrib[pix] : 0;
    float gp0 = 0.f, gp1 = 0.f, gp2 = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        gp0 = g_color[pix]; gp1 = g_color[P + pix]; gp2 = g_color[2 * P + pix];
        if (g_depth) gpd = g_depth[pix];
        if (g_alpha) gpa = g_alpha[pix];
    }
    const float bgdot = p.bg[0] * gp0 + p.bg[1] * gp1 + p.bg[2] * gp2;
    if (tid == 0) smax = 0;
    __syncthreads();
    atomicMax(&smax, last);
    __syncthreads();
    const int n = min((int)(re - rs), smax);  // nothing beyond the deepest contributor matters
    float T = T_final;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f, aa = 0.f;  // sum over later splats of value*w
    const float ddelx = 0.5f * p.W, ddely = 0.5f * p.H;
    for (int hi = n; hi > 0; hi -= 256) {
        const int lo = max(0, hi - 256), cnt = hi - lo;
        __syncthreads();  // previous flush done before acc/s* are reused
        if (tid < cnt) {
            uint32_t g = sorted[rs + (hi - 1 - tid)];
            sgid[tid] = g; s0[tid] = rec0[g]; s1[tid] = rec1[g]; s2[tid] = rec2[g];
        }
#pragma unroll
        for (int c = 0; c < NGRAD; c++) acc[c * 256 + tid] = 0.f;
        __syncthreads();
        for (int j = 0; j < cnt; j++) {
            const int pos = hi - 1 - j;
            bool valid = pos < last;
            float4 a = s0[j]; float4 b = s1[j];
            float dx = a.x - fx, dy = a.y - fy;
            float power = splat_power(b.x, b.y, b.z, dx, dy);
            float Gv = expf(power);
            float alpha = fminf(0.99f, a.w * Gv);
            valid = valid && (power <= 0.f) && (alpha >= (1.f / 255.f));
            if (!__any(valid)) continue;  // wave-uniform skip
            float v[NGRAD];
#pragma unroll
            for (int c = 0; c < NGRAD; c++) v[c] = 0.f;
            if (valid) {
                float4 col = s2[j];
                float inv1a = 1.f / (1.f - alpha);
                T = T * inv1a;
                float w = alpha * T;
                float dL_dalpha = (col.x * T - a0 * inv1a) * gp0 + (col.y * T - a1 * inv1a) * gp1 +
                                  (col.z * T - a2 * inv1a) * gp2 + (a.z * T - ad * inv1a) * gpd +
                                  (T - aa * inv1a) * gpa - T_final * inv1a * bgdot;
                a0 += col.x * w; a1 += col.y * w; a2 += col.z * w; ad += a.z * w; aa += w;
                float dL_dG = a.w * dL_dalpha;
                float gdx = b.x * dx + b.y * dy, gdy = b.z * dy + b.y * dx;
                v[0] = -dL_dG * Gv * gdx * ddelx;
                v[1] = -dL_dG * Gv * gdy * ddely;
                float h = -0.5f * Gv * dL_dG;
                v[2] = h * dx * dx; v[3] = h * dx * dy; v[4] = h * dy * dy;
                v[5] = Gv * dL_dalpha;
                v[6] = w * gp0; v[7] = w * gp1; v[8] = w * gp2; v[9] = w * gpd;
            }
#pragma unroll
            for (int c = 0; c < NGRAD; c++) v[c] = dwg_wave_sum_to_lane63(v[c]);
            if (lane == 63) {
#pragma unroll
                for (int c = 0; c < NGRAD; c++) atomicAdd(&acc[c * 256 + j], v[c]);
            }
        }
        __syncthreads();
        if (tid < cnt) {
            float* dst = gacc + (size_t)sgid[tid] * GSTRIDE;
#pragma unroll
            for (int c = 0; c < NGRAD; c++) {
                float x = acc[c * 256 + tid];
                if (x != 0.f) atomicAdd(dst + c, x);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_preprocess_bwd(Params p, const float* __restrict__ means3D,
                                                        const float* __restrict__ shs, const float* __restrict__ colors,
                                                        const float* __restrict__ scales, const float* __restrict__ rots,
                                                        const float* __restrict__ cov3Dp, const uint2* __restrict__ rect,
                                                        const float4* __restrict__ rec2, const float* __restrict__ gacc,
                                                        float* __restrict__ dmeans3D, float* __restrict__ dmeans2D,
                                                        float* __restrict__ dshs, float* __restrict__ dcolors,
                                                        float* __restrict__ dopac, float* __restrict__ dscales,
                                                        float* __restrict__ drots, float* __restrict__ dcov3D) {
    __shared__ float cam[32];
    if (threadIdx.x < 16) cam[threadIdx.x] = p.view[threadIdx.x];
    else if (threadIdx.x < 32) cam[threadIdx.x] = p.proj[threadIdx.x - 16];
    __syncthreads();
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.G) return;
    const float* view = cam; const float* proj = cam + 16;
    const float* g = gacc + (size_t)i * GSTRIDE;
    float gx = g[0], gy = g[1];
    float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f;
    float gc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint2 rc = rect[i];
    bool live = ((rc.y & 0xffff) > (rc.x & 0xffff)) && ((rc.y >> 16) > (rc.x >> 16));
    if (dmeans2D) { dmeans2D[3 * i] = gx; dmeans2D[3 * i + 1] = gy; dmeans2D[3 * i + 2] = 0.f; }
    if (dopac) dopac[i] = g[5];
    if (dcolors) { dcolors[3 * i] = g[6]; dcolors[3 * i + 1] = g[7]; dcolors[3 * i + 2] = g[8]; }
    float3 pos = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    if (live) {
        float c6[6];
        cov3d_of(scales, rots, cov3Dp, i, p.scale_mod, c6);
        float3 pv = xform43(view, pos);
        Ewa e = ewa_project(p, view, pv, c6);
        float det = e.a * e.c - e.b * e.b;
        float gA = g[2], gB = 2.f * g[3], gC = g[4];
        float d2 = 1.f / (det * det);
        float dL_da = d2 * (-e.c * e.c * gA + e.b * e.c * gB + (det - e.a * e.c) * gC);
        float dL_dc = d2 * (-e.a * e.a * gC + e.a * e.b * gB + (det - e.a * e.c) * gA);
        float dL_db = d2 * (2.f * e.b * e.c * gA - (det + 2.f * e.b * e.b) * gB + 2.f * e.a * e.b * gC);
        float Gm[4] = {dL_da, 0.5f * dL_db, 0.5f * dL_db, dL_dc};
        float dS[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                dS[r * 3 + c] = e.M[r] * (Gm[0] * e.M[c] + Gm[1] * e.M[3 + c]) + e.M[3 + r] * (Gm[2] * e.M[c] + Gm[3] * e.M[3 + c]);
        gc6[0] = dS[0]; gc6[3] = dS[4]; gc6[5] = dS[8];
        gc6[1] = dS[1] + dS[3]; gc6[2] = dS[2] + dS[6]; gc6[4] = dS[5] + dS[7];
        float dM[6];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            dM[c] = 2.f * (Gm[0] * e.MS[c] + Gm[1] * e.MS[3 + c]);
            dM[3 + c] = 2.f * (Gm[2] * e.MS[c] + Gm[3] * e.MS[3 + c]);
        }
        // dJ[u][k] = sum_c dM[u][c] * Wr[k][c],  Wr[k][c] = view[4*c + k]
        float dJ00 = dM[0] * view[0] + dM[1] * view[4] + dM[2] * view[8];
        float dJ02 = dM[0] * view[2] + dM[1] * view[6] + dM[2] * view[10];
        float dJ11 = dM[3] * view[1] + dM[4] * view[5] + dM[5] * view[9];
        float dJ12 = dM[3] * view[2] + dM[4] * view[6] + dM[5] * view[10];
        float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = e.xmul * (-p.focal_x * tz2 * dJ02);
        float dty = e.ymul * (-p.focal_y * tz2 * dJ12);
        float dtz = -p.focal_x * tz2 * dJ00 - p.focal_y * tz2 * dJ11 + (2.f * p.focal_x * e.tx) * tz3 * dJ02 +
                    (2.f * p.focal_y * e.ty) * tz3 * dJ12;
        dtz += g[9];  // depth = p_view.z
        gm0 += view[0] * dtx + view[1] * dty + view[2] * dtz;
        gm1 += view[4] * dtx + view[5] * dty + view[6] * dtz;
        gm2 += view[8] * dtx + view[9] * dty + view[10] * dtz;
        float4 ph = xform44(proj, pos);
        float mw = 1.f / (ph.w + 1e-7f);
        float mul1 = ph.x * mw * mw, mul2 = ph.y * mw * mw;
        gm0 += (proj[0] * mw - proj[3] * mul1) * gx + (proj[1] * mw - proj[3] * mul2) * gy;
        gm1 += (proj[4] * mw - proj[7] * mul1) * gx + (proj[5] * mw - proj[7] * mul2) * gy;
        gm2 += (proj[8] * mw - proj[11] * mul1) * gx + (proj[9] * mw - proj[11] * mul2) * gy;
    }
    // SH colour backward
    if (!colors && shs && dshs) {
        const int M = p.sh_coeffs, deg = p.sh_degree;
        float* gsh = dshs + (size_t)i * M * 3;
        for (int k = 0; k < M * 3; k++) gsh[k] = 0.f;
        if (live) {
            const float* sh = shs + (size_t)i * M * 3;
            unsigned cb = __float_as_uint(rec2[i].w);
            float3 d = make_float3(pos.x - p.campos[0], pos.y - p.campos[1], pos.z - p.campos[2]);
            float nrm = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
            float inv = 1.f / nrm;
            float x = d.x * inv, y = d.y * inv, z = d.z * inv;
            float gd0 = 0.f, gd1 = 0.f, gd2 = 0.f;
            for (int ch = 0; ch < 3; ch++) {
                float gr = ((cb >> ch) & 1u) ? 0.f : g[6 + ch];
#define SHC(k) sh[(k) * 3 + ch]
#define GSH(k, v) gsh[(k) * 3 + ch] = (v) * gr
                float ddx = 0.f, ddy = 0.f, ddz = 0.f;
                GSH(0, kSH_C0);
                if (deg > 0) {
                    GSH(1, -kSH_C1 * y); GSH(2, kSH_C1 * z); GSH(3, -kSH_C1 * x);
                    ddx = -kSH_C1 * SHC(3); ddy = -kSH_C1 * SHC(1); ddz = kSH_C1 * SHC(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        GSH(4, kSH_C2[0] * xy); GSH(5, kSH_C2[1] * yz); GSH(6, kSH_C2[2] * (2.f * zz - xx - yy));
                        GSH(7, kSH_C2[3] * xz); GSH(8, kSH_C2[4] * (xx - yy));
                        ddx += kSH_C2[0] * y * SHC(4) + kSH_C2[2] * 2.f * -x * SHC(6) + kSH_C2[3] * z * SHC(7) + kSH_C2[4] * 2.f * x * SHC(8);
                        ddy += kSH_C2[0] * x * SHC(4) + kSH_C2[1] * z * SHC(5) + kSH_C2[2] * 2.f * -y * SHC(6) + kSH_C2[4] * 2.f * -y * SHC(8);
                        ddz += kSH_C2[1] * y * SHC(5) + kSH_C2[2] * 4.f * z * SHC(6) + kSH_C2[3] * x * SHC(7);
                        if (deg > 2) {
                            GSH(9, kSH_C3[0] * y * (3.f * xx - yy)); GSH(10, kSH_C3[1] * xy * z);
                            GSH(11, kSH_C3[2] * y * (4.f * zz - xx - yy)); GSH(12, kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                            GSH(13, kSH_C3[4] * x * (4.f * zz - xx - yy)); GSH(14, kSH_C3[5] * z * (xx - yy));
                            GSH(15, kSH_C3[6] * x * (xx - 3.f * yy));
                            ddx += kSH_C3[0] * SHC(9) * 6.f * xy + kSH_C3[1] * SHC(10) * yz + kSH_C3[2] * SHC(11) * -2.f * xy +
                                   kSH_C3[3] * SHC(12) * -6.f * xz + kSH_C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                                   kSH_C3[5] * SHC(14) * 2.f * xz + kSH_C3[6] * SHC(15) * 3.f * (xx - yy);
                            ddy += kSH_C3[0] * SHC(9) * 3.f * (xx - yy) + kSH_C3[1] * SHC(10) * xz + kSH_C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) +
                                   kSH_C3[3] * SHC(12) * -6.f * yz + kSH_C3[4] * SHC(13) * -2.f * xy +
                                   kSH_C3[5] * SHC(14) * -2.f * yz + kSH_C3[6] * SHC(15) * -6.f * xy;
                            ddz += kSH_C3[1] * SHC(10) * xy + kSH_C3[2] * SHC(11) * 8.f * yz + kSH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) +
                                   kSH_C3[4] * SHC(13) * 8.f * xz + kSH_C3[5] * SHC(14) * (xx - yy);
                        }
                    }
                }
#undef SHC
#undef GSH
                gd0 += ddx * gr; gd1 += ddy * gr; gd2 += ddz * gr;
            }
            float inv3 = inv * inv * inv;
            float dot = d.x * gd0 + d.y * gd1 + d.z * gd2;
            gm0 += gd0 * inv - d.x * dot * inv3;
            gm1 += gd1 * inv - d.y * dot * inv3;
            gm2 += gd2 * inv - d.z * dot * inv3;
        }
    }
    dmeans3D[3 * i] = gm0; dmeans3D[3 * i + 1] = gm1; dmeans3D[3 * i + 2] = gm2;
    if (dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) dcov3D[6 * i + k] = gc6[k];
    }
    if (!cov3Dp && scales && rots) {
        float4 q = make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]);
        float R[9]; quat_to_R(q, R);
        float sv[3] = {p.scale_mod * scales[3 * i], p.scale_mod * scales[3 * i + 1], p.scale_mod * scales[3 * i + 2]};
        float dSig[9] = {gc6[0], 0.5f * gc6[1], 0.5f * gc6[2], 0.5f * gc6[1], gc6[3], 0.5f * gc6[4], 0.5f * gc6[2], 0.5f * gc6[4], gc6[5]};
        float dMm[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                dMm[r * 3 + c] = 2.f * (dSig[r * 3] * R[c] * sv[c] + dSig[r * 3 + 1] * R[3 + c] * sv[c] + dSig[r * 3 + 2] * R[6 + c] * sv[c]);
        if (dscales) {
#pragma unroll
            for (int c = 0; c < 3; c++)
                dscales[3 * i + c] = p.scale_mod * (R[c] * dMm[c] + R[3 + c] * dMm[3 + c] + R[6 + c] * dMm[6 + c]);
        }
        if (drots) {
            float dR[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) dR[r * 3 + c] = dMm[r * 3 + c] * sv[c];
            float r = q.x, x = q.y, y = q.z, z = q.w;
            drots[4 * i + 0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            drots[4 * i + 1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
            drots[4 * i + 2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
            drots[4 * i + 3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
}

static int make_params(const dwg_raster_settings* cfg, int G, Params* p) {
    if (!cfg || G < 0 || cfg->image_height <= 0 || cfg->image_width <= 0) return DWG_E_ARG;
    if (!cfg->bg || !cfg->viewmatrix || !cfg->projmatrix) return DWG_E_ARG;
    p->G = G; p->H = cfg->image_height; p->W = cfg->image_width;
    p->tiles_x = dwg_cdiv(p->W, DWG_TILE); p->tiles_y = dwg_cdiv(p->H, DWG_TILE);
    if (p->tiles_x > 0xffff || p->tiles_y > 0xffff) return DWG_E_ARG;
    p->tanfovx = cfg->tanfovx; p->tanfovy = cfg->tanfovy;
    p->focal_x = p->W / (2.f * cfg->tanfovx); p->focal_y = p->H / (2.f * cfg->tanfovy);
    p->scale_mod = cfg->scale_modifier;
    p->sh_degree = cfg->sh_degree; p->sh_coeffs = cfg->sh_coeffs;
    p->bg = cfg->bg; p->view = cfg->viewmatrix; p->proj = cfg->projmatrix; p->campos = cfg->campos;
    return DWG_OK;
}

}  // namespace

extern "C" {

int dwg_raster_workspace_sizes(int32_t G, int32_t H, int32_t W, int64_t pair_capacity, size_t* geom_bytes,
                               size_t* pairs_bytes, size_t* image_bytes) {
    if (G < 0 || H <= 0 || W <= 0 || pair_capacity < 0) return DWG_E_ARG;
    if (geom_bytes) *geom_bytes = geom_layout(G, H, W).total;
    if (pairs_bytes) *pairs_bytes = pair_layout(pair_capacity).total;
    if (image_bytes) *image_bytes = image_layout(H, W).total;
    return DWG_OK;
}

const int32_t* dwg_raster_num_pairs_ptr(const void* ws_geom) { return reinterpret_cast<const int32_t*>(ws_geom); }

int dwg_raster_forward_bin(const dwg_raster_settings* cfg, int32_t G, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, int32_t* radii, void* ws_geom,
                           dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, G, &p);
    if (rc) return rc;
    if (!ws_geom) return DWG_E_ARG;
    if (G > 0) {  // with G == 0 every per-Gaussian pointer may be NULL
        if (!means3D || !opacities || !radii) return DWG_E_ARG;
        if ((shs == nullptr) == (colors_precomp == nullptr)) return DWG_E_ARG;      // exactly one colour source
        if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) return DWG_E_ARG;
        if (shs && (!cfg->campos || cfg->sh_degree < 0 || cfg->sh_degree > 3 ||
                    cfg->sh_coeffs < (cfg->sh_degree + 1) * (cfg->sh_degree + 1))) return DWG_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    char* ws = (char*)ws_geom;
    int T = p.tiles_x * p.tiles_y;
    if (hipMemsetAsync(ws + L.tile_count, 0, L.tile_start - L.tile_count, stream) != hipSuccess) return DWG_E_LAUNCH;
    if (G > 0) {
        const int use_lds_hist = T <= 16384;
        DWG_LAUNCH("raster_preprocess", k_preprocess, dim3(dwg_cdiv(G, 256)), dim3(256), use_lds_hist ? (size_t)T * 4 : 0, stream, p,
                   means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, (float4*)(ws + L.rec0),
                   (float4*)(ws + L.rec1), (float4*)(ws + L.rec2), (uint2*)(ws + L.rect), (uint32_t*)(ws + L.tile_count),
                   use_lds_hist);
    }
    DWG_LAUNCH("raster_scan_tiles", k_scan_tiles, dim3(1), dim3(1024), 0, stream, T, (const uint32_t*)(ws + L.tile_count),
                       (uint32_t*)(ws + L.tile_start), (int32_t*)(ws + L.header));
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_forward_render(const dwg_raster_settings* cfg, int32_t G, void* ws_geom, void* ws_pairs,
                              int64_t pair_capacity, void* ws_image, float* out_color, float* out_depth,
                              float* out_alpha, dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, G, &p);
    if (rc) return rc;
    if (!ws_geom || !ws_pairs || !ws_image || !out_color || !out_depth || !out_alpha || pair_capacity < 0) return DWG_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    PairLayout PL = pair_layout(pair_capacity);
    ImageLayout IL = image_layout(p.H, p.W);
    char* ws = (char*)ws_geom; char* wp = (char*)ws_pairs; char* wi = (char*)ws_image;
    int T = p.tiles_x * p.tiles_y;
    uint64_t* keys = (uint64_t*)(wp + PL.keys);
    uint32_t* sorted = (uint32_t*)(wp + PL.sorted);
    const uint32_t* tile_start = (const uint32_t*)(ws + L.tile_start);
    if (G > 0) {
        const int use_lds = T <= 8192;
        DWG_LAUNCH("raster_scatter", k_scatter, dim3(dwg_cdiv(G, 256)), dim3(256), use_lds ? (size_t)T * 8 : 0, stream, G, p.tiles_x, T,
                   (const float4*)(ws + L.rec0), (const uint2*)(ws + L.rect), tile_start, (uint32_t*)(ws + L.tile_cursor), keys,
                   pair_capacity, (int32_t*)(ws + L.header), use_lds);
        // three size classes: (1,2048] in 16 KiB LDS, (2048,8192] in 64 KiB LDS, >8192 in global memory
        DWG_LAUNCH("raster_tile_sort", (k_tile_sort<2048, false>), dim3(T), dim3(256), 2048 * 8, stream, tile_start, keys, sorted, 0,
                           pair_capacity);
        DWG_LAUNCH("raster_tile_sort_l", (k_tile_sort<8192, false>), dim3(T), dim3(256), 8192 * 8, stream, tile_start, keys, sorted, 2048,
                           pair_capacity);
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_sort<16384, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                16384 * 8);
            attr_set = true;
        }
        DWG_LAUNCH("raster_tile_sort_xl", (k_tile_sort<16384, false>), dim3(T), dim3(256), 16384 * 8, stream, tile_start, keys, sorted,
                   8192, pair_capacity);
        DWG_LAUNCH("raster_tile_sort_g", (k_tile_sort<0, true>), dim3(T), dim3(256), 0, stream, tile_start, keys, sorted, 16384,
                   pair_capacity);
    }
    DWG_LAUNCH("raster_render_fwd", k_render_fwd, dim3(T), dim3(256), 0, stream, p, tile_start, (const uint32_t*)sorted,
                       (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1), (const float4*)(ws + L.rec2),
                       pair_capacity, (float*)(wi + IL.final_T), (int*)(wi + IL.n_contrib), out_color, out_depth,
                       out_alpha);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_backward(const dwg_raster_settings* cfg, int32_t G, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        const float* rotations, const float* cov3D_precomp, const void* ws_geom, const void* ws_pairs,
                        int64_t pair_capacity, const void* ws_image, void* ws_grad, const float* dL_dout_color,
                        const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans3D,
                        float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                        float* dL_dscales, float* dL_drotations, float* dL_dcov3D, dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, G, &p);
    if (rc) return rc;
    if (!ws_geom || !ws_pairs || !ws_image || !ws_grad || !dL_dout_color || !dL_dmeans3D || pair_capacity < 0)
        return DWG_E_ARG;
    (void)opacities;
    if (G == 0) return DWG_OK;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return DWG_E_ARG;
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) return DWG_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    PairLayout PL = pair_layout(pair_capacity);
    ImageLayout IL = image_layout(p.H, p.W);
    const char* ws = (const char*)ws_geom; const char* wp = (const char*)ws_pairs; const char* wi = (const char*)ws_image;
    int T = p.tiles_x * p.tiles_y;
    if (hipMemsetAsync(ws_grad, 0, (size_t)G * GSTRIDE * sizeof(float), stream) != hipSuccess) return DWG_E_LAUNCH;
    DWG_LAUNCH("raster_render_bwd", k_render_bwd, dim3(T), dim3(256), 0, stream, p, (const uint32_t*)(ws + L.tile_start),
                       (const uint32_t*)(wp + PL.sorted), (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1),
                       (const float4*)(ws + L.rec2), pair_capacity, (const float*)(wi + IL.final_T),
                       (const int*)(wi + IL.n_contrib), dL_dout_color, dL_dout_depth, dL_dout_alpha, (float*)ws_grad);
    DWG_LAUNCH("raster_preprocess_bwd", k_preprocess_bwd, dim3(dwg_cdiv(G, 256)), dim3(256), 0, stream, p, means3D, shs, colors_precomp,
                       scales, rotations, cov3D_precomp, (const uint2*)(ws + L.rect), (const float4*)(ws + L.rec2),
                       (const float*)ws_grad, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales,
                       dL_drotations, dL_dcov3D);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
