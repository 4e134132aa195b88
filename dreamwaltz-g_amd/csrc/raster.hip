// raster.hip -- differentiable tile-based 3D-Gaussian-splat rasterizer for gfx950 (MI355X).
//
// Replaces the third-party CUDA extension the reference calls at
//   /root/reference/core/gaussian/gaussian_renderer.py:186-195 (forward) and its autograd backward
// (boundary B1).  Written from the algorithm, not from the CUDA sources.
//
// Semantics are the reference's: a pixel composites, in (depth, id) order, every Gaussian whose 3-sigma square touches the
// pixel's 16x16 screen tile.  The WORK decomposition is not the reference's:
//
//   * one wave64 owns one 8x8 pixel block (a quarter of a reference tile): per-block lists are ~2x shorter than per-tile
//     lists, the forward needs no barrier, and the backward reduces per-splat partials inside ONE wave;
//   * exact pair culling: a (Gaussian, block) pair is only emitted if some pixel of the block can reach alpha >= 1/255
//     (minimum of the conic's quadratic form over the block rectangle against 2 ln(255 opacity), with a safety margin far
//     above fp32 rounding) -- pairs the per-pixel test would reject for all 64 pixels never exist, results are unchanged;
//   * round 5 -- SORT COARSE, SPLIT FINE: Gaussians are binned and depth-sorted per 32x32-pixel SUPERTILE (4x4 blocks; ~4x
//     fewer keys than (Gaussian, block) pairs; every 1024-key chunk of every list sorted by its own workgroup, the chunks of
//     a list merged by rank -- one binary search per sibling chunk and key, no merge tree, no workgroup owning a long list),
//     and the sorted list is split STABLY into its sixteen block lists (a block's list is a subsequence of its supertile's).  This replaced a per-block 64-bit sort of every pair (102 of the 290 us forward chain at 100 k
//     Gaussians / 512^2) and the per-pair LDS histogram / scatter atomics of the two binning kernels (39 + 63 us);
//   * the forward checkpoints the per-pixel compositing state every SEG splats; the backward then runs one wave per
//     (block, segment) FRONT-TO-BACK from its checkpoint -- perfectly balanced, no serial chain over a long list -- using
//     "sum over later splats" = (final sum) - (prefix sum);
//   * blocks are rendered longest-list-first (bucketed by log2 of the list length);
//   * every forward kernel runs on a (work, frames) grid: F frames (poses and / or cameras) per launch chain.
//
//   stage A  k_preprocess      1 thread / Gaussian, index order (coalesced): project, EWA covariance, 3-sigma radius, tile rect,
//                              48-byte splat record, exact pair count (scanline), supertile histogram (LDS-privatised)
//            k_scan_super      supertile starts + size classes; per-Gaussian pair-row offsets (goff); the frame's pair count
//   stage B  k_scatter_super   1 thread / Gaussian: (depth|id) 64-bit key into its supertiles' ranges
//            k_chunk_sort      1 workgroup / 1024 keys of a supertile's list: register-blocked bitonic network
//            k_rank_merge      1 thread / key: place in the supertile's sorted list by binary searches in the sibling chunks; block mask
//            k_split           1 workgroup / supertile: stable split into the sixteen block lists; list / segment / render-order pools
//            k_render_fwd      1 wave / block, records staged through LDS, next batch prefetched into registers
//   backward k_render_bwd      1 wave / (block, segment): pair-ordered partial rows, no atomics
//            k_gather_partials per-Gaussian sum of its rows
//            k_preprocess_bwd  1 thread / Gaussian: conic -> cov2D -> cov3D -> (scale, quaternion), mean chain
#include "dwg_common.h"
#include <atomic>
#include "dwg_prof_internal.h"
#include "../../include/dwg_raster.h"

namespace {

#define RT 16            // the reference's tile edge: decides WHICH Gaussians a pixel sees
#define BT 8             // pixel-block edge of this implementation (one wave64)
#ifndef ST
#define ST 4
#endif                   // blocks per supertile edge: Gaussians are binned and depth-sorted per 32x32-pixel supertile
#define SEG 128          // splats per backward segment / forward checkpoint interval
#define CHUNK 1024       // keys per sorting workgroup: a supertile's list is sorted in chunks, then rank-merged
#define NBUCKET 20       // render-order buckets (log2 of the list length)
#define GTILE 1024       // Gaussian indices per workgroup of the pair-row scan (k_scan_tiles)
#define IDBIN 64         // ... whose base is the sum of per-IDBIN-indices pair counts (fine bins: ~IDBIN integer atomics per address in k_preprocess;
                         // with one bin per GTILE indices the 1000 same-address atomics of a bin tripled that kernel's time)

struct Params {
    int G, H, W, tiles_x, tiles_y;      // tiles_* count 8x8 BLOCKS
    int rtiles_x, rtiles_y;             // reference 16x16 tiles
    float tanfovx, tanfovy, focal_x, focal_y, scale_mod;
    int sh_degree, sh_coeffs;
    const float* bg;
    const float* view;
    const float* proj;
    const float* campos;
    const int32_t* visit_order;         // permutation in which the binning stages walk the Gaussians (NULL: index order)
    const float* tanfov_dev;            // [tanfovx, tanfovy] in device memory (per frame at cam_stride) or NULL: the scalars above
    int stiles_x, stiles_y;             // supertiles (ST x ST blocks)
    // frames (blockIdx.y of the forward kernels): distance between consecutive frames'
    int64_t in_stride;                  // ... per-Gaussian input rows, in Gaussians (0: every frame reads the same rows)
    int64_t cam_stride;                 // ... viewmatrix / projmatrix / campos, in floats (0: one camera)
    size_t geom_stride, pairs_stride, image_stride;      // ... workspaces, in bytes
    int dbg;                            // DWG_RASTER_DEBUG bits (timing experiments; results are garbage): 1 no sort network, 2 no block masks, 4 no split
};

// header words of the geometry workspace
enum { H_K = 0, H_OVERFLOW = 1, H_KREF = 2, H_NSEG = 3 /* backward segments handed out so far */, H_NCHUNK = 4 /* sorting chunks of the frame */,
       H_TAG = 8, H_KS = 9 /* (Gaussian, supertile) pairs */, H_POOL = 10 /* running end of the block lists handed out so far */,
       H_BKT0 = 16 /* .. + NBUCKET: blocks per render-order bucket */ };

// Frame tags of the backward's pair-ordered partial rows (k_render_bwd / k_gather_partials) are drawn ON THE DEVICE, by the forward's scan
// kernel, from this counter: a tag chosen by the host at launch time is a kernel argument, and kernel arguments are frozen into a captured
// graph -- every replay of a captured step would then carry the SAME tag and rows left over from the previous replay would pass for this
// frame's (step_graph.py / player.py replay the launches, not the host code around them).
__device__ uint32_t g_frame_tag = 0x5eed0001u;

struct GeomLayout {
    size_t header, rec0, rec1, rec2, rect, npairs, goff, tile_count, super_count, super_cursor, idsum, zero_end, tile_start, super_start, seg_start,
        tile_neff, order, chunk_start, kref_part, total;
};

static GeomLayout geom_layout(int G, int H, int W) {
    GeomLayout L;
    size_t T = (size_t)dwg_cdiv(W, BT) * dwg_cdiv(H, BT);
    size_t S = (size_t)dwg_cdiv(dwg_cdiv(W, BT), ST) * dwg_cdiv(dwg_cdiv(H, BT), ST);
    size_t g = (size_t)(G > 0 ? G : 1);
    size_t o = 0;
    L.header = o; o += 256;                                      // header .. idsum are cleared together (one memset per frame)
    L.tile_count = o; o = dwg_align_up(o + T * 4, 256);
    L.super_count = o; o = dwg_align_up(o + S * 4, 256);
    L.super_cursor = o; o = dwg_align_up(o + S * 4, 256);
    L.idsum = o; o = dwg_align_up(o + (g / IDBIN + 24) * 4, 256);   // pair count of every run of IDBIN Gaussian indices (one slot per wave of k_preprocess)
    L.zero_end = o;
    L.rec0 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rec1 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rec2 = o; o = dwg_align_up(o + g * sizeof(float4), 256);
    L.rect = o; o = dwg_align_up(o + g * sizeof(uint2), 256);
    L.npairs = o; o = dwg_align_up(o + g * 4, 256);              // (Gaussian, block) pairs of every Gaussian after exact culling ...
    L.goff = o; o = dwg_align_up(o + (g + 1) * 4, 256);          // ... and their exclusive prefix in INDEX order: pair row q = goff[g] + e
    L.tile_start = o; o = dwg_align_up(o + T * 4, 256);          // a block's list is sorted[tile_start, tile_start + tile_count)
    L.super_start = o; o = dwg_align_up(o + (S + 1) * 4, 256);
    L.seg_start = o; o = dwg_align_up(o + (T + 1) * 4, 256);
    L.tile_neff = o; o = dwg_align_up(o + T * 4, 256);
    L.order = o; o = dwg_align_up(o + (size_t)NBUCKET * T * 4, 256);   // render order: bucket k (log2 of the list length) owns order[k T ..]
    L.chunk_start = o; o = dwg_align_up(o + (S + 1) * 4, 256);
    L.kref_part = o; o = dwg_align_up(o + (g / 256 + 2) * 4, 256);
    L.total = o;
    return L;
}

static int64_t seg_capacity(int64_t cap, int H, int W) {
    return cap / SEG + (int64_t)dwg_cdiv(W, BT) * dwg_cdiv(H, BT) + 1;
}
struct PairLayout { size_t keys, cand, sorted, seg_tile, ckpt, part, total; };
static PairLayout pair_layout(int64_t cap, int H, int W) {
    PairLayout L; size_t c = (size_t)(cap > 0 ? cap : 1);
    size_t ns = (size_t)seg_capacity((int64_t)c, H, W);
    L.keys = 0; L.cand = dwg_align_up(c * 8, 256);          // (depth | id) keys per supertile; (block mask | id) candidates in sorted order
    L.sorted = dwg_align_up(L.cand + c * 8, 256);
    L.seg_tile = dwg_align_up(L.sorted + c * 4, 256);
    L.ckpt = dwg_align_up(L.seg_tile + ns * 4, 256);
    L.part = dwg_align_up(L.ckpt + ns * 6 * 64 * sizeof(float), 256);      // backward: one row of GSTRIDE floats per pair, in pair-row order
    L.total = dwg_align_up(L.part + c * 12 * sizeof(float), 256);
    return L;
}
struct ImageLayout { size_t final_T, n_contrib, craw, total; };
static ImageLayout image_layout(int H, int W) {
    ImageLayout L; size_t P = (size_t)H * W;
    L.final_T = 0; L.n_contrib = dwg_align_up(P * 4, 256); L.craw = dwg_align_up(L.n_contrib + P * 4, 256);
    L.total = dwg_align_up(L.craw + 5 * P * 4, 256);      // un-composited C (3), D, A totals
    return L;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// ((m0 x + m4 y) + m8 z) + m12 with every product and sum rounded once: the view depth is a SORT KEY, and near-tied depths must
// order as in the oracle (oracle/raster_oracle.c is built with -ffp-contract=off).  The pragma is what keeps the backend from
// fusing: HIP's __fmul_rn / __fadd_rn are plain operators and contract like any other.
__device__ __forceinline__ float dot3p(float a, float b, float c, float d, float x, float y, float z) {
#pragma clang fp contract(off)
    return ((a * x + b * y) + c * z) + d;
}
__device__ __forceinline__ float3 xform43(const float* m, float3 p) {
    return make_float3(dot3p(m[0], m[4], m[8], m[12], p.x, p.y, p.z), dot3p(m[1], m[5], m[9], m[13], p.x, p.y, p.z),
                       dot3p(m[2], m[6], m[10], m[14], p.x, p.y, p.z));
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
__device__ __forceinline__ void cov3d_of(const float* scales, const float* rots, const float* cov3Dp, size_t i,
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

// The field of view of a camera that lives in device memory (dwg_raster_settings::tanfov): the launch's kernel arguments stay the same
// from frame to frame -- what a captured step that samples a new camera every step needs -- and the focal lengths follow.
__device__ __forceinline__ void camera_scalars(Params& p, size_t cam_offset) {
    if (p.tanfov_dev) {
        p.tanfovx = p.tanfov_dev[cam_offset]; p.tanfovy = p.tanfov_dev[cam_offset + 1];
        p.focal_x = p.W / (2.f * p.tanfovx); p.focal_y = p.H / (2.f * p.tanfovy);
    }
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
// exact (Gaussian, block) culling
// ------------------------------------------------------------------------------------------------
// A splat contributes to a pixel iff power <= 0 and min(0.99, opacity * exp(power)) >= 1/255, i.e. iff the conic's quadratic form
// q(d) = a dx^2 + 2 b dx dy + c dy^2 is <= 2 ln(255 opacity) =: thr.  The margin (1e-4 relative + 1e-3 absolute on q, i.e. 5e-4
// relative on alpha, plus 1e-3 px on every interval end) is hundreds of times the fp32 rounding of the per-pixel evaluation, so
// no contributing pair is ever dropped.
//
// The blocks an ellipse {q <= thr} touches are enumerated by SCANLINE, not by testing every block of the 3-sigma square: for a
// row of blocks (pixel-centre band y in [y0, y1] relative to the centre) the x-projection of (ellipse n band) is one interval
// [xl, xr] -- x_r(y) = (-b y + sqrt(a thr - det y^2)) / a is concave with its maximum at the ellipse's rightmost point, x_l(y) is
// its mirror -- so a row costs two square roots whatever its length, and a lane with a huge splat does O(rows), not O(blocks), work.
struct BlockSpan {
    int by0, by1;        // block rows [by0, by1)
    int bx0, bx1;        // block columns of the reference tile rect [bx0, bx1)
    float gx, gy, ca, cb, cc, thr, det, hy, yr, ica;
    bool all;            // no culling (conic not positive definite / opacity not a positive number): every block of the rect
};
// hardware reciprocal / square root / log2 (1 ulp; ~2e-7 relative): the enumeration only has to be CONSERVATIVE -- its margins are four orders
// of magnitude above these errors -- and identical wherever it is evaluated, and the IEEE-rounded division / sqrtf / logf sequences were most
// of the instructions of a block row (k_preprocess: issue-stall 0.62 on them)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ BlockSpan block_span(float gx, float gy, float ca, float cb, float cc, float opacity, int tx0, int ty0, int tx1,
                                                int ty1, int tiles_x, int tiles_y) {
    // four kernels (pair count, supertile keys, block masks, the backward's row index) evaluate this enumeration and must agree to the bit:
    // no contraction, so that the result never depends on what a call site's surrounding code lets the backend fuse
#pragma clang fp contract(off)
    BlockSpan s;
    s.gx = gx; s.gy = gy; s.ca = ca; s.cb = cb; s.cc = cc;
    s.bx0 = 2 * tx0; s.bx1 = min(2 * tx1, tiles_x); s.by0 = 2 * ty0; s.by1 = min(2 * ty1, tiles_y);
    s.thr = (2.f * 0.6931471805599453f) * __builtin_amdgcn_logf(255.f * opacity) * (1.f + 1e-4f) + 1e-3f;      // 2 ln(255 opacity), with margin
    s.det = ca * cc - cb * cb;
    s.all = !(ca > 0.f) || !(cc > 0.f) || !(s.det > 0.f) || !(s.thr == s.thr);
    s.hy = 0.f; s.yr = 0.f; s.ica = 0.f;
    if (!s.all) {
        if (!(s.thr > 0.f)) { s.by1 = s.by0; return s; }                  // never reaches alpha 1/255
        const float idet = fast_rcp(s.det);
        s.ica = fast_rcp(ca);
        s.hy = fast_sqrt(s.thr * ca * idet) * (1.f + 1e-5f) + 1e-3f;       // half height of the ellipse
        s.yr = -cb * fast_rcp(cc) * fast_sqrt(s.thr * cc * idet);          // y of its rightmost point (leftmost: -yr)
        s.by0 = max(s.by0, (int)ceilf((gy - s.hy - (float)(BT - 1)) * (1.f / BT)));
        s.by1 = min(s.by1, (int)floorf((gy + s.hy) * (1.f / BT)) + 1);
    }
    return s;
}
// columns [xa, xb) of block row `by` the splat reaches
__device__ __forceinline__ void block_row(const BlockSpan& s, int by, int* xa, int* xb) {
#pragma clang fp contract(off)
    if (s.all) { *xa = s.bx0; *xb = s.bx1; return; }
    const float y0 = (float)(by * BT) - s.gy, y1 = y0 + (float)(BT - 1);
    const float ya = fmaxf(y0, -s.hy), yb = fminf(y1, s.hy);
    if (ya > yb) { *xa = 0; *xb = 0; return; }
    const float yR = fminf(yb, fmaxf(ya, s.yr)), yL = fminf(yb, fmaxf(ya, -s.yr));
    const float dR = fmaxf(0.f, s.ca * s.thr - s.det * yR * yR), dL = fmaxf(0.f, s.ca * s.thr - s.det * yL * yL);
    const float mr = 1.f + 1e-5f;                                         // the hardware rcp / sqrt are 1 ulp: widen by 1e-5 relative, far inside the margin below
    const float xr = (-s.cb * yR + fast_sqrt(dR) * mr) * s.ica, xl = (-s.cb * yL - fast_sqrt(dL) * mr) * s.ica;
    const float wr = fabsf(xr) * 1e-5f + 1e-3f, wl = fabsf(xl) * 1e-5f + 1e-3f;
    // block bx holds pixel centres [BT bx, BT bx + BT - 1]: it meets [xl, xr] iff BT bx <= xr + gx and BT bx + BT - 1 >= xl + gx
    int a = (int)ceilf((xl - wl + s.gx - (float)(BT - 1)) * (1.f / BT)), b = (int)floorf((xr + wr + s.gx) * (1.f / BT)) + 1;
    *xa = max(a, s.bx0); *xb = min(b, s.bx1);
}

// ------------------------------------------------------------------------------------------------
// frames: every forward kernel is launched on a (work, F) grid; blockIdx.y = frame, whose inputs / camera / workspaces / outputs lie at a
// fixed stride behind frame 0's (Params::*_stride).  F = 1 for the autograd path; the playback path renders several posed frames per launch.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T* frame_ptr(T* p, size_t stride_bytes) {
    return p ? (T*)((const char*)p + stride_bytes * blockIdx.y) : p;
}
#define DWG_GEOM(ptr) ptr = frame_ptr(ptr, p.geom_stride)
#define DWG_PAIRS(ptr) ptr = frame_ptr(ptr, p.pairs_stride)
#define DWG_IMAGE(ptr) ptr = frame_ptr(ptr, p.image_stride)

// The supertiles a splat reaches, by block-row scanline: the blocks of one supertile row (ST block rows) span columns [cmin, cmax), so the
// supertiles are the contiguous run cmin / ST .. (cmax - 1) / ST of that row.  Returns the splat's exact (Gaussian, block) pair count.
// k_preprocess (counts) and k_scatter_super (keys) run this very function on the same inputs: their enumerations agree by construction.
template <typename Fn>
__device__ __forceinline__ uint32_t for_each_supertile(const BlockSpan& sp, Fn&& fn) {
    uint32_t ng = 0;
    int cmin = 0x7fffffff, cmax = -1;
    for (int by = sp.by0; by < sp.by1; by++) {
        int xa, xb;
        block_row(sp, by, &xa, &xb);
        if (xb > xa) { ng += (uint32_t)(xb - xa); cmin = min(cmin, xa); cmax = max(cmax, xb); }
        if ((by & (ST - 1)) == ST - 1 || by == sp.by1 - 1) {
            if (cmax > cmin) { const int sy = by / ST; for (int sx = cmin / ST; sx <= (cmax - 1) / ST; sx++) fn(sy, sx); }
            cmin = 0x7fffffff; cmax = -1;
        }
    }
    return ng;
}

// ------------------------------------------------------------------------------------------------
// stage A
// ------------------------------------------------------------------------------------------------
template <int PB>
__global__ __launch_bounds__(PB) void k_preprocess(Params p, const float* __restrict__ means3D,
                                                    const float* __restrict__ shs, const float* __restrict__ colors,
                                                    const float* __restrict__ opac, const float* __restrict__ scales,
                                                    const float* __restrict__ rots, const float* __restrict__ cov3Dp,
                                                    int* __restrict__ radii, float4* __restrict__ rec0,
                                                    float4* __restrict__ rec1, float4* __restrict__ rec2,
                                                    uint2* __restrict__ rect, uint32_t* __restrict__ npairs, uint32_t* __restrict__ idsum,
                                                    uint32_t* __restrict__ super_count, uint32_t* __restrict__ kref_part, int use_lds_hist) {
    __shared__ float cam[35], krs[PB / 64];
    extern __shared__ uint32_t hist[];      // [S] workgroup-private supertile histogram (one global atomic per workgroup and supertile)
    const int S = p.stiles_x * p.stiles_y;
    DWG_GEOM(rec0); DWG_GEOM(rec1); DWG_GEOM(rec2); DWG_GEOM(rect); DWG_GEOM(npairs); DWG_GEOM(idsum); DWG_GEOM(super_count); DWG_GEOM(kref_part);
    const size_t go = (size_t)blockIdx.y * (size_t)p.in_stride;            // this frame's first input row
    const size_t co = (size_t)blockIdx.y * (size_t)p.cam_stride;
    camera_scalars(p, co);
    radii += (size_t)blockIdx.y * p.G;
    if (threadIdx.x < 16) cam[threadIdx.x] = p.view[co + threadIdx.x];
    else if (threadIdx.x < 32) cam[threadIdx.x] = p.proj[co + threadIdx.x - 16];
    else if (threadIdx.x < 35 && p.campos) cam[threadIdx.x] = p.campos[co + threadIdx.x - 32];
    if (use_lds_hist) for (int t = threadIdx.x; t < S; t += PB) hist[t] = 0u;
    __syncthreads();
    int i = blockIdx.x * PB + threadIdx.x;
    const bool live_thread = i < p.G;
    if (!live_thread) i = 0;
    else if (p.visit_order) i = p.visit_order[i];
    const size_t gi = go + (size_t)i;
    const float* view = cam; const float* proj = cam + 16;
    float3 pos = make_float3(0.f, 0.f, 0.f);
    if (p.G > 0) pos = make_float3(means3D[3 * gi], means3D[3 * gi + 1], means3D[3 * gi + 2]);
    int radius = 0;
    uint2 rc = make_uint2(0u, 0u);
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    float kref = 0.f;
    BlockSpan sp;
    sp.by0 = sp.by1 = sp.bx0 = sp.bx1 = 0; sp.all = false;
    sp.gx = sp.gy = sp.ca = sp.cb = sp.cc = sp.thr = sp.det = sp.hy = sp.yr = sp.ica = 0.f;
    float3 pv = xform43(view, pos);
    if (live_thread && pv.z > 0.2f) {
        float4 ph = xform44(proj, pos);
        float pw = 1.f / (ph.w + 1e-7f);
        float ndcx = ph.x * pw, ndcy = ph.y * pw;
        float c6[6];
        cov3d_of(scales, rots, cov3Dp, gi, p.scale_mod, c6);
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
            int tx0 = min(p.rtiles_x, max(0, (int)((px - rad) / RT)));
            int ty0 = min(p.rtiles_y, max(0, (int)((py - rad) / RT)));
            int tx1 = min(p.rtiles_x, max(0, (int)((px + rad + RT - 1) / RT)));
            int ty1 = min(p.rtiles_y, max(0, (int)((py + rad + RT - 1) / RT)));
            if ((tx1 - tx0) * (ty1 - ty0) > 0) {
                radius = rad;
                rc = make_uint2((unsigned)tx0 | ((unsigned)ty0 << 16), (unsigned)tx1 | ((unsigned)ty1 << 16));
                kref = (float)((tx1 - tx0) * (ty1 - ty0));
                float3 col; unsigned cb = 0;
                if (colors) col = make_float3(colors[3 * gi], colors[3 * gi + 1], colors[3 * gi + 2]);
                else col = sh_color(p.sh_degree, p.sh_coeffs, shs + gi * p.sh_coeffs * 3, pos, cam + 32, &cb);
                const float op = opac[gi];
                r0 = make_float4(px, py, pv.z, op);
                r1 = make_float4(e.c * di, -e.b * di, e.a * di, 0.f);
                r2 = make_float4(col.x, col.y, col.z, __uint_as_float(cb));
                sp = block_span(px, py, r1.x, r1.y, r1.z, op, tx0, ty0, tx1, ty1, p.tiles_x, p.tiles_y);
            }
        }
    }
    // exact (Gaussian, block) pair count of this splat (its rows of the pair-ordered backward partials) + the supertile histogram
    if (p.dbg & 8) sp.by1 = sp.by0;
    const uint32_t ng = for_each_supertile(sp, [&](int sy, int sx) {
        const int s = sy * p.stiles_x + sx;
        if (use_lds_hist) atomicAdd(&hist[s], 1u); else atomicAdd(&super_count[s], 1u);
    });
    // pair count of this run of IDBIN = 64 indices.  In index order the run IS this wave: one plain store of the wave's sum (64 lanes adding
    // to ONE address with atomics were serialised by the L2's atomic unit -- 30 of this kernel's 37 us at 100 k Gaussians); under a visit
    // order the lanes' indices are scattered and every lane adds its own (integer: order-independent).
    if (!p.visit_order) {
        const float wsum = dwg_wave_sum_to_lane63((float)ng);           // < 2^24 pairs per 64 splats: exact in fp32
        if ((threadIdx.x & 63) == 63) idsum[(blockIdx.x * PB + threadIdx.x) / IDBIN] = (uint32_t)wsum;
    } else if (live_thread && ng) atomicAdd(&idsum[i / IDBIN], ng);
    if (live_thread) {
        npairs[i] = ng;
        radii[i] = radius;
        rect[i] = rc;
        rec0[i] = r0; rec1[i] = r1; rec2[i] = r2;
    }
    // the workgroup's share of K_ref goes to ITS slot (k_scan_super sums the slots): one atomic per wave on the header word serialised
    // 4.7 k same-address atomics at 300 k Gaussians -- half of this kernel's time
    kref = dwg_wave_sum_to_lane63(kref);
    if ((threadIdx.x & 63) == 63) krs[threadIdx.x >> 6] = kref;
    __syncthreads();
    if (threadIdx.x == 0) {
        float k = 0.f;
        for (int w = 0; w < PB / 64; w++) k += krs[w];
        kref_part[blockIdx.x] = (uint32_t)k;
    }
    if (use_lds_hist && !(p.dbg & 32))
        for (int t = threadIdx.x; t < S; t += PB) { uint32_t c = hist[t]; if (c) atomicAdd(&super_count[t], c); }
}

// sort size classes of a supertile list: one wave / 1024 threads with 34 KiB / 1024 threads with 136 KiB of LDS / in global memory
// Workgroup 0: exclusive scans of the supertile counts (super_start) and of their CHUNK counts (chunk_start: a supertile's key list is
// sorted in chunks of CHUNK keys, one workgroup each).  Workgroups 1 .. ceil(G / GTILE): pair rows -- goff = exclusive prefix of the
// per-Gaussian pair counts in INDEX order: row q = goff[g] + e (e: the pair's place in g's block enumeration) is unique per pair, contiguous
// per Gaussian -- the backward's per-pair partials are written by row (k_render_bwd finds e from the splat's geometry) and summed per
// Gaussian as one streamed range (k_gather_partials): no float atomics.  A workgroup scans its GTILE counts on top of the sum of the earlier
// index runs (idsum, accumulated by k_preprocess with integer atomics): no second launch, no inter-workgroup wait.  The last of them leaves
// the frame's pair count K in the header (what the caller sizes the pair workspace by).
__global__ __launch_bounds__(1024) void k_scan_super(Params p, const uint32_t* __restrict__ super_count, uint32_t* __restrict__ super_start,
                                                     uint32_t* __restrict__ chunk_start, int32_t* __restrict__ header,
                                                     const uint32_t* __restrict__ npairs, const uint32_t* __restrict__ idsum,
                                                     uint32_t* __restrict__ goff, const uint32_t* __restrict__ kref_part, int pb) {
    __shared__ uint32_t part[1024], parts[1024];
    DWG_GEOM(super_count); DWG_GEOM(super_start); DWG_GEOM(chunk_start); DWG_GEOM(header); DWG_GEOM(npairs); DWG_GEOM(idsum); DWG_GEOM(goff);
    DWG_GEOM(kref_part);
    const int G = p.G, tid = threadIdx.x;
    if (blockIdx.x > 0) {
        const int b = blockIdx.x - 1, g = b * GTILE + tid;
        uint32_t pre = 0;
        for (int t = tid; t < b * (GTILE / IDBIN); t += 1024) pre += idsum[t];
        part[tid] = pre;
        const uint32_t v = g < G ? npairs[g] : 0u;
        parts[tid] = v;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {             // base: plain tree sum
            if (tid < off) part[tid] += part[tid + off];
            __syncthreads();
        }
        const uint32_t base = part[0];
        for (int off = 1; off < 1024; off <<= 1) {              // inclusive scan of the run's counts
            const uint32_t u = tid >= off ? parts[tid - off] : 0u;
            __syncthreads();
            parts[tid] += u;
            __syncthreads();
        }
        if (g < G) goff[g] = base + parts[tid] - v;
        if (g == G - 1) { goff[G] = base + parts[tid]; header[H_K] = (int32_t)(base + parts[tid]); }
        return;
    }
    const int S = p.stiles_x * p.stiles_y;
    const int chunk = (S + 1023) / 1024;
    const int lo = min(S, tid * chunk), hi = min(S, lo + chunk);
    uint32_t s = 0, sc = 0;
    for (int t = lo; t < hi; t++) { const uint32_t n = super_count[t]; s += n; sc += (n + CHUNK - 1) / CHUNK; }
    part[tid] = s; parts[tid] = sc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = tid >= off ? part[tid - off] : 0u, vc = tid >= off ? parts[tid - off] : 0u;
        __syncthreads();
        part[tid] += v; parts[tid] += vc;
        __syncthreads();
    }
    uint32_t run = part[tid] - s, runc = parts[tid] - sc;
    for (int t = lo; t < hi; t++) {
        const uint32_t n = super_count[t];
        super_start[t] = run; chunk_start[t] = runc;
        run += n; runc += (n + CHUNK - 1) / CHUNK;
    }
    if (tid == 1023) {
        super_start[S] = part[1023]; chunk_start[S] = parts[1023];
        header[H_KS] = (int32_t)part[1023]; header[H_NCHUNK] = (int32_t)parts[1023];
        header[H_TAG] = (int32_t)(atomicAdd(&g_frame_tag, 0x9e3779b1u) | 1u);      // this frame's tag (odd: never the zero of a fresh buffer)
    }
    // K_ref = the sum of k_preprocess's per-workgroup shares
    __syncthreads();
    uint32_t kr = 0;
    for (int t = tid; t < (G + pb - 1) / pb; t += 1024) kr += kref_part[t];
    part[tid] = kr;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) header[H_KREF] = (int32_t)part[0];
}

// ------------------------------------------------------------------------------------------------
// stage B
// ------------------------------------------------------------------------------------------------
// One (depth bits << 32 | id) key per (Gaussian, supertile) into the supertile's range.  Two sweeps over this workgroup's splats: (1) count
// per supertile in LDS, reserve ONE contiguous range per (workgroup, supertile) with a single returning global atomic; (2) hand out slots
// inside the reserved ranges with LDS atomics (the counter then holds the absolute running slot).  The order inside a range is whatever the
// atomics made it: the list is sorted next.
template <int PB>
__global__ __launch_bounds__(PB) void k_scatter_super(Params p, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                       const uint2* __restrict__ rect, const uint32_t* __restrict__ super_start,
                                                       uint32_t* __restrict__ super_cursor, uint64_t* __restrict__ keys,
                                                       int64_t cap, int32_t* __restrict__ header, int use_lds) {
    extern __shared__ uint32_t cnt[];       // [S]
    DWG_GEOM(rec0); DWG_GEOM(rec1); DWG_GEOM(rect); DWG_GEOM(super_start); DWG_GEOM(super_cursor); DWG_GEOM(header); DWG_PAIRS(keys);
    const int S = p.stiles_x * p.stiles_y, G = p.G;
    int i = blockIdx.x * PB + threadIdx.x;
    uint64_t key = 0;
    BlockSpan sp;
    sp.by0 = sp.by1 = sp.bx0 = sp.bx1 = 0; sp.all = false;
    sp.gx = sp.gy = sp.ca = sp.cb = sp.cc = sp.thr = sp.det = sp.hy = sp.yr = sp.ica = 0.f;
    if (i < G) {
        if (p.visit_order) i = p.visit_order[i];
        const uint2 rc = rect[i];
        const int tx0 = (int)(rc.x & 0xffff), ty0 = (int)(rc.x >> 16), tx1 = (int)(rc.y & 0xffff), ty1 = (int)(rc.y >> 16);
        const float4 a = rec0[i], b = rec1[i];
        key = ((uint64_t)__float_as_uint(a.z) << 32) | (uint32_t)i;
        if (tx1 > tx0 && ty1 > ty0) sp = block_span(a.x, a.y, b.x, b.y, b.z, a.w, tx0, ty0, tx1, ty1, p.tiles_x, p.tiles_y);
    }
    if (!use_lds) {
        for_each_supertile(sp, [&](int sy, int sx) {
            const int s = sy * p.stiles_x + sx;
            const int64_t slot = (int64_t)super_start[s] + atomicAdd(&super_cursor[s], 1u);
            if (slot < cap) keys[slot] = key; else header[H_OVERFLOW] = 1;
        });
        return;
    }
    for (int t = threadIdx.x; t < S; t += PB) cnt[t] = 0u;
    __syncthreads();
    for_each_supertile(sp, [&](int sy, int sx) { atomicAdd(&cnt[sy * p.stiles_x + sx], 1u); });
    __syncthreads();
    for (int t = threadIdx.x; t < S; t += PB) {
        const uint32_t c = cnt[t];
        if (c) cnt[t] = super_start[t] + atomicAdd(&super_cursor[t], c);
    }
    __syncthreads();
    for_each_supertile(sp, [&](int sy, int sx) {
        const int64_t slot = (int64_t)atomicAdd(&cnt[sy * p.stiles_x + sx], 1u);
        if (slot < cap) keys[slot] = key; else header[H_OVERFLOW] = 1;
    });
}

struct GlbMem { const uint64_t* p; __device__ uint64_t get(int i) const { return p[i]; } };

// Register-blocked bitonic sort.  A thread owns EPT CONSECUTIVE keys in registers (key i lives in thread i / EPT, register i % EPT), so
// every compare-exchange with stride < EPT is register-only, strides up to 32 * EPT go lane to lane inside the wave, and only the last
// strides of a multi-wave list cross waves through LDS (a pure LDS network moves every key through the CU's one LDS pipe on every one of
// its log^2 passes and is bound by it).  Padding above n is real +inf keys, so the plain network applies.
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}
// LDS slot of key i: one 8-byte pad per 16 keys, so that a thread's consecutive keys and the lanes' strided accesses spread over banks
__device__ __forceinline__ int sort_slot(int i) { return i + (i >> 4); }
struct LdsSlotMem {
    uint64_t* p;
    __device__ uint64_t get(int i) const { return p[sort_slot(i)]; }
    __device__ void set(int i, uint64_t v) { p[sort_slot(i)] = v; }
};

// sorts keys_in[0, n) ascending; on return (behind a barrier) key e is at lds[sort_slot(e)]
template <int THREADS, int EPT>
__device__ __forceinline__ void sort_in_lds(const uint64_t* __restrict__ keys_in, int n, uint64_t* __restrict__ lds, int skip) {
    constexpr int N = THREADS * EPT;
    const int tid = threadIdx.x;
    for (int e = tid; e < N; e += THREADS) lds[sort_slot(e)] = e < n ? keys_in[e] : ~0ull;      // coalesced in, +inf above n
    __syncthreads();
    if (skip) return;
    uint64_t v[EPT];
#pragma unroll
    for (int r = 0; r < EPT; r++) v[r] = lds[sort_slot(tid * EPT + r)];
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= EPT) {
                const int m = j / EPT;                         // partner thread = tid ^ m, same register
                const bool lower = (tid & m) == 0;
                const bool asc = ((tid * EPT) & k) == 0;       // k >= 2 j >= 2 EPT: the direction bit lies in the thread index
                if (m < 64) {
#pragma unroll
                    for (int r = 0; r < EPT; r++) {
                        const uint64_t o = shfl_xor_u64(v[r], m);
                        const bool keep_min = lower == asc;
                        const bool take = keep_min ? (o < v[r]) : (o > v[r]);
                        v[r] = take ? o : v[r];
                    }
                } else {                                       // across waves: through LDS
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EPT; r++) lds[sort_slot(tid * EPT + r)] = v[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EPT; r++) {
                        const uint64_t o = lds[sort_slot((tid ^ m) * EPT + r)];
                        const bool keep_min = lower == asc;
                        const bool take = keep_min ? (o < v[r]) : (o > v[r]);
                        v[r] = take ? o : v[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < EPT; r++) {
                    if ((r & j) == 0) {
                        const int r2 = r | j;
                        const bool asc = ((tid * EPT + r) & k) == 0;
                        const uint64_t a = v[r], b = v[r2];
                        const bool sw = asc ? (b < a) : (a < b);
                        v[r] = sw ? b : a; v[r2] = sw ? a : b;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; r++) lds[sort_slot(tid * EPT + r)] = v[r];
    __syncthreads();
}

// Lanes 0 .. 15 of the workgroup each publish one block of the supertile: its list (start, length), its first checkpoint segment, and its
// place in the render order -- bucket k = log2 of the list length, one array per bucket, a counter per bucket in the header; k_render_fwd
// walks the buckets longest first.  (This replaced a one-workgroup scan kernel over all blocks between the split and the render.)
__device__ __forceinline__ void register_blocks(const Params& p, int sx, int sy, const uint32_t* __restrict__ bs, const uint32_t* __restrict__ bc,
                                                uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
                                                uint32_t* __restrict__ seg_start, uint32_t* __restrict__ order, int32_t* __restrict__ header) {
    if (threadIdx.x >= 64) return;                         // the first wave; its lanes 0 .. 15 own a block each
    const int b = threadIdx.x;
    const int by = sy * ST + b / ST, bx = sx * ST + b % ST;
    const bool mine = b < ST * ST && bx < p.tiles_x && by < p.tiles_y;
    const int t = by * p.tiles_x + bx;
    int k = -1;
    if (mine) {
        uint32_t seg = bc[ST * ST];
        for (int q = 0; q < b; q++) seg += (bc[q] + SEG - 1) / SEG;
        const uint32_t n = bc[b];
        tile_start[t] = bs[b]; tile_count[t] = n; seg_start[t] = seg;
        k = n ? min(NBUCKET - 1, 32 - __clz((int)n)) : 0;
    }
    // one atomic per bucket PRESENT among the supertile's blocks (a returning atomic per block put ~8 k of them on one header word at
    // 1024^2: 130 of this kernel's 173 us)
    const unsigned long long below = (1ull << b) - 1ull;
    unsigned long long todo = __ballot(k >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int kk = __shfl(k, src);
        const unsigned long long same = __ballot(k == kk);
        uint32_t base = 0;
        if (b == src) base = atomicAdd(reinterpret_cast<uint32_t*>(&header[H_BKT0 + kk]), (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, src);
        if (k == kk) order[(size_t)kk * (p.tiles_x * p.tiles_y) + base + (uint32_t)__popcll(same & below)] = (uint32_t)t;
        todo &= ~same;
    }
    (void)t;
}

// The depth-sorted candidates of a supertile -- (block mask << 32 | id), written by k_rank_merge -- split STABLY into the lists of its
// ST x ST blocks (a block's list is a subsequence of its supertile's: ONE sort per 32x32 pixels orders sixteen lists).
//   E2  per wave (a contiguous run of 64-candidate chunks) and per bit: how many candidates carry the bit; a scan over the waves;
//   E3  the supertile takes a contiguous piece of the frame's pair pool with ONE atomic (where a list lands depends on arrival order, what
//       it holds does not) and cuts it into sixteen lists: tile_start / tile_count of its blocks;
//   E4  every wave walks its chunks again: a candidate with bit b goes to slot run_b + (candidates of the chunk below it with bit b).
template <int THREADS, typename Mem>
__device__ __forceinline__ void split_to_blocks(const Params& p, Mem m, int n, int s, uint32_t* __restrict__ tile_count,
                                                uint32_t* __restrict__ tile_start, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ order,
                                                uint32_t* __restrict__ sorted, int64_t cap, int32_t* __restrict__ header,
                                                uint32_t* __restrict__ wt /* [THREADS / 64][16] */, uint32_t* __restrict__ bs /* [16] list starts */,
                                                uint32_t* __restrict__ bc /* [17] list lengths + the first segment */) {
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sx = s % p.stiles_x, sy = s / p.stiles_x;
    const int C = (n + 63) >> 6, per = (C + NW - 1) / NW, c0 = min(C, w * per), c1 = min(C, c0 + per);
    uint32_t mine = 0;
    for (int c = c0; c < c1; c++) {
        const int e = c * 64 + lane;
        const uint32_t mask = e < n ? (uint32_t)(m.get(e) >> 32) : 0u;
#pragma unroll
        for (int b = 0; b < ST * ST; b++) {
            const unsigned long long bal = __ballot((mask >> b) & 1u);
            if (lane == b) mine += (uint32_t)__popcll(bal);
        }
    }
    if (lane < ST * ST) wt[w * (ST * ST) + lane] = mine;
    __syncthreads();
    if (tid < ST * ST) {
        uint32_t run = 0;
        for (int k = 0; k < NW; k++) { const uint32_t t = wt[k * (ST * ST) + tid]; wt[k * (ST * ST) + tid] = run; run += t; }
        bs[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int b = 0; b < ST * ST; b++) tot += bs[b];
        const uint32_t base = tot ? atomicAdd(reinterpret_cast<uint32_t*>(&header[H_POOL]), tot) : 0u;
        if ((int64_t)base + (int64_t)tot > cap) header[H_OVERFLOW] = 1;          // truncated: the caller renders the frame again
        uint32_t run = base, nseg = 0;
        for (int b = 0; b < ST * ST; b++) {
            const uint32_t c = bs[b];
            const int64_t room = cap - (int64_t)run;
            const uint32_t cc = room <= 0 ? 0u : (uint32_t)min((int64_t)c, room);      // what of the list fits the pair workspace
            bs[b] = run; bc[b] = cc;
            nseg += (cc + SEG - 1) / SEG;
            run += c;
        }
        // the supertile's checkpoint segments: one piece of the frame's segment numbering (k_render_bwd runs one wave per segment)
        bc[ST * ST] = nseg ? atomicAdd(reinterpret_cast<uint32_t*>(&header[H_NSEG]), nseg) : 0u;
    }
    __syncthreads();
    register_blocks(p, sx, sy, bs, bc, tile_count, tile_start, seg_start, order, header);
    uint32_t runb = lane < ST * ST ? bs[lane] + wt[w * (ST * ST) + lane] : 0u;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = c0; c < c1; c++) {
        const int e = c * 64 + lane;
        const uint64_t kv = e < n ? m.get(e) : 0ull;
        const uint32_t mask = (uint32_t)(kv >> 32), id = (uint32_t)kv;
#pragma unroll
        for (int b = 0; b < ST * ST; b++) {
            const bool bit = (mask >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)runb, b);
            if (bit) {
                const int64_t pos = (int64_t)r0 + __popcll(bal & below);
                if (pos < cap) sorted[pos] = id;
            }
            if (lane == b) runb += (uint32_t)__popcll(bal);
        }
    }
}

// Which supertile chunk c of the frame belongs to, its place in the supertile's key list and its length (wave-uniform: every lane searches
// chunk_start alike).  false: no such chunk.
struct ChunkRef { int s; int j; int64_t sa; int nsup; int a; int n; };      // supertile, chunk index, list start, list length, chunk offset / length
__device__ __forceinline__ bool find_chunk(const Params& p, int c, const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ super_start,
                                           const int32_t* __restrict__ header, int64_t cap, ChunkRef* r) {
    if (c >= header[H_NCHUNK]) return false;
    int lo = 0, hi = p.stiles_x * p.stiles_y;              // largest s with chunk_start[s] <= c (chunk_start[S] = total > c)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)chunk_start[mid] <= c) lo = mid; else hi = mid; }
    int64_t sa = super_start[lo], se = super_start[lo + 1];
    if (sa > cap) sa = cap; if (se > cap) se = cap;
    r->s = lo; r->j = c - (int)chunk_start[lo]; r->sa = sa; r->nsup = (int)(se - sa);
    r->a = r->j * CHUNK; r->n = min(CHUNK, r->nsup - r->a);
    return r->n > 0;
}

// One workgroup per CHUNK keys of a supertile's list: sorted in place.  Every chunk of every supertile of the frame at once -- the dense
// supertiles (thousands of keys, a few dozen of them under a body) no longer pin ONE workgroup each to a long barrier-bound network while
// the rest of the chip idles (SQ counters of that version: waves parked 0.48 of the time).
__global__ __launch_bounds__(256) void k_chunk_sort(Params p, const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ super_start,
                                                    const int32_t* __restrict__ header, uint64_t* __restrict__ keys, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DWG_GEOM(chunk_start); DWG_GEOM(super_start); DWG_GEOM(header); DWG_PAIRS(keys);
    ChunkRef r;
    if (!find_chunk(p, (int)blockIdx.x, chunk_start, super_start, header, cap, &r)) return;
    uint64_t* lds = reinterpret_cast<uint64_t*>(smem_raw);
    uint64_t* kc = keys + r.sa + r.a;
    if (r.n <= 256) sort_in_lds<256, 1>(kc, r.n, lds, p.dbg & 1);
    else if (r.n <= 512) sort_in_lds<256, 2>(kc, r.n, lds, p.dbg & 1);
    else sort_in_lds<256, 4>(kc, r.n, lds, p.dbg & 1);
    for (int e = threadIdx.x; e < r.n; e += 256) kc[e] = lds[sort_slot(e)];
}

// first index of sorted run[0, len) whose key is not below `key` = the number of its keys below `key` (keys are unique: (depth, id))
__device__ __forceinline__ int lower_bound_u64(const uint64_t* __restrict__ run, int len, uint64_t key) {
    int lo = 0;
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = run[lo + half] < key;
        lo = lt ? lo + half + 1 : lo;
        len = lt ? len - half - 1 : half;
    }
    return lo;
}

// One workgroup per chunk again, one thread per key: the key's place in its supertile's sorted list is its place in its own chunk plus the
// number of keys below it in every sibling chunk (a binary search each: the siblings are sorted) -- a multiway merge without a merge tree,
// every key independent.  The thread also works out the candidate's 16-bit BLOCK MASK -- bit 4 r + c set iff block (ST sx + c, ST sy + r)
// lies in the splat's scanline enumeration (block_span / block_row, the functions k_preprocess counted the pairs with) -- and leaves
// (mask << 32 | id) at the key's sorted place: what k_split cuts into block lists.
__global__ __launch_bounds__(256) void k_rank_merge(Params p, const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ super_start,
                                                    const int32_t* __restrict__ header, const uint64_t* __restrict__ keys,
                                                    const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                    const uint2* __restrict__ rect, uint64_t* __restrict__ cand, int64_t cap) {
    DWG_GEOM(chunk_start); DWG_GEOM(super_start); DWG_GEOM(header); DWG_GEOM(rec0); DWG_GEOM(rec1); DWG_GEOM(rect); DWG_PAIRS(keys); DWG_PAIRS(cand);
    ChunkRef r;
    if (!find_chunk(p, (int)blockIdx.x, chunk_start, super_start, header, cap, &r)) return;
    const uint64_t* ks = keys + r.sa;
    const int nch = (r.nsup + CHUNK - 1) / CHUNK;
    const int sx = r.s % p.stiles_x, sy = r.s / p.stiles_x;
    constexpr int KPT = CHUNK / 256;
    uint64_t key[KPT]; uint2 rc[KPT]; float4 ra[KPT], rb[KPT]; int pos[KPT];
#pragma unroll
    for (int u = 0; u < KPT; u++) {                            // the record gathers of the mask are in flight under the searches
        const int e = threadIdx.x + u * 256;
        key[u] = e < r.n ? ks[r.a + e] : 0ull;
        const uint32_t id = (uint32_t)key[u];
        rc[u] = rect[id]; ra[u] = rec0[id]; rb[u] = rec1[id];
        pos[u] = e;                                            // place in the own chunk ...
    }
    // ... plus the keys below it in every sibling chunk, four siblings per round.  A search is two-level: six steps over the sibling's 64
    // PIVOTS (every 16th key, staged in LDS by the workgroup once per round), then at most five steps inside the one 16-key window they
    // leave -- 128 bytes, one or two lines -- instead of eleven dependent probes spread over the sibling's 8 KiB (the flat searches were
    // bound by the request rate of those probes: 156 us at 300 k Gaussians / 1024^2, unchanged by keeping sixteen of them in flight).
    __shared__ uint64_t piv[4][64];
    for (int j0 = 0; j0 < nch; j0 += 4) {
        __syncthreads();                                       // the previous round's pivots have been read
        {
            const int q = threadIdx.x >> 6, i = threadIdx.x & 63, jj = j0 + q;
            const int l = (jj < nch && jj != r.j) ? min(CHUNK, r.nsup - jj * CHUNK) : 0;
            piv[q][i] = 16 * i + 15 < l ? ks[(size_t)jj * CHUNK + 16 * i + 15] : ~0ull;      // +inf behind the last whole window
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jj = j0 + q;
            const int l = (jj < nch && jj != r.j) ? min(CHUNK, r.nsup - jj * CHUNK) : 0;
            if (l == 0) continue;                              // wave-uniform
            const uint64_t* run = ks + (size_t)jj * CHUNK;
            int w0[KPT], lo[KPT], len[KPT];
#pragma unroll
            for (int u = 0; u < KPT; u++) {                    // pivots below the key: all of their windows lie below it too
                int c = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) c += (piv[q][c + step - 1] < key[u]) ? step : 0;
                c += (c == 63 && piv[q][63] < key[u]) ? 1 : 0;
                w0[u] = 16 * c; lo[u] = 0; len[u] = max(0, min(16, l - 16 * c));
            }
#pragma unroll
            for (int step = 0; step < 5; step++) {
#pragma unroll
                for (int u = 0; u < KPT; u++) {
                    const int half = len[u] >> 1;
                    const bool live = len[u] > 0;
                    const uint64_t v = live ? run[w0[u] + lo[u] + half] : 0ull;
                    const bool lt = live && v < key[u];
                    lo[u] = lt ? lo[u] + half + 1 : lo[u];
                    len[u] = lt ? len[u] - half - 1 : half;
                }
            }
#pragma unroll
            for (int u = 0; u < KPT; u++) pos[u] += w0[u] + lo[u];
        }
    }
#pragma unroll
    for (int u = 0; u < KPT; u++) {
        const int e = threadIdx.x + u * 256;
        if (e >= r.n) break;
        uint32_t mask = 0;
        if (!(p.dbg & 2)) {
            const float4 a = ra[u], b = rb[u];
            const BlockSpan sp = block_span(a.x, a.y, b.x, b.y, b.z, a.w, (int)(rc[u].x & 0xffff), (int)(rc[u].x >> 16), (int)(rc[u].y & 0xffff),
                                            (int)(rc[u].y >> 16), p.tiles_x, p.tiles_y);
#pragma unroll
            for (int rr = 0; rr < ST; rr++) {
                const int by = sy * ST + rr;
                if (by >= sp.by0 && by < sp.by1) {
                    int xa, xb;
                    block_row(sp, by, &xa, &xb);
                    const int lo = max(xa, sx * ST), hi = min(xb, sx * ST + ST);
                    if (hi > lo) mask |= ((1u << (hi - lo)) - 1u) << (rr * ST + lo - sx * ST);
                }
            }
        } else mask = 1u;
        cand[r.sa + pos[u]] = ((uint64_t)mask << 32) | (uint32_t)key[u];
    }
}

// One workgroup per supertile: its sorted candidates -> its sixteen block lists (split_to_blocks); a supertile nothing reaches only
// publishes its (empty) blocks.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_split(Params p, int32_t* __restrict__ header, const uint32_t* __restrict__ super_start,
                                                   const uint64_t* __restrict__ cand, uint32_t* __restrict__ tile_count,
                                                   uint32_t* __restrict__ tile_start, uint32_t* __restrict__ seg_start,
                                                   uint32_t* __restrict__ order, uint32_t* __restrict__ sorted, int64_t cap) {
    __shared__ uint32_t wt[(THREADS / 64) * ST * ST], bs[ST * ST], bc[ST * ST + 1];
    DWG_GEOM(header); DWG_GEOM(super_start); DWG_GEOM(tile_count); DWG_GEOM(tile_start); DWG_GEOM(seg_start); DWG_GEOM(order);
    DWG_PAIRS(cand); DWG_PAIRS(sorted);
    const int s = (int)blockIdx.x;
    int64_t a = super_start[s], e = super_start[s + 1];
    if (a > cap) a = cap; if (e > cap) e = cap;
    const int n = (int)(e - a);
    if (n <= 0) {
        if (threadIdx.x <= ST * ST) { if (threadIdx.x < ST * ST) bs[threadIdx.x] = 0u; bc[threadIdx.x] = 0u; }
        __syncthreads();
        register_blocks(p, s % p.stiles_x, s / p.stiles_x, bs, bc, tile_count, tile_start, seg_start, order, header);
        return;
    }
    GlbMem m{cand + a};
    split_to_blocks<THREADS>(p, m, n, s, tile_count, tile_start, seg_start, order, sorted, cap, header, wt, bs, bc);
}

// One wave64 per 8x8 pixel block, longest list first.  Splat records of the current batch of 64 live in LDS (broadcast reads);
// the next batch's records are gathered into registers while the current one is composited.
__global__ __launch_bounds__(64) void k_render_fwd(Params p, const int32_t* __restrict__ header, const uint32_t* __restrict__ order,
                                                   const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_count,
                                                   const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ sorted,
                                                   const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                   const float4* __restrict__ rec2, int64_t cap, int64_t cap_segs,
                                                   uint32_t* __restrict__ seg_tile, float* __restrict__ ckpt,
                                                   uint32_t* __restrict__ tile_neff, float* __restrict__ final_T,
                                                   int* __restrict__ n_contrib, float* __restrict__ craw,
                                                   float* __restrict__ out_color, float* __restrict__ out_depth,
                                                   float* __restrict__ out_alpha) {
    __shared__ float4 s0[64], s1[64], s2[64];
    DWG_GEOM(header); DWG_GEOM(order); DWG_GEOM(tile_start); DWG_GEOM(tile_count); DWG_GEOM(seg_start); DWG_GEOM(rec0); DWG_GEOM(rec1); DWG_GEOM(rec2);
    DWG_GEOM(tile_neff);
    DWG_PAIRS(sorted); DWG_PAIRS(seg_tile); DWG_PAIRS(ckpt); DWG_IMAGE(final_T); DWG_IMAGE(n_contrib); DWG_IMAGE(craw);
    {
        const size_t fo = (size_t)blockIdx.y * (size_t)p.H * p.W;
        out_color += 3 * fo; out_depth += fo; out_alpha += fo;
    }
    // longest lists first: bucket NBUCKET - 1 down to 0, each with its own array (filled by k_split in arrival order)
    int tile;
    {
        int r = (int)blockIdx.x, k = NBUCKET - 1;
        for (; k > 0; k--) { const int c = header[H_BKT0 + k]; if (r < c) break; r -= c; }
        tile = (int)order[(size_t)k * (p.tiles_x * p.tiles_y) + r];
    }
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int lane = threadIdx.x;
    const int px = tx * BT + (lane & 7), py = ty * BT + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const int64_t rs = tile_start[tile];                    // the block's list: sorted[rs, rs + n) (n already clamped to the capacity)
    const int n = (int)tile_count[tile];
    const int64_t sbase = seg_start[tile];
    for (int s = lane; s < (n + SEG - 1) / SEG; s += 64) if (sbase + s < cap_segs) seg_tile[sbase + s] = (uint32_t)tile;
    const float fx = (float)px, fy = (float)py;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    int last = 0;
    bool done = !inside;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    if (lane < n) { const uint32_t g = sorted[rs + lane]; r0 = rec0[g]; r1 = rec1[g]; r2 = rec2[g]; }
    for (int base = 0; base < n; base += 64) {
        if (!__any(!done)) break;
        if (ckpt && base > 0 && (base % SEG) == 0) {          // segment 0 starts from the known state (T = 1, sums 0): neither written nor read
            const int64_t seg = sbase + base / SEG;
            if (seg < cap_segs) {
                float* c = ckpt + (size_t)seg * 6 * 64 + lane;
                c[0] = T; c[64] = C0; c[128] = C1; c[192] = C2; c[256] = D; c[320] = A;
            }
        }
        s0[lane] = r0; s1[lane] = r1; s2[lane] = r2;
        const int nb = base + 64 + lane;
        if (nb < n) { const uint32_t g = sorted[rs + nb]; r0 = rec0[g]; r1 = rec1[g]; r2 = rec2[g]; }
        __syncthreads();
        const int cnt = min(64, n - base);
        // Branch-free walk over the batch (round 3; SQ counters of the nested-branch version: as many SALU as VALU instructions -- the
        // exec-mask bookkeeping of three nested divergent ifs per splat -- and waves parked 42 % of the time on LDS reads issued one
        // splat at a time).  A pixel that skips a splat or has terminated adds w = 0 (fma(c, 0, C) == C exactly), so the arithmetic of
        // every contributing splat -- and with it every output bit -- is the one of the branchy loop.
        bool alive = !done;
        // four splats per trip: their twelve 16-byte LDS reads are issued together, the four evaluations (serial through T) run under
        // the later reads' latency
        for (int j = 0; j < cnt && __any(alive); j += 4) {
            float4 ra[4], rb[4], rc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ju = min(j + u, cnt - 1);
                ra[u] = s0[ju]; rb[u] = s1[ju]; rc[u] = s2[ju];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float4 a = ra[u], b = rb[u], c = rc[u];
                const float dx = a.x - fx, dy = a.y - fy;
                const float power = splat_power(b.x, b.y, b.z, dx, dy);
                const float alpha = fminf(0.99f, a.w * expf(power));
                const bool hit = alive && (j + u < cnt) && !(power > 0.f) && !(alpha < (1.f / 255.f));     // negated forms: as the skip tests treat a NaN
                const float test_T = __fmul_rn(T, 1.f - alpha);
                const bool stop = hit && test_T < 0.0001f;
                const bool upd = hit && !stop;
                alive = alive && !stop;
                const float w = upd ? __fmul_rn(alpha, T) : 0.f;
                C0 = __fmaf_rn(c.x, w, C0); C1 = __fmaf_rn(c.y, w, C1); C2 = __fmaf_rn(c.z, w, C2);
                D = __fmaf_rn(a.z, w, D); A += w;
                T = upd ? test_T : T;
                last = upd ? base + j + u + 1 : last;
            }
        }
        done = !alive;
        __syncthreads();
    }
    // deepest contributor of the block: nothing behind it matters to the backward
    int mx = last;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    if (lane == 0) tile_neff[tile] = (uint32_t)mx;
    if (inside) {
        const size_t P = (size_t)p.H * p.W, pix = (size_t)py * p.W + px;
        final_T[pix] = T; n_contrib[pix] = last;
        craw[pix] = C0; craw[P + pix] = C1; craw[2 * P + pix] = C2; craw[3 * P + pix] = D; craw[4 * P + pix] = A;
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

// One wave64 per (block, segment of SEG splats), front to back from the forward's checkpoint.  For splat i of a pixel:
//   w_i = alpha_i T_i,   sum over LATER splats of c_j w_j = (final sum) - (prefix sum including i)
//   dL/dalpha_i = sum_ch (c_i T_i - later_ch / (1 - alpha_i)) g_ch - T_final / (1 - alpha_i) (bg . g_rgb)
// T_i follows the forward's own recurrence from the checkpoint, bit for bit.
//
// Round 4: no cross-lane reduction per splat and no float atomics.  The walk (a lane = a pixel, serial through T) only leaves two numbers
// per (splat, pixel) in LDS -- Q = dL/dalpha * G and W = alpha T; every gradient component of a pair is a sum over the block's pixels of
// Q or W times a polynomial in (dx, dy) / the pixel's incoming gradient.  After SUB splats the wave TURNS: four lanes per splat, each
// summing a quarter of the pixels in registers (9 running sums, 16 trips), two quad-DPP steps to join the quarters -- ~18 instructions per
// splat where nine 6-step wave reductions took 54 -- and the pair's row of partials goes out with plain 16-byte stores to row q of the
// pair-ordered buffer: row q = goff[g] + e, e = the block's place in Gaussian g's own block enumeration (the scanline walk of k_preprocess,
// recomputed here from the splat's record by the four lanes of its quad: ~100 instructions per TURN for typical splats).  The forward is
// untouched by all this.  k_gather_partials sums a Gaussian's rows, a contiguous range, in a fixed order: the rasterizer's backward is now
// bit-reproducible, and the ~9 M float atomics per frame (each forwarded to the memory side on this chip) are gone.  A row carries the
// frame's TAG in its last word: pairs no segment reaches (behind the block's deepest contributor) simply keep an old tag and are skipped by
// the gather -- no zero-fill, no memset.
// GD: a depth-map gradient is given (component 9 is non-zero only then; the SDS path has none).
// second word of a row's 64-bit frame tag (a 32-bit tag alone would match left-over bits once per ~4000 frames at a million rows)
__device__ __forceinline__ uint32_t dwg_tag2(uint32_t tag) { return (tag * 0x85ebca6bu) ^ 0xc2b2ae35u; }
#define SUB 16           // splats per turn of the two-phase reduction
#define SUBLD 65         // row stride of the Q / W tables: lane (splat s, quarter h) reads [s][16 h + i] -> bank (s + 16 h + i) mod 64, conflict-free
template <bool GD>
__global__ __launch_bounds__(64) void k_render_bwd(Params p, const int32_t* __restrict__ header, int64_t cap_segs,
                                                   const uint32_t* __restrict__ seg_tile, const uint32_t* __restrict__ seg_start,
                                                   const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_count,
                                                   const uint32_t* __restrict__ tile_neff,
                                                   const uint32_t* __restrict__ sorted, const uint2* __restrict__ rect,
                                                   const uint32_t* __restrict__ goff, uint32_t tag, const float4* __restrict__ rec0,
                                                   const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                                                   int64_t cap, const float* __restrict__ ckpt, const float* __restrict__ final_T,
                                                   const int* __restrict__ n_contrib, const float* __restrict__ craw,
                                                   const float* __restrict__ g_color, const float* __restrict__ g_depth,
                                                   const float* __restrict__ g_alpha, float* __restrict__ part /* [cap][GSTRIDE] */) {
    __shared__ float4 s0[64], s1[64], s2[64];
    __shared__ uint2 srect[64];
    __shared__ uint32_t sgo[64];
    __shared__ float qt[SUB * SUBLD], wt[SUB * SUBLD];
    __shared__ float gpx[4][64];
    // frames (blockIdx.y): this frame's workspaces and its slice of the image gradients
    DWG_GEOM(header); DWG_PAIRS(seg_tile); DWG_GEOM(seg_start); DWG_GEOM(tile_start); DWG_GEOM(tile_count); DWG_GEOM(tile_neff);
    DWG_PAIRS(sorted); DWG_GEOM(rect); DWG_GEOM(goff); DWG_GEOM(rec0); DWG_GEOM(rec1); DWG_GEOM(rec2); DWG_PAIRS(ckpt);
    DWG_IMAGE(final_T); DWG_IMAGE(n_contrib); DWG_IMAGE(craw); DWG_PAIRS(part);
    {
        const size_t fo = (size_t)blockIdx.y * (size_t)p.H * p.W;
        g_color += 3 * fo;
        if (g_depth) g_depth += fo;
        if (g_alpha) g_alpha += fo;
    }
    const int64_t seg = blockIdx.x;
    if (seg >= (int64_t)header[H_NSEG] || seg >= cap_segs || header[H_OVERFLOW]) return;    // a truncated frame is redone by the caller
    tag = (uint32_t)header[H_TAG];                             // the frame's tag (the argument is unused: see g_frame_tag)
    const int tile = (int)seg_tile[seg];
    if ((unsigned)tile >= (unsigned)(p.tiles_x * p.tiles_y)) return;
    const int sidx = (int)(seg - (int64_t)seg_start[tile]);
    const int64_t rs = tile_start[tile];
    const int lo = sidx * SEG;
    const int seg_hi = min((int)tile_count[tile], lo + SEG);       // the rows this wave owns: [lo, seg_hi)
    const int hi = min(seg_hi, (int)tile_neff[tile]);              // ... of which [lo, hi) can carry a gradient
    if (lo >= seg_hi) return;
    const int lane = threadIdx.x;
    if (lo >= hi) return;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int px = tx * BT + (lane & 7), py = ty * BT + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float fx = (float)px, fy = (float)py;
    const size_t P = (size_t)p.H * p.W, pix = (size_t)py * p.W + px;
    float T_final = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, td = 0.f, ta = 0.f;     // totals of the forward
    int last = 0;
    float gp0 = 0.f, gp1 = 0.f, gp2 = 0.f, gpd = 0.f, gpa = 0.f;
    if (inside) {
        T_final = final_T[pix]; last = n_contrib[pix];
        t0 = craw[pix]; t1 = craw[P + pix]; t2 = craw[2 * P + pix]; td = craw[3 * P + pix]; ta = craw[4 * P + pix];
        gp0 = g_color[pix]; gp1 = g_color[P + pix]; gp2 = g_color[2 * P + pix];
        if (GD) gpd = g_depth[pix];
        if (g_alpha) gpa = g_alpha[pix];
    }
    gpx[0][lane] = gp0; gpx[1][lane] = gp1; gpx[2][lane] = gp2; gpx[3][lane] = gpd;
    const float bgdot = p.bg[0] * gp0 + p.bg[1] * gp1 + p.bg[2] * gp2;
    const float* ck = ckpt + (size_t)seg * 6 * 64 + lane;
    float T = 1.f, P0 = 0.f, P1 = 0.f, P2 = 0.f, Pd = 0.f, Pa = 0.f;
    if (sidx > 0) { T = ck[0]; P0 = ck[64]; P1 = ck[128]; P2 = ck[192]; Pd = ck[256]; Pa = ck[320]; }
    const float ddelx = 0.5f * p.W, ddely = 0.5f * p.H;
    // second phase: lane = 4 s + h sums pixels 16 h .. 16 h + 15 (block rows 2 h, 2 h + 1) of splat s
    const int ps = lane >> 2, ph = lane & 3;
    const float pfy0 = (float)(ty * BT + 2 * ph), pfx0 = (float)(tx * BT);
    for (int base = lo; base < hi; base += 64) {
        const int cnt = min(64, hi - base);
        __syncthreads();
        if (lane < cnt) {
            const uint32_t gid = sorted[rs + base + lane];
            s0[lane] = rec0[gid]; s1[lane] = rec1[gid]; s2[lane] = rec2[gid]; srect[lane] = rect[gid]; sgo[lane] = goff[gid];
        }
        __syncthreads();
        for (int jb = 0; jb < cnt; jb += SUB) {
            const int jn = min(SUB, cnt - jb);
            unsigned anym = 0u;                                 // wave-uniform: splats of this turn with a contributing pixel
            float4 an = s0[jb], bn = s1[jb], cn = s2[jb];
            for (int j = 0; j < jn; j++) {
                const float4 a = an, b = bn, col = cn;
                const int jx = jb + min(j + 1, jn - 1);
                an = s0[jx]; bn = s1[jx]; cn = s2[jx];          // the next splat's records are in flight while this one is evaluated
                const float dx = a.x - fx, dy = a.y - fy;
                const float power = splat_power(b.x, b.y, b.z, dx, dy);
                const float Gv = expf(power);
                const float alpha = fminf(0.99f, a.w * Gv);
                const bool valid = (base + jb + j < last) && (power <= 0.f) && (alpha >= (1.f / 255.f));
                float Qv = 0.f, Wv = 0.f;
                if (valid) {
                    const float om = 1.f - alpha;
                    const float inv1a = 1.f / om;
                    const float w = __fmul_rn(alpha, T);
                    P0 = __fmaf_rn(col.x, w, P0); P1 = __fmaf_rn(col.y, w, P1); P2 = __fmaf_rn(col.z, w, P2);
                    Pd = __fmaf_rn(a.z, w, Pd); Pa += w;
                    const float dL_dalpha = (col.x * T - (t0 - P0) * inv1a) * gp0 + (col.y * T - (t1 - P1) * inv1a) * gp1 +
                                            (col.z * T - (t2 - P2) * inv1a) * gp2 + (a.z * T - (td - Pd) * inv1a) * gpd +
                                            (T - (ta - Pa) * inv1a) * gpa - T_final * inv1a * bgdot;
                    Qv = dL_dalpha * Gv; Wv = w;
                    T = __fmul_rn(T, om);
                }
                if (__any(valid)) anym |= 1u << j;
                qt[j * SUBLD + lane] = Qv; wt[j * SUBLD + lane] = Wv;
            }
            __syncthreads();
            // the turn: per (splat, pixel quarter) sums, joined across the quad
            {
                const bool mine = ps < jn;
                const bool work = mine && ((anym >> ps) & 1u);
                float S = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Cd = 0.f;
                const float4 a = s0[jb + (mine ? ps : 0)];
                if (work) {
                    const float* qrow = qt + ps * SUBLD + 16 * ph; const float* wrow = wt + ps * SUBLD + 16 * ph;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float Q = qrow[i], Wq = wrow[i];
                        const float dx = a.x - (pfx0 + (float)(i & 7)), dy = a.y - (pfy0 + (float)(i >> 3));
                        const float Qx = Q * dx, Qy = Q * dy;
                        S += Q; Sx += Qx; Sy += Qy;
                        Sxx = fmaf(Qx, dx, Sxx); Sxy = fmaf(Qx, dy, Sxy); Syy = fmaf(Qy, dy, Syy);
                        const int pp = 16 * ph + i;
                        C0 = fmaf(Wq, gpx[0][pp], C0); C1 = fmaf(Wq, gpx[1][pp], C1); C2 = fmaf(Wq, gpx[2][pp], C2);
                        if (GD) Cd = fmaf(Wq, gpx[3][pp], Cd);
                    }
                }
#define DWG_QUAD_SUM(v) do { v = dwg_dpp_add<0xB1, 0xF>(v); v = dwg_dpp_add<0x4E, 0xF>(v); } while (0)
                DWG_QUAD_SUM(S); DWG_QUAD_SUM(Sx); DWG_QUAD_SUM(Sy); DWG_QUAD_SUM(Sxx); DWG_QUAD_SUM(Sxy); DWG_QUAD_SUM(Syy);
                DWG_QUAD_SUM(C0); DWG_QUAD_SUM(C1); DWG_QUAD_SUM(C2);
                if (GD) DWG_QUAD_SUM(Cd);
#undef DWG_QUAD_SUM
                if (mine) {
                    const float4 b = s1[jb + ps];
                    // the pair's row: goff[g] + (blocks of g's enumeration before this one).  The quad shares the splat's block rows above
                    // this block; the enumeration is k_preprocess's (same block_span / block_row), so rows are a bijection onto g's range.
                    const uint2 rc = srect[jb + ps];
                    const BlockSpan sp = block_span(a.x, a.y, b.x, b.y, b.z, a.w, (int)(rc.x & 0xffff), (int)(rc.x >> 16), (int)(rc.y & 0xffff),
                                                    (int)(rc.y >> 16), p.tiles_x, p.tiles_y);
                    int before = 0;
                    for (int by = sp.by0 + ph; by < ty; by += 4) { int xa, xb; block_row(sp, by, &xa, &xb); before += max(0, xb - xa); }
                    before += __shfl_xor(before, 1); before += __shfl_xor(before, 2);
                    int xa, xb;
                    block_row(sp, ty, &xa, &xb);
                    const int64_t q = (int64_t)sgo[jb + ps] + before + (tx - xa);
                    if (ph < 3 && q < cap) {
                        const float ao = a.w;                           // opacity: dL/dG = opacity * dL/dalpha
                        float4 o;
                        if (ph == 0) o = make_float4(-ao * ddelx * (b.x * Sx + b.y * Sy), -ao * ddely * (b.z * Sy + b.y * Sx), -0.5f * ao * Sxx, -0.5f * ao * Sxy);
                        else if (ph == 1) o = make_float4(-0.5f * ao * Syy, S, C0, C1);
                        else o = make_float4(C2, Cd, __uint_as_float(dwg_tag2(tag)), __uint_as_float(tag));
                        reinterpret_cast<float4*>(part + (size_t)q * GSTRIDE)[ph] = o;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// gacc[g][:] = sum of Gaussian g's pair rows [goff[g], goff[g+1]) -- a contiguous range -- in row order: deterministic, no atomics.
// Four lanes per Gaussian: lane h of the quad adds 16-byte piece h of every row, so a quad reads one contiguous 48-byte row per trip (16
// rows per wave-instruction, neighbouring Gaussians' rows being neighbours in memory) and ends up holding piece h of the sum -- no
// cross-lane reduction, no LDS, no barrier.  (Tried first: a thread per Gaussian reading its own rows -- 64 scattered lines per instruction,
// 53 us at 50 k Gaussians; streaming a workgroup's whole run through LDS -- the chunk loop's length is set by the big splats in the run,
// 150-960 us.)  Big splats (> GBIG rows; 64 by default: the quad path costs a wave its LONGEST quad, a dependent trip per four rows, and the pair
// counts are heavy-tailed -- at 192 the launch was its tail, waves parked 0.82 of the time) are summed by their whole wave, lane-strided over
// the rows, and joined by a fixed tree.
__global__ __launch_bounds__(256) void k_gather_partials(int G, const uint32_t* __restrict__ goff, const float* __restrict__ part, int64_t cap,
                                                         const int32_t* __restrict__ header, uint32_t tag, float* __restrict__ gacc, int GBIG,
                                                         size_t geom_stride, size_t pairs_stride) {
    goff = frame_ptr(goff, geom_stride); header = frame_ptr(header, geom_stride); part = frame_ptr(part, pairs_stride);     // frame blockIdx.y
    gacc += (size_t)blockIdx.y * (size_t)G * GSTRIDE;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2), h = threadIdx.x & 3, lane = threadIdx.x & 63;
    const bool ok = !header[H_OVERFLOW];                       // a truncated frame is redone by the caller: zeros
    tag = (uint32_t)header[H_TAG];
    const int64_t capc = cap > 0 ? cap : 0;
    int64_t q0 = 0, q1 = 0;
    if (i < G && ok) { q0 = min((int64_t)goff[i], capc); q1 = min((int64_t)goff[i + 1], capc); }
    const int64_t n = q1 - q0;
    const bool big = n > GBIG;
    const float4* rows = reinterpret_cast<const float4*>(part);
    // a row counts only if THIS frame's backward wrote it: piece 2 carries the 64-bit frame tag in its last two words (k_render_bwd)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // lane 2 of the quad holds piece 2 of every row, whose last two words are the tag: it judges the rows and tells the quad (one DPP
        // broadcast per four rows); every lane of the wave takes part in the broadcasts, lanes without work add nothing
        const bool work = !big && h < 3;
        const float4* src = rows + q0 * 3 + min(h, 2);
        const uint32_t tag2 = dwg_tag2(tag);
        const int64_t nn = work ? n : 0;
        int64_t nmax = nn;                                     // quad-uniform trip count (n is per Gaussian = per quad already)
        int64_t r = 0;
        // sixteen rows in flight per trip, added in row order (a typical splat has ~10 rows: one round trip)
        for (; r < nmax; r += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = r + u < nmax ? src[3 * (r + u)] : make_float4(0.f, 0.f, 0.f, 0.f);
            int okm = 0;
            if (h == 2) {
#pragma unroll
                for (int u = 0; u < 16; u++) okm |= (int)(__float_as_uint(v[u].z) == tag2 && __float_as_uint(v[u].w) == tag) << u;
            }
            okm = __builtin_amdgcn_update_dpp(0, okm, 0xAA, 0xF, 0xF, false);      // quad_perm [2,2,2,2]
#pragma unroll
            for (int u = 0; u < 16; u++)
                if ((okm >> u) & 1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    unsigned long long m = __ballot(big && h == 0);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int64_t wq0 = (int64_t)((uint32_t)__shfl((int)(uint32_t)q0, src)), wn = (int64_t)((uint32_t)__shfl((int)(uint32_t)n, src));
        const float4* row = rows + wq0 * 3;
        float t[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t r = lane; r < wn; r += 64) {
            const float4 a = row[3 * r], b = row[3 * r + 1], c = row[3 * r + 2];
            if (__float_as_uint(c.w) == tag && __float_as_uint(c.z) == dwg_tag2(tag)) {
                t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; t[4] += b.x; t[5] += b.y; t[6] += b.z; t[7] += b.w; t[8] += c.x; t[9] += c.y;
            }
        }
#pragma unroll
        for (int c = 0; c < 10; c++) t[c] = dwg_wave_sum_all(t[c]);
        // the splat's quad: lane src + hh takes piece hh
        if ((lane & ~3) == src && h < 3) acc = make_float4(t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3]);
    }
    if (h == 2) { acc.z = 0.f; acc.w = 0.f; }                   // the tag's slots
    if (i < G && h < 3) reinterpret_cast<float4*>(gacc + (size_t)i * GSTRIDE)[h] = acc;
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
    // frames (blockIdx.y): this frame's camera, input rows, workspaces and gradient rows ([F, G, ...] outputs)
    const size_t co = (size_t)blockIdx.y * (size_t)p.cam_stride, go = (size_t)blockIdx.y * (size_t)p.in_stride, fg = (size_t)blockIdx.y * (size_t)p.G;
    if (threadIdx.x < 16) cam[threadIdx.x] = p.view[co + threadIdx.x];
    else if (threadIdx.x < 32) cam[threadIdx.x] = p.proj[co + threadIdx.x - 16];
    camera_scalars(p, co);
    const float* campos = p.campos ? p.campos + co : nullptr;
    DWG_GEOM(rect); DWG_GEOM(rec2);
    means3D += 3 * go;
    if (shs) shs += go * (size_t)p.sh_coeffs * 3;
    if (colors) colors += 3 * go;
    if (scales) scales += 3 * go;
    if (rots) rots += 4 * go;
    if (cov3Dp) cov3Dp += 6 * go;
    gacc += fg * GSTRIDE;
    dmeans3D += 3 * fg;
    if (dmeans2D) dmeans2D += 3 * fg;
    if (dshs) dshs += fg * (size_t)p.sh_coeffs * 3;
    if (dcolors) dcolors += 3 * fg;
    if (dopac) dopac += fg;
    if (dscales) dscales += 3 * fg;
    if (drots) drots += 4 * fg;
    if (dcov3D) dcov3D += 6 * fg;
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
            float3 d = make_float3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
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

// viewmatrix = extrinsic^T, projmatrix = viewmatrix @ projection^T, campos = c2w[:3, 3] (gaussian_renderer.py:38-41): one launch
// instead of a transpose, a library 4x4 GEMM and a slice
__global__ void k_camera_setup(const float* __restrict__ extrinsic, const float* __restrict__ projection,
                               const float* __restrict__ c2w, float* __restrict__ out /*[16 + 16 + 3]*/) {
    const int t = threadIdx.x;
    if (t < 16) {
        const int r = t >> 2, c = t & 3;
        out[t] = extrinsic[4 * c + r];
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) v += extrinsic[4 * k + r] * projection[4 * c + k];     // sum_k E^T[r,k] P^T[k,c]
        out[16 + t] = v;
    } else if (t < 19) {
        out[32 + (t - 16)] = c2w[4 * (t - 16) + 3];
    }
}

// the same three products plus the field of view: out37 = [viewmatrix 16 | projmatrix 16 | campos 3 | tanfovx | tanfovy]
__global__ void k_camera_block(const float* __restrict__ extrinsic, const float* __restrict__ projection, const float* __restrict__ c2w,
                               const float* __restrict__ tanfovy, const float* __restrict__ tanfovx, float* __restrict__ out) {
    const int t = threadIdx.x;
    if (t < 16) {
        const int r = t >> 2, c = t & 3;
        out[t] = extrinsic[4 * c + r];
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) v += extrinsic[4 * k + r] * projection[4 * c + k];
        out[16 + t] = v;
    } else if (t < 19) {
        out[32 + (t - 16)] = c2w[4 * (t - 16) + 3];
    } else if (t == 19) {
        out[35] = tanfovx ? tanfovx[0] : tanfovy[0];
        out[36] = tanfovy[0];
    }
}

static int make_params(const dwg_raster_settings* cfg, const dwg_raster_frames* fr, int G, Params* p) {
    if (!cfg || G < 0 || cfg->image_height <= 0 || cfg->image_width <= 0) return DWG_E_ARG;
    if (!cfg->bg || !cfg->viewmatrix || !cfg->projmatrix) return DWG_E_ARG;
    if (fr && (fr->num_frames < 1 || fr->gaussian_stride < 0 || fr->camera_stride < 0)) return DWG_E_ARG;
    p->G = G; p->H = cfg->image_height; p->W = cfg->image_width;
    p->tiles_x = dwg_cdiv(p->W, BT); p->tiles_y = dwg_cdiv(p->H, BT);
    p->rtiles_x = dwg_cdiv(p->W, RT); p->rtiles_y = dwg_cdiv(p->H, RT);
    p->stiles_x = dwg_cdiv(p->tiles_x, ST); p->stiles_y = dwg_cdiv(p->tiles_y, ST);
    if (p->rtiles_x > 0x7fff || p->rtiles_y > 0x7fff) return DWG_E_ARG;
    if (!(cfg->tanfovx > 0.f) || !(cfg->tanfovy > 0.f)) return DWG_E_ARG;
    p->tanfovx = cfg->tanfovx; p->tanfovy = cfg->tanfovy; p->tanfov_dev = cfg->tanfov;
    p->focal_x = p->W / (2.f * cfg->tanfovx); p->focal_y = p->H / (2.f * cfg->tanfovy);
    p->scale_mod = cfg->scale_modifier;
    p->sh_degree = cfg->sh_degree; p->sh_coeffs = cfg->sh_coeffs;
    p->bg = cfg->bg; p->view = cfg->viewmatrix; p->proj = cfg->projmatrix; p->campos = cfg->campos;
    p->visit_order = cfg->visit_order;
    p->in_stride = fr ? fr->gaussian_stride : 0; p->cam_stride = fr ? fr->camera_stride : 0;
    p->geom_stride = p->pairs_stride = p->image_stride = 0;      // set by the callers that know the capacity
    static const int dbg = getenv("DWG_RASTER_DEBUG") ? atoi(getenv("DWG_RASTER_DEBUG")) : 0;
    p->dbg = dbg;
    return DWG_OK;
}

}  // namespace

// Workgroup-private LDS supertile histograms (one global atomic per workgroup and supertile instead of one per (Gaussian, supertile)
// pair) up to 16384 supertiles (4096^2 pixels); DWG_RASTER_LDS_MAXS lowers the limit (experiment switch).
// Threads per workgroup of the two per-Gaussian binning kernels: 1024 from 128 k Gaussians (a quarter of the (workgroup, supertile) global
// atomics of 256-thread workgroups: scatter 52 -> 26 us on 300 k shuffled Gaussians), 256 below (small frames need the workgroup count).
static int binning_block(int G) {
    static const int forced = getenv("DWG_RASTER_PB") ? atoi(getenv("DWG_RASTER_PB")) : 0;
    if (forced == 256 || forced == 1024) return forced;
    return G >= (1 << 17) ? 1024 : 256;
}

static int lds_hist_ok(int S) {
    static const int lds_max_s = getenv("DWG_RASTER_LDS_MAXS") ? atoi(getenv("DWG_RASTER_LDS_MAXS")) : 16384;
    return S <= lds_max_s && S <= 16384;
}

extern "C" {

int dwg_raster_workspace_sizes(int32_t G, int32_t H, int32_t W, int64_t pair_capacity, size_t* geom_bytes,
                               size_t* pairs_bytes, size_t* image_bytes) {
    if (G < 0 || H <= 0 || W <= 0 || pair_capacity < 0) return DWG_E_ARG;
    if (geom_bytes) *geom_bytes = geom_layout(G, H, W).total;
    if (pairs_bytes) *pairs_bytes = pair_layout(pair_capacity, H, W).total;
    if (image_bytes) *image_bytes = image_layout(H, W).total;
    return DWG_OK;
}

const int32_t* dwg_raster_num_pairs_ptr(const void* ws_geom) { return reinterpret_cast<const int32_t*>(ws_geom); }

int dwg_raster_camera_setup(const float* extrinsic, const float* projection, const float* c2w, float* out35, dwg_stream_t stream_) {
    if (!extrinsic || !projection || !c2w || !out35) return DWG_E_ARG;
    DWG_LAUNCH("raster_camera_setup", k_camera_setup, dim3(1), dim3(64), 0, (hipStream_t)stream_, extrinsic, projection, c2w, out35);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_camera_block(const float* extrinsic, const float* projection, const float* c2w, const float* tanfovy, const float* tanfovx,
                            float* out37, dwg_stream_t stream_) {
    if (!extrinsic || !projection || !c2w || !tanfovy || !out37) return DWG_E_ARG;
    DWG_LAUNCH("raster_camera_setup", k_camera_block, dim3(1), dim3(64), 0, (hipStream_t)stream_, extrinsic, projection, c2w, tanfovy, tanfovx, out37);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_forward_bin_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t G, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                                  const float* rotations, const float* cov3D_precomp, int32_t* radii, void* ws_geom,
                                  dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, frames, G, &p);
    if (rc) return rc;
    if (!ws_geom) return DWG_E_ARG;
    if (G > 0) {  // with G == 0 every per-Gaussian pointer may be NULL
        if (!means3D || !opacities || !radii) return DWG_E_ARG;
        if ((shs == nullptr) == (colors_precomp == nullptr)) return DWG_E_ARG;      // exactly one colour source
        if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) return DWG_E_ARG;
        if (shs && (!cfg->campos || cfg->sh_degree < 0 || cfg->sh_degree > 3 ||
                    cfg->sh_coeffs < (cfg->sh_degree + 1) * (cfg->sh_degree + 1))) return DWG_E_ARG;
    }
    const int F = frames ? frames->num_frames : 1;
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    p.geom_stride = L.total;
    char* ws = (char*)ws_geom;
    const int S = p.stiles_x * p.stiles_y;
    for (int f = 0; f < F; f++) {
        char* wf = ws + (size_t)f * L.total;
        if (hipMemsetAsync(wf + L.header, 0, L.zero_end - L.header, stream) != hipSuccess) return DWG_E_LAUNCH;
    }
    static bool attr_set = false;
    if (!attr_set) {      // 64 KiB of histogram + the static camera words is over the 64 KiB default limit
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_preprocess<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_preprocess<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_super<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_super<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
        attr_set = true;
    }
    if (G > 0) {
        const int use_lds_hist = lds_hist_ok(S);
#define DWG_PRE_ARGS p, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, (float4*)(ws + L.rec0),                  \
                     (float4*)(ws + L.rec1), (float4*)(ws + L.rec2), (uint2*)(ws + L.rect), (uint32_t*)(ws + L.npairs),                        \
                     (uint32_t*)(ws + L.idsum), (uint32_t*)(ws + L.super_count), (uint32_t*)(ws + L.kref_part), use_lds_hist
        if (binning_block(G) == 1024) DWG_LAUNCH("raster_preprocess", k_preprocess<1024>, dim3(dwg_cdiv(G, 1024), F), dim3(1024), use_lds_hist ? (size_t)S * 4 : 0, stream, DWG_PRE_ARGS);
        else DWG_LAUNCH("raster_preprocess", k_preprocess<256>, dim3(dwg_cdiv(G, 256), F), dim3(256), use_lds_hist ? (size_t)S * 4 : 0, stream, DWG_PRE_ARGS);
#undef DWG_PRE_ARGS
    }
    // workgroup 0: supertile lists (starts, size classes); workgroups 1..: the pair rows of GTILE Gaussians each + the frame's pair count
    DWG_LAUNCH("raster_scan_super", k_scan_super, dim3(1 + dwg_cdiv(G, GTILE), F), dim3(1024), 0, stream, p,
               (const uint32_t*)(ws + L.super_count), (uint32_t*)(ws + L.super_start), (uint32_t*)(ws + L.chunk_start), (int32_t*)(ws + L.header),
               (const uint32_t*)(ws + L.npairs), (const uint32_t*)(ws + L.idsum), (uint32_t*)(ws + L.goff), (const uint32_t*)(ws + L.kref_part), binning_block(G));
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_forward_bin(const dwg_raster_settings* cfg, int32_t G, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, int32_t* radii, void* ws_geom,
                           dwg_stream_t stream_) {
    return dwg_raster_forward_bin_frames(cfg, nullptr, G, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii,
                                         ws_geom, stream_);
}

int dwg_raster_forward_render_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t G, void* ws_geom,
                                     void* ws_pairs, int64_t pair_capacity, void* ws_image, float* out_color, float* out_depth,
                                     float* out_alpha, dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, frames, G, &p);
    if (rc) return rc;
    if (!ws_geom || !ws_pairs || !ws_image || !out_color || !out_depth || !out_alpha || pair_capacity < 0) return DWG_E_ARG;
    if (pair_capacity > 0xfffffff0ll) return DWG_E_ARG;       // list offsets are 32-bit
    const int F = frames ? frames->num_frames : 1;
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    PairLayout PL = pair_layout(pair_capacity, p.H, p.W);
    ImageLayout IL = image_layout(p.H, p.W);
    p.geom_stride = L.total; p.pairs_stride = PL.total; p.image_stride = IL.total;
    const int64_t cap_segs = seg_capacity(pair_capacity > 0 ? pair_capacity : 1, p.H, p.W);
    char* ws = (char*)ws_geom; char* wp = (char*)ws_pairs; char* wi = (char*)ws_image;
    const int T = p.tiles_x * p.tiles_y, S = p.stiles_x * p.stiles_y;
    uint64_t* keys = (uint64_t*)(wp + PL.keys);
    uint32_t* sorted = (uint32_t*)(wp + PL.sorted);
    uint32_t* tile_start = (uint32_t*)(ws + L.tile_start);
    uint32_t* tile_count = (uint32_t*)(ws + L.tile_count);
    const uint32_t* super_start = (const uint32_t*)(ws + L.super_start);
    const uint32_t* chunk_start = (const uint32_t*)(ws + L.chunk_start);
    uint64_t* cand = (uint64_t*)(wp + PL.cand);
    int32_t* header = (int32_t*)(ws + L.header);
    if (G > 0) {
        const int use_lds = lds_hist_ok(S);
#define DWG_SCAT_ARGS p, (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1), (const uint2*)(ws + L.rect), super_start,                      \
                      (uint32_t*)(ws + L.super_cursor), keys, pair_capacity, header, use_lds
        if (binning_block(G) == 1024) DWG_LAUNCH("raster_scatter", k_scatter_super<1024>, dim3(dwg_cdiv(G, 1024), F), dim3(1024), use_lds ? (size_t)S * 4 : 0, stream, DWG_SCAT_ARGS);
        else DWG_LAUNCH("raster_scatter", k_scatter_super<256>, dim3(dwg_cdiv(G, 256), F), dim3(256), use_lds ? (size_t)S * 4 : 0, stream, DWG_SCAT_ARGS);
#undef DWG_SCAT_ARGS
        // A frame's keys number at most its (Gaussian, block) pairs <= capacity, in at most capacity / CHUNK + S chunks (every supertile's
        // last chunk may be short); surplus workgroups exit on their first instruction.
        const int64_t max_chunks = pair_capacity / CHUNK + S;
        if (max_chunks > 0x7fffffffll) return DWG_E_ARG;
        DWG_LAUNCH("raster_chunk_sort", k_chunk_sort, dim3((unsigned)max_chunks, F), dim3(256), (CHUNK + CHUNK / 16) * 8, stream, p, chunk_start,
                   super_start, (const int32_t*)header, keys, pair_capacity);
        DWG_LAUNCH("raster_rank_merge", k_rank_merge, dim3((unsigned)max_chunks, F), dim3(256), 0, stream, p, chunk_start, super_start,
                   (const int32_t*)header, (const uint64_t*)keys, (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1),
                   (const uint2*)(ws + L.rect), cand, pair_capacity);
    }
    // sixteen waves per supertile: a dense list (up to 16 k candidates under a body) is two passes of 16 chunks per wave, not of 64; every
    // supertile publishes its blocks (list, segments, render-order bucket) -- also when there are no Gaussians at all
    DWG_LAUNCH("raster_split", k_split<1024>, dim3(S, F), dim3(1024), 0, stream, p, header, super_start, (const uint64_t*)cand, tile_count,
               tile_start, (uint32_t*)(ws + L.seg_start), (uint32_t*)(ws + L.order), sorted, pair_capacity);
    DWG_LAUNCH("raster_render_fwd", k_render_fwd, dim3(T, F), dim3(64), 0, stream, p, (const int32_t*)header, (const uint32_t*)(ws + L.order),
               (const uint32_t*)tile_start, (const uint32_t*)tile_count,
               (const uint32_t*)(ws + L.seg_start), (const uint32_t*)sorted, (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1),
               (const float4*)(ws + L.rec2), pair_capacity, cap_segs, (uint32_t*)(wp + PL.seg_tile), (float*)(wp + PL.ckpt),
               (uint32_t*)(ws + L.tile_neff), (float*)(wi + IL.final_T), (int*)(wi + IL.n_contrib), (float*)(wi + IL.craw), out_color,
               out_depth, out_alpha);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_forward_render(const dwg_raster_settings* cfg, int32_t G, void* ws_geom, void* ws_pairs,
                              int64_t pair_capacity, void* ws_image, float* out_color, float* out_depth,
                              float* out_alpha, dwg_stream_t stream_) {
    return dwg_raster_forward_render_frames(cfg, nullptr, G, ws_geom, ws_pairs, pair_capacity, ws_image, out_color, out_depth, out_alpha, stream_);
}

int dwg_raster_backward_frames(const dwg_raster_settings* cfg, const dwg_raster_frames* frames, int32_t G, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* opacities, const float* scales,
                               const float* rotations, const float* cov3D_precomp, const void* ws_geom, const void* ws_pairs,
                               int64_t pair_capacity, const void* ws_image, void* ws_grad, const float* dL_dout_color,
                               const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                               float* dL_dcolors, float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                               dwg_stream_t stream_) {
    Params p;
    int rc = make_params(cfg, frames, G, &p);
    if (rc) return rc;
    if (!ws_geom || !ws_pairs || !ws_image || !ws_grad || !dL_dout_color || !dL_dmeans3D || pair_capacity < 0)
        return DWG_E_ARG;
    (void)opacities;
    if (G == 0) return DWG_OK;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return DWG_E_ARG;
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) return DWG_E_ARG;
    const int F = frames ? frames->num_frames : 1;
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout L = geom_layout(G, p.H, p.W);
    PairLayout PL = pair_layout(pair_capacity, p.H, p.W);
    ImageLayout IL = image_layout(p.H, p.W);
    p.geom_stride = L.total; p.pairs_stride = PL.total; p.image_stride = IL.total;
    const int64_t cap_segs = seg_capacity(pair_capacity > 0 ? pair_capacity : 1, p.H, p.W);
    const char* ws = (const char*)ws_geom; const char* wp = (const char*)ws_pairs; const char* wi = (const char*)ws_image;
    // per-pair partials in pair-row order (no atomics), then each Gaussian's contiguous rows summed into ws_grad [F][G][GSTRIDE]
    // a fresh tag per backward: rows of the pair-ordered partials count only if this frame wrote them (no clearing of the buffer)
    // (the tag itself is header[H_TAG], drawn on the device by the forward's scan kernel: see g_frame_tag)
    const uint32_t tag = 0u;
    static const int gbig = getenv("DWG_RASTER_GBIG") ? atoi(getenv("DWG_RASTER_GBIG")) : 64;      // experiment switch (see k_gather_partials)
#define DWG_BWD_ARGS p, (const int32_t*)(ws + L.header), cap_segs, (const uint32_t*)(wp + PL.seg_tile), (const uint32_t*)(ws + L.seg_start),   \
        (const uint32_t*)(ws + L.tile_start), (const uint32_t*)(ws + L.tile_count), (const uint32_t*)(ws + L.tile_neff), (const uint32_t*)(wp + PL.sorted), \
        (const uint2*)(ws + L.rect), (const uint32_t*)(ws + L.goff), tag, (const float4*)(ws + L.rec0), (const float4*)(ws + L.rec1),           \
        (const float4*)(ws + L.rec2), pair_capacity, (const float*)(wp + PL.ckpt), (const float*)(wi + IL.final_T),                              \
        (const int*)(wi + IL.n_contrib), (const float*)(wi + IL.craw), dL_dout_color, dL_dout_depth, dL_dout_alpha,                              \
        (float*)(const_cast<char*>(wp) + PL.part)
    if (dL_dout_depth) DWG_LAUNCH("raster_render_bwd", k_render_bwd<true>, dim3((unsigned)cap_segs, F), dim3(64), 0, stream, DWG_BWD_ARGS);
    else DWG_LAUNCH("raster_render_bwd", k_render_bwd<false>, dim3((unsigned)cap_segs, F), dim3(64), 0, stream, DWG_BWD_ARGS);
#undef DWG_BWD_ARGS
    DWG_LAUNCH("raster_gather_bwd", k_gather_partials, dim3(dwg_cdiv(G, 64), F), dim3(256), 0, stream, G, (const uint32_t*)(ws + L.goff),
               (const float*)(wp + PL.part), pair_capacity, (const int32_t*)(ws + L.header), tag, (float*)ws_grad, gbig, p.geom_stride, p.pairs_stride);
    DWG_LAUNCH("raster_preprocess_bwd", k_preprocess_bwd, dim3(dwg_cdiv(G, 256), F), dim3(256), 0, stream, p, means3D, shs, colors_precomp,
               scales, rotations, cov3D_precomp, (const uint2*)(ws + L.rect), (const float4*)(ws + L.rec2),
               (const float*)ws_grad, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales,
               dL_drotations, dL_dcov3D);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_raster_backward(const dwg_raster_settings* cfg, int32_t G, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        const float* rotations, const float* cov3D_precomp, const void* ws_geom, const void* ws_pairs,
                        int64_t pair_capacity, const void* ws_image, void* ws_grad, const float* dL_dout_color,
                        const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                        float* dL_dscales, float* dL_drotations, float* dL_dcov3D, dwg_stream_t stream_) {
    return dwg_raster_backward_frames(cfg, nullptr, G, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, ws_geom, ws_pairs,
                                      pair_capacity, ws_image, ws_grad, dL_dout_color, dL_dout_depth, dL_dout_alpha, dL_dmeans3D, dL_dmeans2D,
                                      dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, stream_);
}

}  // extern "C"
