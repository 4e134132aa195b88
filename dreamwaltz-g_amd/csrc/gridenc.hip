// gridenc.hip -- multi-resolution hash / tiled grid encoder (D = 3, C = 2) for gfx950.
//
// Native restatement of the reference's in-repo CUDA extension
//   /root/reference/core/nerf/gridencoder/src/gridencoder.cu : get_grid_index :66-84, kernel_grid :87-242,
//   kernel_grid_backward :245-337, kernel_input_backward :340-366   (boundary B2: `_gridencoder` backend,
//   prototypes gridencoder.h:12-14, Python caller grid.py:28-96)
// with a different work decomposition: one lane per (point, level) with the 16 levels of a point in 16 adjacent
// lanes, so that the [B, L*C] feature row (128 B), the dy_dx row (384 B) and the upstream gradient row are each
// written / read as whole cache lines by a quarter-wave, while the 8 corner gathers of every lane are independent
// 8-byte loads in flight (the 50 MB table lives in the 256 MB Infinity Cache / per-XCD L2).
// Gather-bound: 16 levels x 8 corners x 8 B = 1 KiB gathered per point and direction.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_gridenc.h"

namespace {

struct GridP {
    uint32_t B, L;
    float S;
    uint32_t H, gridtype, align_corners, interp, layout;  // layout 0: [L,B,C] (reference backend), 1: [B,L*C]
};

__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align, uint32_t hashmap_size, uint32_t res,
                                               uint32_t x, uint32_t y, uint32_t z) {
    // gridencoder.cu:66-84 for D = 3
    uint32_t stride = 1, index = 0;
    const uint32_t step = align ? res : (res + 1);
    if (stride <= hashmap_size) { index += x * stride; stride *= step; }
    if (stride <= hashmap_size) { index += y * stride; stride *= step; }
    if (stride <= hashmap_size) { index += z * stride; stride *= step; }
    if (gridtype == 0 && stride > hashmap_size) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return (index % hashmap_size) * 2u;
}

struct Cell {
    bool oob;
    float scale;
    uint32_t res, hsize;
    float w[3], dw[3];
    uint32_t g[3];
};

__device__ __forceinline__ Cell locate(const GridP& p, const int* __restrict__ offsets, uint32_t level, float x0, float x1,
                                       float x2) {
    Cell c;
    c.oob = (x0 < 0.f || x0 > 1.f || x1 < 0.f || x1 > 1.f || x2 < 0.f || x2 > 1.f);
    c.hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
    c.scale = exp2f((float)level * p.S) * (float)p.H - 1.0f;
    c.res = (uint32_t)ceilf(c.scale) + 1u;
    const float xs[3] = {x0, x1, x2};
    // NB (bug-compatible with gridencoder.cu:137): pos_deriv is initialised {1, 0, 0}; smoothstep overwrites all three
    c.dw[0] = 1.f; c.dw[1] = 0.f; c.dw[2] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float pos = xs[d] * c.scale + (p.align_corners ? 0.0f : 0.5f);
        float fl = floorf(pos);
        c.g[d] = (uint32_t)fl;
        pos -= fl;
        if (p.interp == 1) { c.dw[d] = 6.f * pos * (1.f - pos); pos = pos * pos * (3.f - 2.f * pos); }
        c.w[d] = pos;
    }
    return c;
}

__global__ __launch_bounds__(256) void k_grid_fwd(GridP p, const float* __restrict__ x, const float2* __restrict__ table,
                                                  const int* __restrict__ offsets, float* __restrict__ out,
                                                  float* __restrict__ dy_dx) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t b = t / p.L, level = t - b * p.L;
    if (b >= p.B) return;
    const float x0 = x[3 * b], x1 = x[3 * b + 1], x2 = x[3 * b + 2];
    Cell c = locate(p, offsets, level, x0, x1, x2);
    float* o = p.layout ? out + (size_t)b * p.L * 2 + level * 2 : out + ((size_t)level * p.B + b) * 2;
    float* dd = dy_dx ? dy_dx + ((size_t)b * p.L + level) * 6 : nullptr;
    if (c.oob) {
        o[0] = 0.f; o[1] = 0.f;
        if (dd) { for (int k = 0; k < 6; k++) dd[k] = 0.f; }
        return;
    }
    const float2* g = table + (uint32_t)offsets[level];
    float2 v[8];
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        uint32_t gx = c.g[0] + (idx & 1), gy = c.g[1] + ((idx >> 1) & 1), gz = c.g[2] + ((idx >> 2) & 1);
        v[idx] = g[grid_index(p.gridtype, p.align_corners, c.hsize, c.res, gx, gy, gz) >> 1];
    }
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        float w = ((idx & 1) ? c.w[0] : 1.f - c.w[0]) * ((idx & 2) ? c.w[1] : 1.f - c.w[1]) * ((idx & 4) ? c.w[2] : 1.f - c.w[2]);
        r0 += w * v[idx].x; r1 += w * v[idx].y;
    }
    *reinterpret_cast<float2*>(o) = make_float2(r0, r1);
    if (dd) {
        // d out / d x_gd = scale * dsmooth(gd) * sum over the 4 corner pairs along gd (gridencoder.cu:196-240)
#pragma unroll
        for (int gd = 0; gd < 3; gd++) {
            float g0 = 0.f, g1 = 0.f;
            const int d1 = gd == 0 ? 1 : 0, d2 = gd == 2 ? 1 : 2;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int b1 = k & 1, b2 = (k >> 1) & 1;
                float w = c.scale * (b1 ? c.w[d1] : 1.f - c.w[d1]) * (b2 ? c.w[d2] : 1.f - c.w[d2]);
                int left = (b1 << d1) | (b2 << d2), right = left | (1 << gd);
                g0 += w * (v[right].x - v[left].x) * c.dw[gd];
                g1 += w * (v[right].y - v[left].y) * c.dw[gd];
            }
            dd[gd * 2] = g0; dd[gd * 2 + 1] = g1;
        }
    }
}

// grad_table += w * grad (atomics), and grad_x[b,d] = sum_{l,c} grad[b,l,c] * dy_dx[b,l,d,c]
// XCD-private accumulation (XCD = true): device-scope float atomics are executed at the memory side on this chip (the 8 XCD L2s
// are not coherent with each other): one fabric transaction per atomic, ~13 G atomics/s measured.  Instead every XCD adds into
// ITS OWN copy of the table gradient with workgroup-scope atomics, which the XCD's L2 executes in cache (all CUs of an XCD share
// that L2, and a wave never leaves its XCD: HW_REG_XCC_ID is where it physically runs).  The copies are summed (and cleared) by
// k_xcd_reduce_clear afterwards.  Which copy a contribution lands in changes only the summation order.
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]

template <bool XCD>
__device__ __forceinline__ void table_add(float* p, float v) {
    if (XCD) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(p, v);
}

template <bool XCD>
__global__ __launch_bounds__(256) void k_grid_bwd(GridP p, const float* __restrict__ grad, const float* __restrict__ x,
                                                  const int* __restrict__ offsets, float* __restrict__ grad_table,
                                                  const float* __restrict__ dy_dx, float* __restrict__ grad_x, uint32_t first_table_level,
                                                  size_t xcd_stride) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t b = t / p.L, level = t - b * p.L;
    const bool live = b < p.B;
    float g0 = 0.f, g1 = 0.f;
    float gx[3] = {0.f, 0.f, 0.f};
    if (live) {
        const float* gsrc = p.layout ? grad + (size_t)b * p.L * 2 + level * 2 : grad + ((size_t)level * p.B + b) * 2;
        g0 = gsrc[0]; g1 = gsrc[1];
        const float x0 = x[3 * b], x1 = x[3 * b + 1], x2 = x[3 * b + 2];
        Cell c = locate(p, offsets, level, x0, x1, x2);
        if (!c.oob) {
            if (grad_table && level >= first_table_level) {
                float* gt = grad_table + (size_t)(uint32_t)offsets[level] * 2 + (XCD ? xcc_id() * xcd_stride : (size_t)0);
#pragma unroll
                for (int idx = 0; idx < 8; idx++) {
                    uint32_t cx = c.g[0] + (idx & 1), cy = c.g[1] + ((idx >> 1) & 1), cz = c.g[2] + ((idx >> 2) & 1);
                    float w = ((idx & 1) ? c.w[0] : 1.f - c.w[0]) * ((idx & 2) ? c.w[1] : 1.f - c.w[1]) *
                              ((idx & 4) ? c.w[2] : 1.f - c.w[2]);
                    uint32_t index = grid_index(p.gridtype, p.align_corners, c.hsize, c.res, cx, cy, cz);
                    table_add<XCD>(gt + index, w * g0);
                    table_add<XCD>(gt + index + 1, w * g1);
                }
            }
            if (dy_dx && grad_x) {
                const float* dd = dy_dx + ((size_t)b * p.L + level) * 6;
#pragma unroll
                for (int d = 0; d < 3; d++) gx[d] = g0 * dd[2 * d] + g1 * dd[2 * d + 1];
            }
        }
    }
    if (dy_dx && grad_x) {
        // reduce over the L (<= 16, power of two lanes) levels of a point: they sit in adjacent lanes
        if (p.L == 16) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                float v = gx[d];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                gx[d] = v;
            }
            if (live && level == 0) { grad_x[3 * b] = gx[0]; grad_x[3 * b + 1] = gx[1]; grad_x[3 * b + 2] = gx[2]; }
        } else if (live) {
            atomicAdd(&grad_x[3 * b], gx[0]); atomicAdd(&grad_x[3 * b + 1], gx[1]); atomicAdd(&grad_x[3 * b + 2], gx[2]);
        }
    }
}

// XCD-OWNED table slabs: every 128-byte line of the table gradient belongs to ONE XCD (owner = line index mod 8).  Each chunk of
// 256 (point, level) lanes is visited by a workgroup on EVERY XCD; a workgroup evaluates all 8 corners but only adds to the entries
// its own XCD owns -- with workgroup-scope atomics, which that XCD's L2 executes in cache.  No line is ever cached by two XCDs, so
// there are no private copies to allocate, reduce or clear (the 8-copy variant above moved 4.9x the algorithmic traffic), and an
// XCD's share of the table (1/8 = 6.3 MB) mostly stays in its 4 MiB L2 / the Infinity Cache.  The redundant index arithmetic
// (8x) is a few tens of microseconds of VALU.  Work is handed out per PHYSICAL XCD (HW_REG_XCC_ID) through 8 counters, so
// correctness does not depend on how workgroups are placed -- only on every XCD running at least one of them (checked: `done`).
__global__ __launch_bounds__(256) void k_grid_bwd_owner(GridP p, uint32_t nchunks, const float* __restrict__ grad, const float* __restrict__ x,
                                                        const int* __restrict__ offsets, float* __restrict__ grad_table,
                                                        const float* __restrict__ dy_dx, float* __restrict__ grad_x,
                                                        uint32_t first_table_level, uint32_t* __restrict__ counters /*[8] next chunk, [8..15] done*/) {
    __shared__ uint32_t s_chunk;
    const uint32_t my = xcc_id();
    for (;;) {
        if (threadIdx.x == 0) s_chunk = atomicAdd(&counters[my], 1u);
        __syncthreads();
        const uint32_t chunk = s_chunk;
        __syncthreads();
        if (chunk >= nchunks) break;
        const uint32_t t = chunk * 256u + threadIdx.x;
        const uint32_t b = t / p.L, level = t - b * p.L;
        const bool live = b < p.B;
        const bool do_x = dy_dx && grad_x && ((chunk & 7u) == my);      // exactly one of the 8 visits of a chunk
        float gx[3] = {0.f, 0.f, 0.f};
        if (live) {
            const float* gsrc = p.layout ? grad + (size_t)b * p.L * 2 + level * 2 : grad + ((size_t)level * p.B + b) * 2;
            const float g0 = gsrc[0], g1 = gsrc[1];
            Cell c = locate(p, offsets, level, x[3 * b], x[3 * b + 1], x[3 * b + 2]);
            if (!c.oob) {
                if (level >= first_table_level) {
                    const uint32_t lvl_off = (uint32_t)offsets[level] * 2u;
#pragma unroll
                    for (int idx = 0; idx < 8; idx++) {
                        uint32_t cx = c.g[0] + (idx & 1), cy = c.g[1] + ((idx >> 1) & 1), cz = c.g[2] + ((idx >> 2) & 1);
                        const uint32_t index = lvl_off + grid_index(p.gridtype, p.align_corners, c.hsize, c.res, cx, cy, cz);
                        if (((index >> 5) & 7u) != my) continue;        // 32 floats = one 128-byte line
                        float w = ((idx & 1) ? c.w[0] : 1.f - c.w[0]) * ((idx & 2) ? c.w[1] : 1.f - c.w[1]) *
                                  ((idx & 4) ? c.w[2] : 1.f - c.w[2]);
                        table_add<true>(grad_table + index, w * g0);
                        table_add<true>(grad_table + index + 1, w * g1);
                    }
                }
                if (do_x) {
                    const float* dd = dy_dx + ((size_t)b * p.L + level) * 6;
#pragma unroll
                    for (int d = 0; d < 3; d++) gx[d] = g0 * dd[2 * d] + g1 * dd[2 * d + 1];
                }
            }
        }
        if (do_x) {
            if (p.L == 16) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float v = gx[d];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    gx[d] = v;
                }
                if (live && level == 0) { grad_x[3 * b] = gx[0]; grad_x[3 * b + 1] = gx[1]; grad_x[3 * b + 2] = gx[2]; }
            } else if (live) {
                atomicAdd(&grad_x[3 * b], gx[0]); atomicAdd(&grad_x[3 * b + 1], gx[1]); atomicAdd(&grad_x[3 * b + 2], gx[2]);
            }
        }
    }
    if (threadIdx.x == 0) atomicMax(&counters[8 + my], 1u);             // this XCD took part
}

// Coarse levels (a few thousand cells, hundreds of points per cell): the whole level table is privatised in LDS so the
// same-address contention stays on chip; one global atomic per (workgroup, touched entry) afterwards.
__global__ __launch_bounds__(256) void k_grid_bwd_coarse(GridP p, uint32_t level, uint32_t points_per_block,
                                                         const float* __restrict__ grad, const float* __restrict__ x,
                                                         const int* __restrict__ offsets, float* __restrict__ grad_table) {
    extern __shared__ float tab[];          // [hsize * 2]
    const uint32_t hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
    for (uint32_t e = threadIdx.x; e < hsize * 2; e += 256) tab[e] = 0.f;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * points_per_block, b1 = min(p.B, b0 + points_per_block);
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += 256) {
        const float* gsrc = p.layout ? grad + (size_t)b * p.L * 2 + level * 2 : grad + ((size_t)level * p.B + b) * 2;
        const float g0 = gsrc[0], g1 = gsrc[1];
        Cell c = locate(p, offsets, level, x[3 * b], x[3 * b + 1], x[3 * b + 2]);
        if (c.oob) continue;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            uint32_t cx = c.g[0] + (idx & 1), cy = c.g[1] + ((idx >> 1) & 1), cz = c.g[2] + ((idx >> 2) & 1);
            float w = ((idx & 1) ? c.w[0] : 1.f - c.w[0]) * ((idx & 2) ? c.w[1] : 1.f - c.w[1]) * ((idx & 4) ? c.w[2] : 1.f - c.w[2]);
            uint32_t index = grid_index(p.gridtype, p.align_corners, c.hsize, c.res, cx, cy, cz);
            atomicAdd(&tab[index], w * g0);
            atomicAdd(&tab[index + 1], w * g1);
        }
    }
    __syncthreads();
    float* gt = grad_table + (size_t)(uint32_t)offsets[level] * 2;
    for (uint32_t e = threadIdx.x; e < hsize * 2; e += 256) { float v = tab[e]; if (v != 0.f) atomicAdd(gt + e, v); }
}

// dst[i] += sum over the 8 XCD-private copies, which are cleared for the next use (one pass: 8 reads + 8 zero writes + 1 RMW)
__global__ __launch_bounds__(256) void k_xcd_reduce_clear(size_t n4, float4* __restrict__ scratch, size_t stride4, float4* __restrict__ dst) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = dst[i];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            float4 v = scratch[x * stride4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            scratch[x * stride4 + i] = z;
        }
        dst[i] = a;
    }
}

static int check(uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    if (D != 3 || C != 2 || L == 0 || L > 32) return DWG_E_ARG;  // the avatar's encoder: D=3, C=2, L=16
    if ((uint64_t)B * L > 0xffffff00ull) return DWG_E_ARG;
    return DWG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// SLAB-BINNED table gradient (round 2, after the PMC passes of tools/pmc_grid.sh): on this chip EVERY global float atomic is
// forwarded to the memory side -- TCC_EA0_ATOMIC == the number of atomics issued, L2 hit or not, workgroup scope or not, and a plain
// load of the line first does not change it -- so the 51 M adds of a 100 k-point backward run at the ~30 G/s of the fabric's atomic
// units (1.7 ms) whatever the XCD placement.  This path issues (almost) no global atomics: the (entry, value pair) contributions are
// BINNED by 4096-entry slab of the table (count per workgroup in LDS -> scan -> scatter 16-byte records into slab order), then one
// workgroup per slab accumulates its records in a 32 KiB LDS image with LDS atomics and writes the slab out with plain 16-byte
// stores.  Only slabs with more than GS_MAXREC records (the densest few levels) are split over several workgroups that add their
// images with global atomics.  The coarse levels keep their LDS-privatised kernel.  Needs a zero-filled gradient table (the
// untouched entries of a plain-stored slab are written as zeros again) and B * L * 8 * 16 bytes of record workspace.
// ---------------------------------------------------------------------------------------------------------------------
#define GS_SLAB 4096u          // table entries (float2) per slab: 32 KiB of LDS (four accumulate workgroups per CU)
#define GS_NWG 256u            // workgroups of the count / scatter passes (each owns a contiguous range of (point, level) chunks)
#define GS_MAXREC 8192u        // records one accumulate workgroup takes (round 4: 32768 -- the kernel's time was its longest unit: 128 trips of a
                               // one-load-in-flight loop; the extra units of a dense slab add their non-zero entries with global atomics)
struct GsUnit { uint32_t slab, begin, end, multi; };       // multi: 0 = the slab's only unit, else 1 + the slab's slot among the shared 64-bit images

// DETERMINISTIC accumulation (round 5).  A slab's contributions are summed as 64-bit FIXED-POINT integers: value x 2^s rounded once, s
// chosen per slab from the largest gradient magnitude of the call (found without atomics by the count pass) and the slab's record count so
// that no sum can leave 63 bits -- integer addition commutes, so the LDS atomics' arrival order (and the global atomics' of the few
// slabs split over several workgroups) no longer reaches the result: two runs, or a captured replay and an eager step, give the same
// bits.  Float atomics -- here until round 4, and in the reference's own kernel (gridencoder.cu:245-337 atomicAdd) -- do not.  The
// quantum is 2^-s >= gmax x 2^-61 x records: far below an fp32 ulp of anything the sum can be compared with; an entry whose whole sum is
// below ~2^-48 gmax comes out as an exact zero instead of a noise-signed denormal-scale value.
// A NON-FINITE incoming gradient must stay visible (the float-atomic path and the reference's atomicAdd let NaN / Inf reach the table
// gradient, where a GradScaler or the replica check sees it; a double -> integer conversion would turn it into 0 or a saturated value):
// the count pass folds NaN / Inf into its maximum as +Inf, the scan then marks every slab GS_NONFINITE and the writers store NaN.
#define GS_NONFINITE 0x7fffffff
__device__ __forceinline__ float gs_abs_or_inf(float v) { const float a = fabsf(v); return a <= 3.0e38f ? a : __builtin_inff(); }   // NaN, Inf -> Inf
__device__ __forceinline__ long long gs_to_fixed(float v, int s) { return __double2ll_rn(ldexp((double)v, s)); }
__device__ __forceinline__ float gs_from_fixed(long long q, int s) { return (float)ldexp((double)q, -s); }

__device__ __forceinline__ uint32_t gs_wg_chunks(uint32_t nchunks) { return (nchunks + GS_NWG - 1u) / GS_NWG; }

// passes 1 and 3 share this walk: SCATTER = false counts records per slab, true writes them at the reserved positions
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_gs_bin(GridP p, uint32_t nchunks, uint32_t nslab, uint32_t first_entry, const float* __restrict__ grad,
                                                const float* __restrict__ x, const int* __restrict__ offsets, uint32_t first_table_level,
                                                uint32_t* __restrict__ counts /*[nslab][GS_NWG]: counts, then prefixes*/,
                                                const uint32_t* __restrict__ slab_start, uint4* __restrict__ records,
                                                const float* __restrict__ dy_dx, float* __restrict__ grad_x, float* __restrict__ wgmax /*[GS_NWG]*/,
                                                const uint32_t* __restrict__ n_multi, unsigned long long* __restrict__ gimg) {
    extern __shared__ uint32_t cur[];       // [nslab]
    __shared__ float smax[4];
    const uint32_t wg = blockIdx.x;
    float gmax = 0.f;
    if (SCATTER) {      // the 64-bit images of the slabs that several accumulate workgroups share start from zero (k_gs_scan counted them)
        const size_t nz = (size_t)(*n_multi) * GS_SLAB * 2u;
        for (size_t i = (size_t)wg * 256u + threadIdx.x; i < nz; i += (size_t)GS_NWG * 256u) gimg[i] = 0ull;
    }
    for (uint32_t s_ = threadIdx.x; s_ < nslab; s_ += 256) cur[s_] = SCATTER ? slab_start[s_] + counts[(size_t)s_ * GS_NWG + wg] : 0u;
    __syncthreads();
    const uint32_t cpw = gs_wg_chunks(nchunks);
    const uint32_t c0 = wg * cpw, c1 = min(nchunks, c0 + cpw);
    for (uint32_t chunk = c0; chunk < c1; chunk++) {
        const uint32_t t = chunk * 256u + threadIdx.x;
        const uint32_t b = t / p.L, level = t - b * p.L;
        const bool live = b < p.B;
        const bool do_x = SCATTER && dy_dx && grad_x;
        float gx[3] = {0.f, 0.f, 0.f};
        if (live) {
            const float* gsrc = p.layout ? grad + (size_t)b * p.L * 2 + level * 2 : grad + ((size_t)level * p.B + b) * 2;
            const float g0 = gsrc[0], g1 = gsrc[1];
            Cell c = locate(p, offsets, level, x[3 * b], x[3 * b + 1], x[3 * b + 2]);
            if (!c.oob) {
                if (level >= first_table_level) {
                    if (!SCATTER) gmax = fmaxf(gmax, fmaxf(gs_abs_or_inf(g0), gs_abs_or_inf(g1)));      // interpolation weights are <= 1: bounds every record
                    const uint32_t lvl_entry = (uint32_t)offsets[level] - first_entry;
#pragma unroll
                    for (int idx = 0; idx < 8; idx++) {
                        uint32_t cx = c.g[0] + (idx & 1), cy = c.g[1] + ((idx >> 1) & 1), cz = c.g[2] + ((idx >> 2) & 1);
                        const uint32_t e = lvl_entry + (grid_index(p.gridtype, p.align_corners, c.hsize, c.res, cx, cy, cz) >> 1);
                        const uint32_t slab = e / GS_SLAB;
                        const uint32_t pos = atomicAdd(&cur[slab], 1u);
                        if (SCATTER) {
                            float w = ((idx & 1) ? c.w[0] : 1.f - c.w[0]) * ((idx & 2) ? c.w[1] : 1.f - c.w[1]) *
                                      ((idx & 4) ? c.w[2] : 1.f - c.w[2]);
                            records[pos] = make_uint4(e - slab * GS_SLAB, __float_as_uint(w * g0), __float_as_uint(w * g1), 0u);
                        }
                    }
                }
                if (do_x) {
                    const float* dd = dy_dx + ((size_t)b * p.L + level) * 6;
#pragma unroll
                    for (int d = 0; d < 3; d++) gx[d] = g0 * dd[2 * d] + g1 * dd[2 * d + 1];
                }
            }
        }
        if (do_x) {
            if (p.L == 16) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float v = gx[d];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    gx[d] = v;
                }
                if (live && level == 0) { grad_x[3 * b] = gx[0]; grad_x[3 * b + 1] = gx[1]; grad_x[3 * b + 2] = gx[2]; }
            } else if (live) {
                atomicAdd(&grad_x[3 * b], gx[0]); atomicAdd(&grad_x[3 * b + 1], gx[1]); atomicAdd(&grad_x[3 * b + 2], gx[2]);
            }
        }
    }
    if (!SCATTER) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = gmax;
        __syncthreads();
        if (threadIdx.x == 0) wgmax[wg] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));      // its own slot: no atomics, order-free
        for (uint32_t s_ = threadIdx.x; s_ < nslab; s_ += 256) counts[(size_t)s_ * GS_NWG + wg] = cur[s_];
    }
}

// pass 2a (one WAVE per slab): the exclusive prefix over the GS_NWG workgroups' counts of a slab (in place; the slab's 256 counts are one
// contiguous KiB: four per lane, one wave scan) and the slab's total.  Round 4: this and pass 2b were ONE workgroup walking all
// GS_NWG x nslab counts with 16 loads in flight per thread -- 50 us of dependent latency at 1500 slabs.
__global__ __launch_bounds__(256) void k_gs_scan_slab(uint32_t nslab, uint32_t* __restrict__ counts, uint32_t* __restrict__ slab_total) {
    static_assert(GS_NWG == 256u, "four counts per lane of one wave");
    const uint32_t s_ = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s_ >= nslab) return;
    uint4* row = reinterpret_cast<uint4*>(counts + (size_t)s_ * GS_NWG) + lane;
    const uint4 c = *row;
    const uint32_t mine = c.x + c.y + c.z + c.w;
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off);
        if ((int)lane >= off) inc += v;
    }
    const uint32_t ex = inc - mine;
    *row = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
    if (lane == 63u) slab_total[s_] = inc;
}

// pass 2b (one workgroup): the slabs' starts and the work units from the slab totals
__global__ __launch_bounds__(1024) void k_gs_scan(uint32_t nslab, const uint32_t* __restrict__ slab_total, uint32_t* __restrict__ slab_start /*[nslab+1]*/,
                                                  GsUnit* __restrict__ units, uint32_t* __restrict__ n_units, const float* __restrict__ wgmax,
                                                  int32_t* __restrict__ sexp /*[nslab]*/, uint32_t* __restrict__ multi_list, uint32_t* __restrict__ n_multi) {
    __shared__ uint32_t tot[1024], ucnt[1024], mcnt[1024];
    __shared__ uint32_t carry_t, carry_u, carry_m;
    __shared__ float gm[256];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { carry_t = 0u; carry_u = 0u; carry_m = 0u; }
    if (tid < 256u) gm[tid] = tid < GS_NWG ? wgmax[tid] : 0.f;
    __syncthreads();
    for (uint32_t off = 128u; off >= 1u; off >>= 1) {           // max of the count pass's per-workgroup maxima (max commutes: any order)
        if (tid < off) gm[tid] = fmaxf(gm[tid], gm[tid + off]);
        __syncthreads();
    }
    int emax = 0;
    const bool nonfinite = !(gm[0] < 3.0e38f);          // a NaN / Inf gradient somewhere in the call (gs_abs_or_inf)
    { const float g = gm[0]; if (g > 0.f && g < 3.0e38f) (void)frexpf(g, &emax); }      // g < 2^emax
    for (uint32_t base = 0; base < nslab; base += 1024u) {
        const uint32_t s_ = base + tid;
        const uint32_t run = s_ < nslab ? slab_total[s_] : 0u;
        const uint32_t nu = s_ < nslab ? (run > GS_MAXREC ? (run + GS_MAXREC - 1u) / GS_MAXREC : 1u) : 0u;
        const uint32_t isM = nu > 1u ? 1u : 0u;
        tot[tid] = run; ucnt[tid] = nu; mcnt[tid] = isM;
        __syncthreads();
        for (uint32_t off = 1; off < 1024u; off <<= 1) {            // inclusive scans of all three
            const uint32_t a = tid >= off ? tot[tid - off] : 0u, u = tid >= off ? ucnt[tid - off] : 0u, m_ = tid >= off ? mcnt[tid - off] : 0u;
            __syncthreads();
            tot[tid] += a; ucnt[tid] += u; mcnt[tid] += m_;
            __syncthreads();
        }
        const uint32_t start = carry_t + tot[tid] - run, ustart = carry_u + ucnt[tid] - nu, mslot = carry_m + mcnt[tid] - isM;
        if (s_ < nslab) {
            slab_start[s_] = start;
            // fixed-point scale of the slab: |record| < 2^emax, at most `run` of them per entry -> sums stay below 2^61
            int clog = 0; while ((1u << clog) < run && clog < 31) clog++;
            sexp[s_] = nonfinite ? GS_NONFINITE : 61 - emax - clog;
            if (isM) multi_list[mslot] = s_;
            for (uint32_t u = 0; u < nu; u++) {
                GsUnit g;
                g.slab = s_; g.begin = start + u * GS_MAXREC; g.end = min(start + run, g.begin + GS_MAXREC); g.multi = isM ? mslot + 1u : 0u;
                units[ustart + u] = g;
            }
        }
        __syncthreads();
        if (tid == 1023) { carry_t += tot[1023]; carry_u += ucnt[1023]; carry_m += mcnt[1023]; }
        __syncthreads();
    }
    if (tid == 0) { slab_start[nslab] = carry_t; *n_units = carry_u; *n_multi = carry_m; }
}

// pass 4: one workgroup per unit -- 64-bit fixed-point sums in LDS (see gs_to_fixed)
__global__ __launch_bounds__(256) void k_gs_accumulate(uint32_t first_entry, uint32_t total_entries, const GsUnit* __restrict__ units,
                                                       const uint32_t* __restrict__ n_units, const uint4* __restrict__ records,
                                                       float* __restrict__ grad_table, int accumulate, const int32_t* __restrict__ sexp,
                                                       unsigned long long* __restrict__ gimg) {
    extern __shared__ unsigned long long tabq[];          // [GS_SLAB * 2]
    if (blockIdx.x >= *n_units) return;
    const GsUnit u = units[blockIdx.x];
    const int sx = sexp[u.slab];
    if (sx == GS_NONFINITE) {                            // non-finite incoming gradient: the slab's gradient is NaN (multi-unit slabs: k_gs_finalize)
        if (!u.multi) {
            const uint32_t e0n = first_entry + u.slab * GS_SLAB, nen = min(GS_SLAB, total_entries - e0n);
            for (uint32_t e = threadIdx.x; e < nen * 2u; e += 256) grad_table[(size_t)e0n * 2 + e] = __builtin_nanf("");
        }
        return;
    }
    for (uint32_t e = threadIdx.x; e < GS_SLAB * 2u; e += 256) tabq[e] = 0ull;
    __syncthreads();
    for (uint32_t r0 = u.begin + threadIdx.x; r0 < u.end; r0 += 1024) {        // four records in flight per thread
        uint4 rec[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t r = r0 + 256u * i; rec[i] = r < u.end ? records[r] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (r0 + 256u * i < u.end) {
                atomicAdd(&tabq[2u * rec[i].x], (unsigned long long)gs_to_fixed(__uint_as_float(rec[i].y), sx));
                atomicAdd(&tabq[2u * rec[i].x + 1u], (unsigned long long)gs_to_fixed(__uint_as_float(rec[i].z), sx));
            }
        }
    }
    __syncthreads();
    const uint32_t e0 = first_entry + u.slab * GS_SLAB;                     // first table entry of this slab
    const uint32_t ne = min(GS_SLAB, total_entries - e0);
    float* dst = grad_table + (size_t)e0 * 2;
    if (u.multi) {
        // one of several workgroups of a dense slab: its non-zero sums join the slab's shared 64-bit image (integer atomics: order-free);
        // k_gs_finalize turns the image into floats
        unsigned long long* img = gimg + (size_t)(u.multi - 1u) * GS_SLAB * 2u;
        for (uint32_t e = threadIdx.x; e < ne * 2u; e += 256) { const unsigned long long v = tabq[e]; if (v) atomicAdd(img + e, v); }
        return;
    }
    if (((uintptr_t)dst & 15) == 0) {
        // this workgroup is the slab's only writer: plain 16-byte stores -- or, accumulating into a gradient buffer that already holds
        // other contributions (the flat gradient buffer of a multi-view step), a plain read-add-write of the same pieces
        for (uint32_t q = threadIdx.x; q < ne / 2u; q += 256) {             // two entries (16 bytes) per store
            float4 v = make_float4(gs_from_fixed((long long)tabq[4 * q], sx), gs_from_fixed((long long)tabq[4 * q + 1], sx),
                                   gs_from_fixed((long long)tabq[4 * q + 2], sx), gs_from_fixed((long long)tabq[4 * q + 3], sx));
            if (accumulate) { const float4 o = reinterpret_cast<const float4*>(dst)[q]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            reinterpret_cast<float4*>(dst)[q] = v;
        }
        if ((ne & 1u) && threadIdx.x == 0) {
            const float o0 = accumulate ? dst[2 * (ne - 1)] : 0.f, o1 = accumulate ? dst[2 * (ne - 1) + 1] : 0.f;
            dst[2 * (ne - 1)] = gs_from_fixed((long long)tabq[2 * (ne - 1)], sx) + o0;
            dst[2 * (ne - 1) + 1] = gs_from_fixed((long long)tabq[2 * (ne - 1) + 1], sx) + o1;
        }
    } else {
        for (uint32_t e = threadIdx.x; e < ne * 2u; e += 256) dst[e] = gs_from_fixed((long long)tabq[e], sx) + (accumulate ? dst[e] : 0.f);
    }
}

// pass 5: one workgroup per slab that several accumulate workgroups shared: its 64-bit image -> floats
__global__ __launch_bounds__(256) void k_gs_finalize(uint32_t first_entry, uint32_t total_entries, const uint32_t* __restrict__ multi_list,
                                                     const uint32_t* __restrict__ n_multi, const int32_t* __restrict__ sexp,
                                                     const unsigned long long* __restrict__ gimg, float* __restrict__ grad_table, int accumulate) {
    if (blockIdx.x >= *n_multi) return;
    const uint32_t slab = multi_list[blockIdx.x];
    const int sx = sexp[slab];
    const uint32_t e0 = first_entry + slab * GS_SLAB;
    const uint32_t ne = min(GS_SLAB, total_entries - e0);
    float* dst = grad_table + (size_t)e0 * 2;
    const unsigned long long* img = gimg + (size_t)blockIdx.x * GS_SLAB * 2u;
    if (sx == GS_NONFINITE) {
        for (uint32_t e = threadIdx.x; e < ne * 2u; e += 256) dst[e] = __builtin_nanf("");
        return;
    }
    for (uint32_t e = threadIdx.x; e < ne * 2u; e += 256) dst[e] = gs_from_fixed((long long)img[e], sx) + (accumulate ? dst[e] : 0.f);
}

}  // namespace

extern "C" {

int dwg_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                            uint32_t gridtype, uint32_t align_corners, uint32_t interp, uint32_t out_layout,
                            dwg_stream_t stream) {
    int rc = check(B, D, C, L);
    if (rc) return rc;
    if (B == 0) return DWG_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return DWG_E_ARG;
    GridP p{B, L, S, H, gridtype, align_corners, interp, out_layout};
    uint32_t n = B * L;
    DWG_LAUNCH("grid_fwd", k_grid_fwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, inputs,
                       (const float2*)embeddings, offsets, outputs, dy_dx);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

static int grid_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                         float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                         const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                         uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, float* xcd_scratch, dwg_stream_t stream) {
    int rc = check(B, D, C, L);
    if (rc) return rc;
    if (B == 0) return DWG_OK;
    (void)embeddings;
    if (!grad || !inputs || !offsets) return DWG_E_ARG;
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return DWG_E_ARG;
    GridP p{B, L, S, H, gridtype, align_corners, interp, grad_layout};
    uint32_t n = B * L;
    if (grad_inputs && L != 16) {
        if (hipMemsetAsync(grad_inputs, 0, (size_t)B * 3 * sizeof(float), (hipStream_t)stream) != hipSuccess) return DWG_E_LAUNCH;
    }
    // levels whose table fits LDS (<= 19 000 entries) and that are heavily shared (B >> entries) take the privatised path
    uint32_t first_table_level = 0;
    if (grad_embeddings && host_offsets && B >= 16384) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_bwd_coarse), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            attr_set = true;
        }
        while (first_table_level < L) {
            uint32_t hs = (uint32_t)(host_offsets[first_table_level + 1] - host_offsets[first_table_level]);
            if ((size_t)hs * 8 > 152 * 1024 || (uint64_t)B * 8 < (uint64_t)hs * 16) break;
            uint32_t ppb = hs * 8 > 64 * 1024 ? 8192 : 2048;
            DWG_LAUNCH("grid_bwd_coarse", k_grid_bwd_coarse, dim3((B + ppb - 1) / ppb), dim3(256), (size_t)hs * 8, (hipStream_t)stream, p,
                       first_table_level, ppb, grad, inputs, offsets, grad_embeddings);
            first_table_level++;
        }
    }
    if (xcd_scratch && grad_embeddings && host_offsets && first_table_level < L) {
        const size_t total = (size_t)(uint32_t)host_offsets[L] * 2;                    // floats per table copy (multiple of 16)
        DWG_LAUNCH("grid_bwd", (k_grid_bwd<true>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, grad, inputs, offsets,
                   xcd_scratch, dy_dx, grad_inputs, first_table_level, total);
        const size_t lo = (size_t)(uint32_t)host_offsets[first_table_level] * 2;       // coarse levels went through LDS straight to dst
        const size_t n4 = (total - lo) / 4;
        size_t blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096;
        DWG_LAUNCH("grid_bwd_xcd_reduce", k_xcd_reduce_clear, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4,
                   reinterpret_cast<float4*>(xcd_scratch + lo), total / 4, reinterpret_cast<float4*>(grad_embeddings + lo));
    } else {
        DWG_LAUNCH("grid_bwd", (k_grid_bwd<false>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, grad, inputs, offsets,
                   grad_embeddings, dy_dx, grad_inputs, first_table_level, (size_t)0);
    }
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_grid_encode_backward_owner(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                   float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                   uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, uint32_t* xcd_counters,
                                   dwg_stream_t stream) {
    int rc = check(B, D, C, L);
    if (rc) return rc;
    if (B == 0) return DWG_OK;
    (void)embeddings;
    if (!grad || !inputs || !offsets || !grad_embeddings || !host_offsets || !xcd_counters) return DWG_E_ARG;
    if (((uintptr_t)grad_embeddings % 128) != 0) return DWG_E_ARG;          // line ownership assumes a line-aligned table
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return DWG_E_ARG;
    GridP p{B, L, S, H, gridtype, align_corners, interp, grad_layout};
    if (grad_inputs && L != 16) {
        if (hipMemsetAsync(grad_inputs, 0, (size_t)B * 3 * sizeof(float), (hipStream_t)stream) != hipSuccess) return DWG_E_LAUNCH;
    }
    if (hipMemsetAsync(xcd_counters, 0, 16 * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) return DWG_E_LAUNCH;
    uint32_t first_table_level = 0;
    hipStream_t main_stream = (hipStream_t)stream, side = nullptr;
    hipEvent_t ev_join = nullptr;
    if (B >= 16384) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_bwd_coarse), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            attr_set = true;
        }
        // The LDS-privatised coarse levels are a handful of fat workgroups (13 - 49 of them, up to 97 KiB of LDS each) that touch
        // their own part of the table gradient: they run on a library-owned side stream NEXT TO the fine-level kernel instead of
        // in front of it (0.27 ms of a 1.9 ms backward otherwise spent on a mostly idle chip).  DWG_GRID_SERIAL_COARSE=1: one stream.
        static const bool serial = getenv("DWG_GRID_SERIAL_COARSE") != nullptr;
        static hipStream_t side_streams[16] = {nullptr};
        static hipEvent_t ev_forks[16] = {nullptr}, ev_joins[16] = {nullptr};
        int dev = 0;
        hipEvent_t ev_fork = nullptr;
        if (!serial && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
            if (!side_streams[dev]) {
                if (hipStreamCreateWithFlags(&side_streams[dev], hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&ev_forks[dev], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ev_joins[dev], hipEventDisableTiming) != hipSuccess) return DWG_E_LAUNCH;
            }
            side = side_streams[dev]; ev_fork = ev_forks[dev]; ev_join = ev_joins[dev];
        }
        bool forked = false;
        while (first_table_level < L) {
            uint32_t hs = (uint32_t)(host_offsets[first_table_level + 1] - host_offsets[first_table_level]);
            if ((size_t)hs * 8 > 152 * 1024 || (uint64_t)B * 8 < (uint64_t)hs * 16) break;
            uint32_t ppb = hs * 8 > 64 * 1024 ? 8192 : 2048;
            if (side && !forked) {          // everything queued on the caller's stream so far (zeroed gradients, inputs) happens first
                if (hipEventRecord(ev_fork, main_stream) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess) return DWG_E_LAUNCH;
                forked = true;
            }
            DWG_LAUNCH("grid_bwd_coarse", k_grid_bwd_coarse, dim3((B + ppb - 1) / ppb), dim3(256), (size_t)hs * 8, side ? side : main_stream, p,
                       first_table_level, ppb, grad, inputs, offsets, grad_embeddings);
            first_table_level++;
        }
        if (!forked) side = nullptr;
        else if (hipEventRecord(ev_join, side) != hipSuccess) return DWG_E_LAUNCH;
    }
    const uint32_t nchunks = (uint32_t)(((uint64_t)B * L + 255) / 256);
    uint32_t blocks = nchunks * 8u; if (blocks > 2048u) blocks = 2048u;       // persistent: 8 per CU, chunks pulled per physical XCD
    if (blocks < 64u) blocks = 64u;                                          // enough that every XCD receives workgroups
    DWG_LAUNCH("grid_bwd", k_grid_bwd_owner, dim3(blocks), dim3(256), 0, main_stream, p, nchunks, grad, inputs, offsets,
               grad_embeddings, dy_dx, grad_inputs, first_table_level, xcd_counters);
    if (side && hipStreamWaitEvent(main_stream, ev_join, 0) != hipSuccess) return DWG_E_LAUNCH;       // join: later work sees both
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

size_t dwg_grid_backward_slabs_workspace_bytes(uint32_t B, uint32_t L, uint32_t total_entries) {
    const size_t nslab = (total_entries + GS_SLAB - 1u) / GS_SLAB;
    const size_t recs = (size_t)B * L * 8;
    const size_t max_units = nslab + recs / GS_MAXREC + 2;
    const size_t max_multi = (nslab < recs / GS_MAXREC + 1 ? nslab : recs / GS_MAXREC + 1);      // slabs with more than GS_MAXREC records
    return dwg_align_up(recs * sizeof(uint4), 256) + dwg_align_up((size_t)GS_NWG * nslab * 4, 256) + 2 * dwg_align_up((nslab + 1) * 4, 256) +
           dwg_align_up(max_units * sizeof(GsUnit), 256) + 256 +
           dwg_align_up((size_t)GS_NWG * 4, 256) + dwg_align_up(nslab * 4, 256) + dwg_align_up(max_multi * 4, 256) + 256 +
           dwg_align_up(max_multi * GS_SLAB * 2 * 8, 256);
}

static int grid_backward_slabs(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                               float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                               const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                               uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, void* workspace,
                               size_t workspace_bytes, dwg_stream_t stream, int accumulate) {
    int rc = check(B, D, C, L);
    if (rc) return rc;
    if (B == 0) return DWG_OK;
    (void)embeddings;
    if (!grad || !inputs || !offsets || !host_offsets || !grad_embeddings || !workspace) return DWG_E_ARG;
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return DWG_E_ARG;
    const uint32_t total_entries = (uint32_t)host_offsets[L];
    if (workspace_bytes < dwg_grid_backward_slabs_workspace_bytes(B, L, total_entries)) return DWG_E_CAPACITY;
    hipStream_t st = (hipStream_t)stream;
    GridP p{B, L, S, H, gridtype, align_corners, interp, grad_layout};
    if (grad_inputs && L != 16) {
        if (hipMemsetAsync(grad_inputs, 0, (size_t)B * 3 * sizeof(float), st) != hipSuccess) return DWG_E_LAUNCH;
    }
    // ALL levels go through the slabs by default: the dense coarse levels become a few oversubscribed slabs (split into units that add
    // their LDS images with atomics) and cost nothing extra in the accumulate pass, where their separate LDS-privatised kernel was two
    // poorly parallel launches of 0.13 ms each (c2 404 -> 464 steps/s).  DWG_GRID_SLAB_COARSE=1 restores that kernel for them.
    uint32_t first_table_level = 0;
    static const bool coarse = getenv("DWG_GRID_SLAB_COARSE") != nullptr;
    if (B >= 16384 && coarse) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_bwd_coarse), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            attr_set = true;
        }
        while (first_table_level < L) {
            uint32_t hs = (uint32_t)(host_offsets[first_table_level + 1] - host_offsets[first_table_level]);
            if ((size_t)hs * 8 > 152 * 1024 || (uint64_t)B * 8 < (uint64_t)hs * 16) break;
            uint32_t ppb = hs * 8 > 64 * 1024 ? 8192 : 2048;
            DWG_LAUNCH("grid_bwd_coarse", k_grid_bwd_coarse, dim3((B + ppb - 1) / ppb), dim3(256), (size_t)hs * 8, st, p, first_table_level, ppb,
                       grad, inputs, offsets, grad_embeddings);
            first_table_level++;
        }
    }
    const uint32_t nchunks = (uint32_t)(((uint64_t)B * L + 255) / 256);
    if (first_table_level >= L) {           // nothing left for the table pass; the input gradient still needs its walk
        first_table_level = L;
    }
    const uint32_t first_entry = (uint32_t)host_offsets[first_table_level < L ? first_table_level : L];
    const uint32_t nslab = (total_entries - first_entry + GS_SLAB - 1u) / GS_SLAB > 0 ? (total_entries - first_entry + GS_SLAB - 1u) / GS_SLAB : 1u;
    const size_t recs = (size_t)B * L * 8;
    const size_t max_units = (size_t)nslab + recs / GS_MAXREC + 2;
    // limits of this path (callers fall back to dwg_grid_encode_backward): the binning kernels keep one u32 counter per slab in LDS
    // (64 KiB without an opt-in: tables up to 67 M entries) and record positions are u32
    if ((size_t)nslab * 4 > 64 * 1024 || recs > 0xffffffffull) return DWG_E_CAPACITY;
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    uint4* records = reinterpret_cast<uint4*>(w); w += dwg_align_up(recs * sizeof(uint4), 256);
    uint32_t* counts = reinterpret_cast<uint32_t*>(w); w += dwg_align_up((size_t)GS_NWG * ((total_entries + GS_SLAB - 1u) / GS_SLAB) * 4, 256);
    uint32_t* slab_start = reinterpret_cast<uint32_t*>(w); w += dwg_align_up(((size_t)(total_entries + GS_SLAB - 1u) / GS_SLAB + 1) * 4, 256);
    uint32_t* slab_total = reinterpret_cast<uint32_t*>(w); w += dwg_align_up(((size_t)(total_entries + GS_SLAB - 1u) / GS_SLAB + 1) * 4, 256);
    GsUnit* units = reinterpret_cast<GsUnit*>(w); w += dwg_align_up(((size_t)(total_entries + GS_SLAB - 1u) / GS_SLAB + recs / GS_MAXREC + 2) * sizeof(GsUnit), 256);
    uint32_t* n_units = reinterpret_cast<uint32_t*>(w); w += 256;
    const size_t nslab_all = ((size_t)total_entries + GS_SLAB - 1u) / GS_SLAB;
    const size_t max_multi = (nslab_all < recs / GS_MAXREC + 1 ? nslab_all : recs / GS_MAXREC + 1);
    float* wgmax = reinterpret_cast<float*>(w); w += dwg_align_up((size_t)GS_NWG * 4, 256);
    int32_t* sexp = reinterpret_cast<int32_t*>(w); w += dwg_align_up(nslab_all * 4, 256);
    uint32_t* multi_list = reinterpret_cast<uint32_t*>(w); w += dwg_align_up(max_multi * 4, 256);
    uint32_t* n_multi = reinterpret_cast<uint32_t*>(w); w += 256;
    unsigned long long* gimg = reinterpret_cast<unsigned long long*>(w);
    static bool attr2 = false;
    if (!attr2) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gs_accumulate), hipFuncAttributeMaxDynamicSharedMemorySize, GS_SLAB * 16);
        attr2 = true;
    }
    DWG_LAUNCH("grid_bwd_count", (k_gs_bin<false>), dim3(GS_NWG), dim3(256), (size_t)nslab * 4, st, p, nchunks, nslab, first_entry, grad, inputs,
               offsets, first_table_level, counts, (const uint32_t*)slab_start, records, dy_dx, grad_inputs, wgmax, (const uint32_t*)n_multi, gimg);
    DWG_LAUNCH("grid_bwd_scan", k_gs_scan_slab, dim3((nslab + 3u) / 4u), dim3(256), 0, st, nslab, counts, slab_total);
    DWG_LAUNCH("grid_bwd_scan", k_gs_scan, dim3(1), dim3(1024), 0, st, nslab, (const uint32_t*)slab_total, slab_start, units, n_units,
               (const float*)wgmax, sexp, multi_list, n_multi);
    DWG_LAUNCH("grid_bwd_scatter", (k_gs_bin<true>), dim3(GS_NWG), dim3(256), (size_t)nslab * 4, st, p, nchunks, nslab, first_entry, grad, inputs,
               offsets, first_table_level, counts, (const uint32_t*)slab_start, records, dy_dx, grad_inputs, wgmax, (const uint32_t*)n_multi, gimg);
    DWG_LAUNCH("grid_bwd", k_gs_accumulate, dim3((unsigned)max_units), dim3(256), (size_t)GS_SLAB * 16, st, first_entry, total_entries,
               (const GsUnit*)units, (const uint32_t*)n_units, (const uint4*)records, grad_embeddings, accumulate, (const int32_t*)sexp, gimg);
    DWG_LAUNCH("grid_bwd", k_gs_finalize, dim3((unsigned)max_multi), dim3(256), 0, st, first_entry, total_entries, (const uint32_t*)multi_list,
               (const uint32_t*)n_multi, (const int32_t*)sexp, (const unsigned long long*)gimg, grad_embeddings, accumulate);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_grid_encode_backward_slabs(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                   float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                   uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, void* workspace,
                                   size_t workspace_bytes, dwg_stream_t stream) {
    return grid_backward_slabs(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners,
                               interp, grad_layout, host_offsets, workspace, workspace_bytes, stream, 0);
}

int dwg_grid_encode_backward_slabs_accumulate(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                              float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                              const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                              uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, void* workspace,
                                              size_t workspace_bytes, dwg_stream_t stream) {
    return grid_backward_slabs(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners,
                               interp, grad_layout, host_offsets, workspace, workspace_bytes, stream, 1);
}

int dwg_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                             float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                             uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, dwg_stream_t stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                         align_corners, interp, grad_layout, host_offsets, nullptr, stream);
}

int dwg_grid_encode_backward_xcd(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 const float* dy_dx, float* grad_inputs, uint32_t gridtype, uint32_t align_corners,
                                 uint32_t interp, uint32_t grad_layout, const int32_t* host_offsets, float* xcd_scratch,
                                 dwg_stream_t stream) {
    if (!xcd_scratch || !host_offsets || ((uintptr_t)xcd_scratch % 16) || (grad_embeddings && ((uintptr_t)grad_embeddings % 16)))
        return DWG_E_ARG;
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                         align_corners, interp, grad_layout, host_offsets, xcd_scratch, stream);
}

}  // extern "C"
