// nn_norm.hip -- bandwidth-bound layers of the SD-1.5 denoiser / VAE encoder on NHWC bf16 activations, fp32 statistics:
//   GroupNorm(32) (+ fused SiLU) forward and input-gradient, LayerNorm forward, GEGLU, row softmax forward/backward.
// They sit between the MFMA contractions of boundary B4 (controlnet.py:83-114, vae.py:34-40); the reference reaches the
// same math through torch.nn.functional.group_norm / layer_norm / gelu / softmax inside diffusers.
// Access pattern: every kernel walks the tensor in 16-byte (8 x bf16) chunks, whole rows per wave, so HBM sees full lines.
#include "dwg_common.h"
#include <cstdlib>
#include "dwg_prof_internal.h"
#include "../../include/dwg_nn.h"
#include "dwg_xfmt.h"

namespace {

// element type T of the activations: __bf16 (default plans), _Float16 (the reference's --optim.fp16 storage), float (its GS-stage fp32), dwg_xs
// (the split-precision f32x plans: fp32-grade values as hi + lo fp16 halves, 32 bytes per 8 channels -- dwg_xfmt.h)
template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));

// eight consecutive channels of one activation row in registers; every kernel below walks its tensor in these units
template <typename T> struct V8 {
    vec8<T> v;
    __device__ __forceinline__ static V8 load(const T* p) { V8 r; r.v = *reinterpret_cast<const vec8<T>*>(p); return r; }
    __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<vec8<T>*>(p) = v; }
    __device__ __forceinline__ float get(int e) const { return (float)v[e]; }
    __device__ __forceinline__ void set(int e, float z) { v[e] = (T)z; }
};
template <> struct V8<dwg_xs> {
    dwg_x8 v;
    __device__ __forceinline__ static V8 load(const dwg_xs* p) { V8 r; r.v = dwg_x8::load(p); return r; }
    __device__ __forceinline__ void store(dwg_xs* p) const { v.store(p); }
    __device__ __forceinline__ float get(int e) const { return v.get(e); }
    __device__ __forceinline__ void set(int e, float z) { v.set(e, z); }
};
// single elements of a row (the row softmax kernels): index in logical elements from a row start
template <typename T> __device__ __forceinline__ float ld1(const T* row, long long j) { return (float)row[j]; }
template <typename T> __device__ __forceinline__ void st1(T* row, long long j, float v) { row[j] = (T)v; }
template <> __device__ __forceinline__ float ld1<dwg_xs>(const dwg_xs* row, long long j) { return dwg_x_get1(row, j); }
template <> __device__ __forceinline__ void st1<dwg_xs>(dwg_xs* row, long long j, float v) { dwg_x_put1(row, j, v); }

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) { float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm statistics: sums[b][g] = (sum x, sum x^2)  |  backward: (sum dyh, sum dyh*xhat)
// grid (chunks, B); each block owns a contiguous pixel range of one image and ALL channels (full-row, coalesced reads).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void k_gn_reduce(int HW, int C, int G, int pix_per_block, const T* __restrict__ x,
                                                   const T* __restrict__ dy, const float* __restrict__ stats,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                   float eps, float* __restrict__ sums /*[B][G][2]*/, int deep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [rows][Cb*2] partials for the current channel pass
    __shared__ float gacc[64 * 2];
    __shared__ float gstat[64 * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    const int C8 = C / 8, cg = C / G;
    const float inv_n = 1.f / ((float)HW * cg);
    if (tid < 2 * G) gacc[tid] = 0.f;
    if (BWD && tid < G) {
        float m = stats[((size_t)b * G + tid) * 2] * inv_n;
        float var = stats[((size_t)b * G + tid) * 2 + 1] * inv_n - m * m;
        gstat[2 * tid] = m; gstat[2 * tid + 1] = rsqrtf(fmaxf(var, 0.f) + eps);
    }
    __syncthreads();
    for (int cb = 0; cb < C8; cb += 256) {
        const int tpx = min(256, C8 - cb);      // threads along the channel-chunk axis
        const int rows = 256 / tpx;             // pixels handled in parallel
        const int cc = cb + tid % tpx, prow = tid / tpx;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { s1[e] = 0.f; s2[e] = 0.f; }
        if (prow < rows) {
            float ga[8], be[8], mu[8], rs[8];
            if (BWD) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    int ch = cc * 8 + e, g = ch / cg;
                    ga[e] = gamma[ch]; be[e] = beta[ch]; mu[e] = gstat[2 * g]; rs[e] = gstat[2 * g + 1];
                }
            }
            // Long pixel ranges (the VAE's 128^2 .. 512^2 tensors: tens of trips) keep FOUR independent 16-byte loads in flight per thread:
            // with one load per trip the pass is bound by memory latency (~1.5 TB/s measured), not by HBM.  Short ranges (every denoiser
            // tensor: 1-2 trips) take the plain loop below -- unrolling THAT one made all of them slower (rejected list).
            int p = p0 + prow;
            if (!BWD) {
                for (; deep && p + 3 * rows < p1; p += 4 * rows) {
                    const T* xp = x + ((size_t)b * HW + p) * C + (size_t)cc * 8;
                    const size_t st = (size_t)rows * C;
                    const V8<T> x0 = V8<T>::load(xp), x1 = V8<T>::load(xp + st),
                                  x2 = V8<T>::load(xp + 2 * st), x3 = V8<T>::load(xp + 3 * st);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float a = x0.get(e), bq = x1.get(e), c = x2.get(e), d = x3.get(e);
                        s1[e] += (a + bq) + (c + d); s2[e] += (a * a + bq * bq) + (c * c + d * d);
                    }
                }
            } else if (deep && p < p1) {
                // backward statistics: the next trip's two loads are issued before this trip's arithmetic (software pipeline; unrolling the
                // arithmetic itself costs 70 more registers and halves the occupancy)
                size_t off = ((size_t)b * HW + p) * C + (size_t)cc * 8;
                const size_t st = (size_t)rows * C;
                V8<T> xc = V8<T>::load(x + off), dc = V8<T>::load(dy + off);
                for (; p < p1; p += rows) {
                    V8<T> xn = xc, dn = dc;
                    if (p + rows < p1) { xn = V8<T>::load(x + off + st); dn = V8<T>::load(dy + off + st); }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float xh = (xc.get(e) - mu[e]) * rs[e];
                        float g = dc.get(e);
                        if (silu) g *= silu_grad(xh * ga[e] + be[e]);
                        g *= ga[e];
                        s1[e] += g; s2[e] += g * xh;
                    }
                    xc = xn; dc = dn; off += st;
                }
            }
            for (; p < p1; p += rows) {
                size_t off = ((size_t)b * HW + p) * C + (size_t)cc * 8;
                V8<T> xv = V8<T>::load(x + off);
                if (!BWD) {
#pragma unroll
                    for (int e = 0; e < 8; e++) { float v = xv.get(e); s1[e] += v; s2[e] += v * v; }
                } else {
                    V8<T> dv = V8<T>::load(dy + off);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float xh = (xv.get(e) - mu[e]) * rs[e];
                        float g = dv.get(e);
                        if (silu) g *= silu_grad(xh * ga[e] + be[e]);
                        g *= ga[e];
                        s1[e] += g; s2[e] += g * xh;
                    }
                }
            }
        }
        // per-channel partials -> LDS -> per-group sums
        const int Cb = tpx * 8;
        if (prow < rows) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                lds[(prow * Cb + (tid % tpx) * 8 + e) * 2] = s1[e];
                lds[(prow * Cb + (tid % tpx) * 8 + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        // one thread per channel of this pass: sum over rows (kept in row 0 of its own column) ...
        for (int c = tid; c < Cb; c += 256) {
            float a = 0.f, q = 0.f;
            for (int r = 0; r < rows; r++) { a += lds[(r * Cb + c) * 2]; q += lds[(r * Cb + c) * 2 + 1]; }
            lds[c * 2] = a; lds[c * 2 + 1] = q;
        }
        __syncthreads();
        // ... then one thread per (group, statistic) adds the group's channels of this pass IN CHANNEL ORDER: no LDS float atomics, whose
        // arrival order -- and with it the last bit of the statistics -- varied between launches (invisible after a bf16 rounding, visible
        // in the fp32-grade f32x plans, whose hipGraph replay must equal the eager run bit for bit)
        if (tid < 2 * G) {
            const int g = tid >> 1, st = tid & 1;
            const int c_lo = max(g * cg, cb * 8) - cb * 8, c_hi = min((g + 1) * cg, cb * 8 + Cb) - cb * 8;
            float a = 0.f;
            for (int c = c_lo; c < c_hi; c++) a += lds[c * 2 + st];
            if (c_hi > c_lo) gacc[tid] += a;
        }
        __syncthreads();
    }
    // per-block partial sums, reduced in a FIXED order by k_gn_finalize: deterministic, no global atomics, no memset
    if (tid < 2 * G) sums[((size_t)b * 2 * G + tid) * gridDim.x + blockIdx.x] = gacc[tid];
}

// stats[b][o] = sum over the chunk partials [b][o][chunk], in a FIXED order: one wave per output, lane-strided loads
// (all chunks of an output are in flight at once) + a DPP tree -> deterministic and latency-flat.
__global__ __launch_bounds__(256) void k_gn_finalize(int chunks, int G, const float* __restrict__ partials, float* __restrict__ stats) {
    // grid (ceil(2G / 4), B): four outputs per workgroup -- with up to 2048 partials per output (the 512^2 VAE tensors) ONE workgroup per
    // image was a 25-us serial tail; the sum order per output is unchanged (lane-strided, then the DPP tree): still bit-reproducible
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave;
    if (o >= 2 * G) return;
    const float* src = partials + ((size_t)b * 2 * G + o) * chunks;
    float s = 0.f;
    for (int c = lane; c < chunks; c += 64) s += src[c];
    s = dwg_wave_sum_to_lane63(s);
    if (lane == 63) stats[(size_t)b * G * 2 + o] = s;
}

// Small tensors (the denoiser's 8x8 / 16x16 / 32x32 levels): the WHOLE layer in one launch.  grid (G / gb, B): a workgroup owns gb whole
// groups of one image -- gb * (C / G) channels, a multiple of 8, i.e. whole 16-byte chunks -- and walks that channel bundle over all pixels
// twice (the second time out of L2): column sums -> fixed-order tree -> mean / rstd -> normalise.  Replaces statistics + (finalize) + apply,
// three dependent ~5 us launches whose data would fit in one CU's registers.  Few workgroups (16-64) on purpose: the work is latency, not
// bandwidth.  Bit-reproducible (no atomics, fixed summation order); not bit-identical to the three-launch path (another order).
template <typename T>
__global__ __launch_bounds__(256) void k_gn_small(int HW, int C, int G, int gb, const T* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int silu, float eps, T* __restrict__ y,
                                                  float* __restrict__ stats_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][tp][nb] per-lane column partials | [2][nb] column sums
    __shared__ float gsum[16], gstat[16];
    const int b = blockIdx.y, g0 = blockIdx.x * gb, tid = threadIdx.x;
    const int cg = C / G, nb = gb * cg, nc = nb / 8, tp = 256 / nc;
    const int pl = tid / nc, ci = tid - pl * nc;
    const bool active = pl < tp;
    const size_t base = (size_t)b * HW * C + (size_t)g0 * cg + ci * 8;
    const T* xb = x + base;
    const long long st = (long long)tp * C;                       // element stride of one trip
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] = 0.f; q[e] = 0.f; }
    if (active) {
        int p = pl;
        const T* xp = xb + (size_t)p * C;
        for (; p + 7 * tp < HW; p += 8 * tp, xp += 8 * st) {       // eight independent loads in flight: the walk is L2 latency, nothing else
            V8<T> xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) xv[u] = V8<T>::load(xp + u * st);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int e = 0; e < 8; e++) { const float a = xv[u].get(e); s[e] += a; q[e] = fmaf(a, a, q[e]); }
        }
        for (; p + 3 * tp < HW; p += 4 * tp, xp += 4 * st) {
            const V8<T> x0 = V8<T>::load(xp), x1 = V8<T>::load(xp + st),
                          x2 = V8<T>::load(xp + 2 * st), x3 = V8<T>::load(xp + 3 * st);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float a0 = x0.get(e), a1 = x1.get(e), a2 = x2.get(e), a3 = x3.get(e);
                s[e] += (a0 + a1) + (a2 + a3);
                q[e] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        for (; p < HW; p += tp, xp += st) {
            const V8<T> xv = V8<T>::load(xp);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float a = xv.get(e); s[e] += a; q[e] = fmaf(a, a, q[e]); }
        }
    }
    float* cs = lds; float* cq = lds + tp * nb; float* col = lds + 2 * tp * nb;
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; e++) { cs[pl * nb + ci * 8 + e] = s[e]; cq[pl * nb + ci * 8 + e] = q[e]; }
    }
    __syncthreads();
    for (int t = tid; t < 2 * nb; t += 256) {                      // column sums over the tp pixel lanes, in lane order
        const int stt = t >= nb, c = t - stt * nb;
        const float* src = (stt ? cq : cs) + c;
        float a = 0.f;
        for (int r = 0; r < tp; r++) a += src[r * nb];
        col[t] = a;
    }
    __syncthreads();
    if (tid < 2 * gb) {                                            // group sums over the group's columns, in channel order
        const int stt = tid >= gb, gl = tid - stt * gb;
        const float* src = col + stt * nb + gl * cg;
        float a = 0.f;
        for (int c = 0; c < cg; c++) a += src[c];
        gsum[tid] = a;
        stats_out[((size_t)b * G + g0 + gl) * 2 + stt] = a;
    }
    __syncthreads();
    if (tid < gb) {
        const float inv_n = 1.f / ((float)HW * cg);
        const float m = gsum[tid] * inv_n, var = gsum[gb + tid] * inv_n - m * m;
        gstat[2 * tid] = m; gstat[2 * tid + 1] = rsqrtf(fmaxf(var, 0.f) + eps);
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int cl = ci * 8 + e, gl = cl / cg, ch = g0 * cg + cl;
        const float rs = gstat[2 * gl + 1];
        sc[e] = gamma[ch] * rs; sh[e] = beta[ch] - gstat[2 * gl] * sc[e];
    }
    T* yb = y + base;
    int p = pl;
    const T* xp = xb + (size_t)p * C; T* yp = yb + (size_t)p * C;
    for (; p + 7 * tp < HW; p += 8 * tp, xp += 8 * st, yp += 8 * st) {
        V8<T> xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = V8<T>::load(xp + u * st);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            V8<T> o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float z = fmaf(xv[u].get(e), sc[e], sh[e]);
                if (silu) z = silu_f(z);
                o.set(e, z);
            }
            o.store(yp + u * st);
        }
    }
    for (; p + 3 * tp < HW; p += 4 * tp, xp += 4 * st, yp += 4 * st) {
        V8<T> xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = V8<T>::load(xp + u * st);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            V8<T> o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float z = fmaf(xv[u].get(e), sc[e], sh[e]);
                if (silu) z = silu_f(z);
                o.set(e, z);
            }
            o.store(yp + u * st);
        }
    }
    for (; p < HW; p += tp, xp += st, yp += st) {
        const V8<T> xv = V8<T>::load(xp);
        V8<T> o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float z = fmaf(xv.get(e), sc[e], sh[e]);
            if (silu) z = silu_f(z);
            o.set(e, z);
        }
        o.store(yp);
    }
}

// forward apply: y = silu?((x - mean) * rstd * gamma + beta)
// backward apply: dx = rstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)) [+ residual]
// Same thread layout as the reduction: a thread owns ONE 8-channel chunk (its affine parameters and group statistics live
// in registers) and walks the block's pixel range, so the inner loop is load - 8 fma - store with 32-bit index math only.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void k_gn_apply(int HW, int C, int G, int pix_per_block, const T* __restrict__ x,
                                                  const T* __restrict__ dy, const float* __restrict__ stats,
                                                  const float* __restrict__ bsums, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int silu, float eps,
                                                  T* __restrict__ out, const float* __restrict__ partials, int pchunks,
                                                  float* __restrict__ sums_out, const T* __restrict__ residual, int deep) {
    __shared__ float gstat[64 * 4];
    __shared__ float tot[128];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int C8 = C / 8, cg = C / G;
    const float inv_n = 1.f / ((float)HW * cg);
    if (partials) {
        // Few chunks (small tensors): the fixed-order sum of the reduction pass's partials [b][2G][chunk] is redone by every
        // workgroup right here -- 2G x chunks floats out of L2 -- instead of a separate finalize launch between two
        // 10-us kernels.  4 threads per output, strided over the chunks, combined in a fixed order.
        const int o = tid >> 2, part = tid & 3;
        float a = 0.f;
        if (o < 2 * G) {
            const float* src = partials + ((size_t)b * 2 * G + o) * pchunks;
            for (int c = part; c < pchunks; c += 4) a += src[c];
        }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
        if (part == 0 && o < 2 * G) {
            tot[o] = a;
            if (blockIdx.x == 0) sums_out[(size_t)b * 2 * G + o] = a;      // the saved statistics (forward) / scratch (backward)
        }
        __syncthreads();
    }
    if (tid < G) {
        const float* st = BWD ? stats : (partials ? tot : stats + (size_t)b * G * 2);
        const float* bs = BWD ? (partials ? tot : bsums + (size_t)b * G * 2) : nullptr;
        const float* fst = BWD ? stats + (size_t)b * G * 2 : st;
        float m = fst[2 * tid] * inv_n;
        float var = fst[2 * tid + 1] * inv_n - m * m;
        gstat[4 * tid] = m; gstat[4 * tid + 1] = rsqrtf(fmaxf(var, 0.f) + eps);
        if (BWD) { gstat[4 * tid + 2] = bs[2 * tid] * inv_n; gstat[4 * tid + 3] = bs[2 * tid + 1] * inv_n; }
    }
    __syncthreads();
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    for (int cb = 0; cb < C8; cb += 256) {
        const int tpx = min(256, C8 - cb), rows = 256 / tpx;
        const int cc = cb + tid % tpx, prow = tid / tpx;
        if (prow >= rows) continue;
        float ga[8], be[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            int ch = cc * 8 + e, g = ch / cg;
            ga[e] = gamma[ch]; be[e] = beta[ch]; mu[e] = gstat[4 * g]; rs[e] = gstat[4 * g + 1];
            m1[e] = BWD ? gstat[4 * g + 2] : 0.f; m2[e] = BWD ? gstat[4 * g + 3] : 0.f;
        }
        const T* xp = x + ((size_t)b * HW + p0 + prow) * C + (size_t)cc * 8;
        const T* dp = BWD ? dy + ((size_t)b * HW + p0 + prow) * C + (size_t)cc * 8 : nullptr;
        const T* rp = (BWD && residual) ? residual + ((size_t)b * HW + p0 + prow) * C + (size_t)cc * 8 : nullptr;
        T* op = out + ((size_t)b * HW + p0 + prow) * C + (size_t)cc * 8;
        const size_t step = (size_t)rows * C;
        int p = p0 + prow;
        if (!BWD) {
            // long pixel ranges: four independent loads in flight per thread (see k_gn_reduce)
            for (; deep && p + 3 * rows < p1; p += 4 * rows) {
                V8<T> xv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xv[u] = V8<T>::load(xp + u * step);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    V8<T> o;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float z = (xv[u].get(e) - mu[e]) * rs[e] * ga[e] + be[e];
                        o.set(e, silu ? silu_f(z) : z);
                    }
                    o.store(op + u * step);
                }
                xp += 4 * step; op += 4 * step;
            }
        } else {
            for (; deep && p + rows < p1; p += 2 * rows) {
                V8<T> xv[2], dv[2], rv[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    xv[u] = V8<T>::load(xp + u * step); dv[u] = V8<T>::load(dp + u * step);
                    if (rp) rv[u] = V8<T>::load(rp + u * step);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    V8<T> o;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float xh = (xv[u].get(e) - mu[e]) * rs[e];
                        float gq = dv[u].get(e);
                        if (silu) gq *= silu_grad(xh * ga[e] + be[e]);
                        gq *= ga[e];
                        o.set(e, rs[e] * (gq - m1[e] - xh * m2[e]));
                    }
                    if (rp) {
#pragma unroll
                        for (int e = 0; e < 8; e++) o.set(e, o.get(e) + rv[u].get(e));
                    }
                    o.store(op + u * step);
                }
                xp += 2 * step; op += 2 * step; dp += 2 * step;
                if (rp) rp += 2 * step;
            }
        }
        for (; p < p1; p += rows) {
            V8<T> xv = V8<T>::load(xp);
            V8<T> o;
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float z = (xv.get(e) - mu[e]) * rs[e] * ga[e] + be[e];
                    o.set(e, silu ? silu_f(z) : z);
                }
            } else {
                V8<T> dv = V8<T>::load(dp);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float xh = (xv.get(e) - mu[e]) * rs[e];
                    float gq = dv.get(e);
                    if (silu) gq *= silu_grad(xh * ga[e] + be[e]);
                    gq *= ga[e];
                    o.set(e, rs[e] * (gq - m1[e] - xh * m2[e]));
                }
                if (rp) {                       // skip-connection gradient added here instead of a separate add pass
                    V8<T> rv = V8<T>::load(rp);
#pragma unroll
                    for (int e = 0; e < 8; e++) o.set(e, o.get(e) + rv.get(e));
                    rp += step;
                }
                dp += step;
            }
            o.store(op);
            xp += step; op += step;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C % 8 == 0, C <= 2048): one wave per row, values kept in registers between the passes
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_layernorm(int M, int C, const T* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, T* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int C8 = C / 8;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        int cc = lane + it * 64;
        if (cc < C8) {
            V8<T> xv = V8<T>::load(x + (size_t)row * C + (size_t)cc * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) { v[it][e] = xv.get(e); s += v[it][e]; }
        }
    }
    s = dwg_wave_sum_all(s);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        int cc = lane + it * 64;
        if (cc < C8) {
#pragma unroll
            for (int e = 0; e < 8; e++) { float d = v[it][e] - mean; q += d * d; }
        }
    }
    q = dwg_wave_sum_all(q);
    const float rstd = rsqrtf(q / C + eps);
#pragma unroll
    for (int it = 0; it < 4; it++) {
        int cc = lane + it * 64;
        if (cc < C8) {
            V8<T> o;
#pragma unroll
            for (int e = 0; e < 8; e++) { int ch = cc * 8 + e; o.set(e, (v[it][e] - mean) * rstd * gamma[ch] + beta[ch]); }
            o.store(y + (size_t)row * C + (size_t)cc * 8);
        }
    }
}

// GEGLU: out[m][f] = x[m][f] * gelu(x[m][F + f])   (diffusers GEGLU: hidden, gate = proj(x).chunk(2); hidden * gelu(gate))
template <typename T>
__global__ __launch_bounds__(256) void k_geglu(long long M, int F, const T* __restrict__ x, T* __restrict__ out) {
    const int F8 = F / 8;
    const long long n = M * F8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        long long m = i / F8; int fc = (int)(i % F8);
        V8<T> a = V8<T>::load(x + m * 2 * F + (size_t)fc * 8);
        V8<T> g = V8<T>::load(x + m * 2 * F + F + (size_t)fc * 8);
        V8<T> o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float gv = g.get(e);
            o.set(e, a.get(e) * 0.5f * gv * (1.f + dwg_erf_fast(gv * 0.70710678118654752f)));
        }
        o.store(out + m * F + (size_t)fc * 8);
    }
}

// Row softmax of fp32 scores (one wave per row, n <= 8192): P = softmax(scale * S) stored as bf16 (row stride ldp)
template <typename T>
__global__ __launch_bounds__(256) void k_softmax_rows(int rows, int n, float scale, const float* __restrict__ S, long long lds_,
                                                      T* __restrict__ P, long long ldp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* s = S + (size_t)row * lds_;
    float mx = -3.0e38f;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, s[j] * scale);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += __expf(s[j] * scale - mx);
    sum = dwg_wave_sum_all(sum);
    const float inv = 1.f / sum;
    T* p = P + (size_t)row * ldp;
    for (int j = lane; j < n; j += 64) st1<T>(p, j, __expf(s[j] * scale - mx) * inv);
}

// dS = scale * P * (dP - sum_j dP_j P_j)   (P bf16, dP fp32) -> bf16
template <typename T>
__global__ __launch_bounds__(256) void k_softmax_rows_bwd(int rows, int n, float scale, const T* __restrict__ P, long long ldp,
                                                          const float* __restrict__ dP, long long lddp, T* __restrict__ dS,
                                                          long long ldds) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* p = P + (size_t)row * ldp; const float* dp = dP + (size_t)row * lddp;
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += ld1<T>(p, j) * dp[j];
    dot = dwg_wave_sum_all(dot);
    T* ds = dS + (size_t)row * ldds;
    for (int j = lane; j < n; j += 64) st1<T>(ds, j, scale * ld1<T>(p, j) * (dp[j] - dot));
}

// DWG_GN_SHALLOW=1: the round-2 loops (one load in flight per thread) -- A/B switch for the long-range paths
static int gn_deep() { static const int v = getenv("DWG_GN_SHALLOW") ? 0 : 1; return v; }

static int gn_geometry(int HW, int C, int G, int* pix_per_block, int* chunks, size_t* lds) {
    if (C % 8 || G <= 0 || G > 64 || C % G) return DWG_E_ARG;
    int ppb = HW / 256; if (ppb < 16) ppb = 16; if (ppb > 256) ppb = 256;
    *pix_per_block = ppb; *chunks = dwg_cdiv(HW, ppb);
    *lds = (size_t)256 * 8 * 2 * sizeof(float);
    return DWG_OK;
}
// reduction pass: ~64K elements per workgroup, at most GN_MAX_CHUNKS partials per image
#define GN_MAX_CHUNKS_DEEP 2048     // 512^2 x 128-channel VAE tensors: 2048 workgroups of the statistics pass (8 per CU) keep HBM busy
#define GN_MAX_CHUNKS (gn_deep() ? GN_MAX_CHUNKS_DEEP : 512)
#define GN_FOLD_MAX_CHUNKS 32      // up to this many partials per output the apply pass sums them itself (no finalize launch)
static void gn_reduce_geometry(int HW, int C, int* pix_per_block, int* chunks) {
    long long ppb = 16384 / C; if (ppb < 8) ppb = 8;
    // small images (the denoiser's <= 64x64 levels): at most DWG_GN_SMALL_CHUNKS partials per image, so that the apply pass can sum
    // them itself and the finalize launch disappears (these layers are launch-latency-bound, not bandwidth-bound)
    static const int small_chunks = getenv("DWG_GN_SMALL_CHUNKS") ? atoi(getenv("DWG_GN_SMALL_CHUNKS")) : 0;
    if (small_chunks > 0 && HW <= 4096 && ppb * small_chunks < HW) ppb = (HW + small_chunks - 1) / small_chunks;
    int ch = (int)((HW + ppb - 1) / ppb);
    if (ch > GN_MAX_CHUNKS) { ch = GN_MAX_CHUNKS; ppb = (HW + ch - 1) / ch; ch = (int)((HW + ppb - 1) / ppb); }
    *pix_per_block = (int)ppb; *chunks = ch;
}

}  // namespace

// element-type dispatch of the entry points (DWG_DTYPE_* of include/dwg_types.h)
#define DWG_DT_SWITCH(DT, ...)                                                           \
    switch (DT) {                                                                        \
        case DWG_DTYPE_BF16: { typedef __bf16 T; __VA_ARGS__; } break;                   \
        case DWG_DTYPE_F16: { typedef _Float16 T; __VA_ARGS__; } break;                  \
        case DWG_DTYPE_F32: { typedef float T; __VA_ARGS__; } break;                     \
        case DWG_DTYPE_F32X: { typedef dwg_xs T; __VA_ARGS__; } break;                   \
        default: return DWG_E_ARG;                                                       \
    }

extern "C" {

size_t dwg_groupnorm_workspace_floats(int32_t B, int32_t G) { return (size_t)(B > 0 ? B : 1) * GN_MAX_CHUNKS_DEEP * (G > 0 ? G : 1) * 2; }

int dwg_groupnorm_forward_dt(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const float* gamma, const float* beta,
                             float eps, int32_t fuse_silu, void* y, float* stats, float* workspace, dwg_stream_t stream_) {
    if (B <= 0 || HW <= 0 || !x || !gamma || !beta || !y || !stats || !workspace) return DWG_E_ARG;
    int ppb, chunks; size_t lds;
    int rc = gn_geometry(HW, C, G, &ppb, &chunks, &lds);
    if (rc) return rc;
    int rppb, rchunks;
    gn_reduce_geometry(HW, C, &rppb, &rchunks);
    hipStream_t stream = (hipStream_t)stream_;
    {
        // small images: one launch for the whole layer (k_gn_small) when a bundle of gb whole groups is a whole number of 16-byte chunks and
        // a workgroup's two walks over it stay short (<= 24 trips per thread)
        static const bool no_small = getenv("DWG_GN_NO_SMALL") != nullptr;
        const int cg = C / G;
        int gb = 0;
        for (int c = 1; c <= 8 && !gb; c *= 2) if ((c * cg) % 8 == 0 && G % c == 0) gb = c;
        static const int small_max = getenv("DWG_GN_SMALL_MAX") ? atoi(getenv("DWG_GN_SMALL_MAX")) : 6144;     // chunks per workgroup walk
        // ... and only for the small batches it was built for (B <= 4, tensor <= 8 MB: L2-resident).  A bundle is an 80-240-byte slice of every
        // pixel row, so the walks touch partial cache lines that neighbouring bundles touch again -- free out of L2, several times the traffic
        // beyond it -- and with 8x the batch the three-launch path has all the parallelism it needs: the batched 8-view step (batch 16) fell
        // from 92 to 68 views/s with this kernel on (A/B, DWG_GN_NO_SMALL), so it keeps the three-launch path
        const long long tensor_bytes = (long long)B * HW * C * (dtype == DWG_DTYPE_F32 || dtype == DWG_DTYPE_F32X ? 4 : 2);
        if (!no_small && gb && B <= 4 && HW <= 1024 && (long long)HW * (gb * cg / 8) <= small_max && gb * cg / 8 <= 64 && tensor_bytes <= (8ll << 20)) {
            const int nb = gb * cg, tp = 256 / (nb / 8);
            const size_t sl = (size_t)(2 * tp * nb + 2 * nb) * sizeof(float);
            DWG_DT_SWITCH(dtype,
                DWG_LAUNCH("gn_small", (k_gn_small<T>), dim3(G / gb, B), dim3(256), sl, stream, HW, C, G, gb, (const T*)x, gamma, beta, fuse_silu,
                           eps, (T*)y, stats))
            DWG_RETURN_IF_LAUNCH_FAILED();
            return DWG_OK;
        }
    }
    static const int fold_max = getenv("DWG_GN_FOLD") ? atoi(getenv("DWG_GN_FOLD")) : GN_FOLD_MAX_CHUNKS;
    const bool fold = rchunks <= fold_max && 2 * G * 4 <= 256;     // finalize folded into the apply pass
    DWG_DT_SWITCH(dtype,
        DWG_LAUNCH("gn_stats", (k_gn_reduce<T, false>), dim3(rchunks, B), dim3(256), lds, stream, HW, C, G, rppb, (const T*)x,
                   (const T*)nullptr, (const float*)nullptr, gamma, beta, 0, eps, workspace, gn_deep());
        if (!fold) DWG_LAUNCH("gn_finalize", k_gn_finalize, dim3((2 * G + 3) / 4, B), dim3(256), 0, stream, rchunks, G, (const float*)workspace, stats);
        DWG_LAUNCH("gn_apply", (k_gn_apply<T, false>), dim3(chunks, B), dim3(256), 0, stream, HW, C, G, ppb, (const T*)x,
                   (const T*)nullptr, (const float*)stats, (const float*)nullptr, gamma, beta, fuse_silu, eps, (T*)y,
                   fold ? (const float*)workspace : (const float*)nullptr, rchunks, stats, (const T*)nullptr, gn_deep()))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_groupnorm_forward(int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const float* gamma, const float* beta,
                          float eps, int32_t fuse_silu, void* y, float* stats, float* workspace, dwg_stream_t stream) {
    return dwg_groupnorm_forward_dt(DWG_DTYPE_BF16, B, HW, C, G, x, gamma, beta, eps, fuse_silu, y, stats, workspace, stream);
}

int dwg_groupnorm_backward_dt(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const void* dy, const float* stats,
                              const float* gamma, const float* beta, float eps, int32_t fuse_silu, void* dx, float* scratch,
                              float* workspace, const void* residual, dwg_stream_t stream_) {
    if (B <= 0 || HW <= 0 || !x || !dy || !stats || !gamma || !beta || !dx || !scratch || !workspace) return DWG_E_ARG;
    int ppb, chunks; size_t lds;
    int rc = gn_geometry(HW, C, G, &ppb, &chunks, &lds);
    if (rc) return rc;
    int rppb, rchunks;
    gn_reduce_geometry(HW, C, &rppb, &rchunks);
    hipStream_t stream = (hipStream_t)stream_;
    static const int fold_max = getenv("DWG_GN_FOLD") ? atoi(getenv("DWG_GN_FOLD")) : GN_FOLD_MAX_CHUNKS;
    const bool fold = rchunks <= fold_max && 2 * G * 4 <= 256;
    DWG_DT_SWITCH(dtype,
        DWG_LAUNCH("gn_bwd_stats", (k_gn_reduce<T, true>), dim3(rchunks, B), dim3(256), lds, stream, HW, C, G, rppb, (const T*)x,
                   (const T*)dy, stats, gamma, beta, fuse_silu, eps, workspace, gn_deep());
        if (!fold) DWG_LAUNCH("gn_finalize", k_gn_finalize, dim3((2 * G + 3) / 4, B), dim3(256), 0, stream, rchunks, G, (const float*)workspace, scratch);
        DWG_LAUNCH("gn_bwd_apply", (k_gn_apply<T, true>), dim3(chunks, B), dim3(256), 0, stream, HW, C, G, ppb, (const T*)x,
                   (const T*)dy, stats, (const float*)scratch, gamma, beta, fuse_silu, eps, (T*)dx,
                   fold ? (const float*)workspace : (const float*)nullptr, rchunks, scratch, (const T*)residual, gn_deep()))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_groupnorm_backward(int32_t B, int32_t HW, int32_t C, int32_t G, const void* x, const void* dy, const float* stats,
                           const float* gamma, const float* beta, float eps, int32_t fuse_silu, void* dx, float* scratch,
                           float* workspace, const void* residual, dwg_stream_t stream) {
    return dwg_groupnorm_backward_dt(DWG_DTYPE_BF16, B, HW, C, G, x, dy, stats, gamma, beta, eps, fuse_silu, dx, scratch, workspace, residual,
                                     stream);
}

int dwg_layernorm_forward_dt(int32_t dtype, int32_t M, int32_t C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                             dwg_stream_t stream) {
    if (M < 0 || C <= 0 || C % 8 || C > 2048 || !x || !gamma || !beta || !y) return DWG_E_ARG;
    if (M == 0) return DWG_OK;
    DWG_DT_SWITCH(dtype, DWG_LAUNCH("layernorm", k_layernorm<T>, dim3(dwg_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, M, C, (const T*)x,
                                    gamma, beta, eps, (T*)y))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_layernorm_forward(int32_t M, int32_t C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                          dwg_stream_t stream) {
    return dwg_layernorm_forward_dt(DWG_DTYPE_BF16, M, C, x, gamma, beta, eps, y, stream);
}

int dwg_geglu_forward_dt(int32_t dtype, int64_t M, int32_t F, const void* x, void* out, dwg_stream_t stream) {
    if (M < 0 || F <= 0 || F % 8 || !x || !out) return DWG_E_ARG;
    if (M == 0) return DWG_OK;
    long long n = M * (F / 8);
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    DWG_DT_SWITCH(dtype, DWG_LAUNCH("geglu", k_geglu<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)M, F, (const T*)x, (T*)out))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_geglu_forward(int64_t M, int32_t F, const void* x, void* out, dwg_stream_t stream) {
    return dwg_geglu_forward_dt(DWG_DTYPE_BF16, M, F, x, out, stream);
}

int dwg_softmax_rows_forward_dt(int32_t dtype, int32_t rows, int32_t n, float scale, const float* S, int64_t lds, void* P, int64_t ldp,
                                dwg_stream_t stream) {
    if (rows < 0 || n <= 0 || !S || !P) return DWG_E_ARG;
    if (rows == 0) return DWG_OK;
    DWG_DT_SWITCH(dtype, DWG_LAUNCH("softmax_rows", k_softmax_rows<T>, dim3(dwg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, rows, n, scale,
                                    S, (long long)lds, (T*)P, (long long)ldp))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_softmax_rows_forward(int32_t rows, int32_t n, float scale, const float* S, int64_t lds, void* P, int64_t ldp,
                             dwg_stream_t stream) {
    return dwg_softmax_rows_forward_dt(DWG_DTYPE_BF16, rows, n, scale, S, lds, P, ldp, stream);
}

int dwg_softmax_rows_backward_dt(int32_t dtype, int32_t rows, int32_t n, float scale, const void* P, int64_t ldp, const float* dP, int64_t lddp,
                                 void* dS, int64_t ldds, dwg_stream_t stream) {
    if (rows < 0 || n <= 0 || !P || !dP || !dS) return DWG_E_ARG;
    if (rows == 0) return DWG_OK;
    DWG_DT_SWITCH(dtype, DWG_LAUNCH("softmax_rows_bwd", k_softmax_rows_bwd<T>, dim3(dwg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, rows, n,
                                    scale, (const T*)P, (long long)ldp, dP, (long long)lddp, (T*)dS, (long long)ldds))
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_softmax_rows_backward(int32_t rows, int32_t n, float scale, const void* P, int64_t ldp, const float* dP, int64_t lddp,
                              void* dS, int64_t ldds, dwg_stream_t stream) {
    return dwg_softmax_rows_backward_dt(DWG_DTYPE_BF16, rows, n, scale, P, ldp, dP, lddp, dS, ldds, stream);
}

}  // extern "C"
