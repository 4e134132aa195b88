// elementwise.hip -- small bandwidth-bound helpers around the GEMM primitive (activation backward + bias gradient,
// fused multi-tensor Adam).  Each is a grid-stride kernel with 16-byte accesses where the layout allows.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_elementwise.h"
#include "dwg_xfmt.h"

namespace {

__device__ __forceinline__ float act_grad_from_output(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;            // relu
        case 2: return y > 0.f ? 1.f : 0.01f;          // leaky relu (sign of output == sign of input)
        case 5: return y * (1.f - y);                  // sigmoid
        default: return 1.f;
    }
}

// dz = dy * act'(y); colsum[n] += sum_m dz[m][n].  N <= 256.  One block handles ROWS_PER_BLOCK rows.
__global__ __launch_bounds__(256) void k_act_bwd_colsum(int M, int N, int act, const float* __restrict__ dy,
                                                        const float* __restrict__ y, float* __restrict__ dz,
                                                        float* __restrict__ colsum, int rows_per_block) {
    __shared__ float part[256 * 4];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    if ((N & 3) == 0 && N <= 256) {
        // four consecutive columns per thread: 16-byte loads / stores (the scalar version streamed 77 MB at 3 TB/s)
        const int tpr = N >> 2, rows_par = 256 / tpr;
        const int cq = tid % tpr, rsub = tid / tpr;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (rsub < rows_par) {
            for (int r = r0 + rsub; r < r1; r += rows_par) {
                const size_t i = ((size_t)r * N >> 2) + cq;
                float4 g = reinterpret_cast<const float4*>(dy)[i];
                if (y) {
                    const float4 yy = reinterpret_cast<const float4*>(y)[i];
                    g.x *= act_grad_from_output(yy.x, act); g.y *= act_grad_from_output(yy.y, act);
                    g.z *= act_grad_from_output(yy.z, act); g.w *= act_grad_from_output(yy.w, act);
                }
                if (dz) reinterpret_cast<float4*>(dz)[i] = g;
                s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) part[tid * 4 + e] = (rsub < rows_par) ? s[e] : 0.f;
        __syncthreads();
        if (colsum && tid < N) {
            const int q = tid >> 2, e = tid & 3;
            float t = 0.f;
            for (int k = 0; k < rows_par; k++) t += part[(k * tpr + q) * 4 + e];
            atomicAdd(&colsum[tid], t);
        }
        return;
    }
    const int tpr = N;                       // threads per row (one thread per column)
    const int rows_par = 256 / tpr;          // rows processed in parallel
    const int col = tid % tpr, rsub = tid / tpr;
    float s = 0.f;
    if (rsub < rows_par) {
        for (int r = r0 + rsub; r < r1; r += rows_par) {
            size_t i = (size_t)r * N + col;
            float g = dy[i] * (y ? act_grad_from_output(y[i], act) : 1.f);
            if (dz) dz[i] = g;
            s += g;
        }
    }
    part[tid] = (rsub < rows_par) ? s : 0.f;
    __syncthreads();
    if (colsum && tid < tpr) {
        float t = 0.f;
        for (int k = 0; k < rows_par; k++) t += part[k * tpr + tid];
        atomicAdd(&colsum[tid], t);
    }
}

// Adam (torch.optim.Adam semantics, amsgrad=False, weight_decay=0, maximize=False):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// `hyper` != nullptr: the per-step scalars come from DEVICE memory -- hyper[0] = lr / (1 - b1^t), hyper[1] = sqrt(1 - b2^t), hyper[2] =
// grad_scale -- so that the launch can sit inside a captured graph whose replays see the learning-rate schedule and the step count move.
__global__ __launch_bounds__(256) void k_adam(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2,
                                              float eps, float bc1, float bc2_sqrt, float grad_scale, const float* __restrict__ hyper) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    float step = lr / bc1;
    if (hyper) { step = hyper[0]; bc2_sqrt = hyper[1]; grad_scale = hyper[2]; }
    for (size_t k = i; k < n4; k += stride) {
        float4 pp = p4[k], gg = g4[k], mm = m4[k], vv = v4[k];
#define UPD(c) { float gr = gg.c * grad_scale; mm.c = b1 * mm.c + (1.f - b1) * gr; vv.c = b2 * vv.c + (1.f - b2) * gr * gr; \
                 pp.c -= step * mm.c / (sqrtf(vv.c) / bc2_sqrt + eps); }
        UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
        p4[k] = pp; m4[k] = mm; v4[k] = vv;
    }
    for (size_t k = n4 * 4 + i; k < n; k += stride) {
        float gr = g[k] * grad_scale;
        float mm = b1 * m[k] + (1.f - b1) * gr, vv = b2 * v[k] + (1.f - b2) * gr * gr;
        m[k] = mm; v[k] = vv;
        p[k] -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

struct AdamGroups { dwg_adam_group g[DWG_ADAM_MAX_GROUPS]; };

// k_adam for several parameter groups: blockIdx.y = group (groups shorter than the grid's x extent leave their surplus workgroups idle)
__global__ __launch_bounds__(256) void k_adam_groups(AdamGroups G, const float* __restrict__ hyper) {
    const dwg_adam_group a = G.g[blockIdx.y];
    const size_t n = (size_t)a.n, n4 = n / 4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4 && n4 * 4 + i >= n) return;
    const size_t stride = (size_t)gridDim.x * 256;
    float* __restrict__ p = a.param; const float* __restrict__ g = a.grad; float* __restrict__ m = a.exp_avg; float* __restrict__ v = a.exp_avg_sq;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    const float* h = hyper + 4 * (size_t)a.hyper_row;
    const float step = h[0], bc2_sqrt = h[1], grad_scale = h[2], b1 = a.beta1, b2 = a.beta2, eps = a.eps;
    for (size_t k = i; k < n4; k += stride) {
        float4 pp = p4[k], gg = g4[k], mm = m4[k], vv = v4[k];
#define UPD(c) { float gr = gg.c * grad_scale; mm.c = b1 * mm.c + (1.f - b1) * gr; vv.c = b2 * vv.c + (1.f - b2) * gr * gr; \
                 pp.c -= step * mm.c / (sqrtf(vv.c) / bc2_sqrt + eps); }
        UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
        p4[k] = pp; m4[k] = mm; v4[k] = vv;
    }
    for (size_t k = n4 * 4 + i; k < n; k += stride) {
        float gr = g[k] * grad_scale;
        float mm = b1 * m[k] + (1.f - b1) * gr, vv = b2 * v[k] + (1.f - b2) * gr * gr;
        m[k] = mm; v[k] = vv;
        p[k] -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

// dW[n][k] = sum_m dz[m][n] * x[m][k]  for the per-Gaussian MLPs (N, K <= 64, M ~ 1e5): every workgroup reduces a slab of
// rows into a 64x64 tile (exact-f32 MFMA, one 32x32 quadrant per wave) from LDS-staged 64-row chunks and writes ONE partial tile; a second
// pass sums the partial tiles in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void k_mlp_wgrad_partial(int M, int N, int K, const float* __restrict__ dz, int lddz,
                                                           const float* __restrict__ x, int ldx, int rows_per_block,
                                                           float* __restrict__ partial /*[blocks][64][64]*/) {
    __shared__ __attribute__((aligned(16))) float sdz[64][68];
    __shared__ __attribute__((aligned(16))) float sx[64][68];
    // exact-f32 MFMA (v_mfma_f32_32x32x2_f32): the 64x64 tile is 2x2 MFMA tiles, one per wave; the contraction runs over the
    // rows of the chunk, two rows per instruction.  A = dz^T (lane: n = l & 31, row parity = l >> 5), B = x (lane: k = l & 31).
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = (wave >> 1) * 32, tk = (wave & 1) * 32;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    // chunk staging is register-prefetched one chunk ahead (thread t owns column t & 63 of rows (t >> 6) + 4 i): the global
    // loads of chunk c+1 are in flight while chunk c is multiplied
    const int sc = tid & 63, sr = tid >> 6;
    float rdz[16], rx[16];
    auto fetch = [&](int base) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = base + sr + 4 * i;
            rdz[i] = (m < r1 && sc < N) ? dz[(size_t)m * lddz + sc] : 0.f;
            rx[i] = (m < r1 && sc < K) ? x[(size_t)m * ldx + sc] : 0.f;
        }
    };
    if (r0 < r1) fetch(r0);
    for (int base = r0; base < r1; base += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) { sdz[sr + 4 * i][sc] = rdz[i]; sx[sr + 4 * i][sc] = rx[i]; }
        __syncthreads();
        if (base + 64 < r1) fetch(base + 64);
        const float* pa = &sdz[lane >> 5][tn + (lane & 31)];
        const float* pb = &sx[lane >> 5][tk + (lane & 31)];
#pragma unroll 8
        for (int st = 0; st < 32; st++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[st * 2 * 68], pb[st * 2 * 68], acc, 0, 0, 0);
    }
    float* dst = partial + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int n = tn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[n * 64 + tk + (lane & 31)] = acc[r];
    }
}

__global__ __launch_bounds__(256) void k_mlp_wgrad_final(int blocks, int N, int K, const float* __restrict__ partial,
                                                         float* __restrict__ dw, int lddw) {
    // 16 outputs per workgroup x 16 strands over the partial tiles (a strand reads ~25 tiles 16 KiB apart: with 4 strands the pass was
    // a 100-deep chain of dependent-latency loads, 25 us for 1.6 MB); fixed combination order
    __shared__ float part[16][17];
    const int o = threadIdx.x & 15, strand = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + o;
    float s = 0.f;
    for (int b = strand; b < blocks; b += 16) s += partial[(size_t)b * 4096 + e];
    part[strand][o] = s;
    __syncthreads();
    if (strand == 0) {
        const int n = e >> 6, k = e & 63;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) t += part[q][o];
        if (n < N && k < K) dw[(size_t)n * lddw + k] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole per-Gaussian MLP in ONE launch: y = L_n(... act(L_1(x)) ...), widths <= 64 (nerf_model.py:12-33: 32-64-64-4, ReLU;
// deform_model.py:102-143: 32(+pose bias)-64-64-64-64-10, leaky ReLU).  Layer by layer these were 8 GEMM launches that each wrote a
// [M, 64] fp32 activation to HBM and read it back (mlp_fwd: 0.35 ms of the 2 ms c5 frame).  Here a wave keeps ITS 32 rows of
// activations in LDS across all layers (overwritten in place once a layer's MFMAs are done), every layer's weights sit in LDS for the
// whole launch, and the contraction is exact-f32 MFMA (v_mfma_f32_32x32x2_f32, transposed: a lane owns one row and groups of four
// consecutive output columns).  Hidden activations are written out only when the caller keeps them for the backward.
//
// Round 5.  At 300 k rows the static network took 93 us and the deformation network 190 us where their MFMAs need 31 and 62.  Timing
// variants (MFMAs off / epilogue off) and the SQ counters put it on the code AROUND the MFMAs: the epilogue chose the activation and
// tested the keep-hidden pointer per VALUE (six scalar branches and a dependent 4-byte LDS read of the bias for each of a lane's 32
// values per layer), and the rows were staged with an integer division per element.  Now the hidden layers' activation and the
// keep-hidden flag are compile-time (ACT_H, KEEP: the avatar's networks use one activation for all hidden layers), the accumulators
// start at the bias (eight 16-byte LDS reads issued before the MFMAs instead of 32 dependent ones after), a layer's output goes to LDS
// as 8-byte pairs, the rows come in as 16-byte pieces split by shifts with the NEXT group's pieces in flight while this group
// computes, and the per-layer scalars are read from LDS instead of the argument struct: 70 and 144 us.  (Tried and dropped: keeping
// the activations in registers -- the weights' columns permuted to the order the previous layer's accumulators hold them, no LDS
// tile, twelve waves per CU -- was SLOWER, 88 / 155 us at its best wave count, whether the rows were loaded per lane in that layout
// or coalesced and turned through LDS; the output of a 10-wide head turned through LDS into consecutive dwords: no change.)
// ---------------------------------------------------------------------------------------------------------------------
#define MLPC_MAXL 6
#define MLPC_LD 68          // row stride in floats: multiple of 4 (16-byte fragment reads)
#define MLPC_WAVES 8        // waves per workgroup sharing one LDS copy of the weights: two per SIMD hide each other's LDS / epilogue latencies
struct MlpChainP {
    const float* x; int M, Kin, ldx, nlayers;
    const float* extra; int n_extra;        // vector folded into the first layer's bias through W_0's trailing columns (may be null / 0)
    int waves;                              // waves per workgroup of this launch (<= MLPC_WAVES)
    int kshift;                             // log2(Kin) when Kin is a power of two and x rows are 16-byte aligned (vector staging), else -1
    const float* W[MLPC_MAXL]; const float* b[MLPC_MAXL]; float* hidden[MLPC_MAXL];
    int ldw[MLPC_MAXL], N[MLPC_MAXL], K[MLPC_MAXL], act[MLPC_MAXL];
    float* out; int ldo;
};

__device__ __forceinline__ float mlpc_act(float v, int act) {
    return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : 0.01f * v) : (act == 5 ? 1.f / (1.f + __expf(-v)) : v));
}
// LDS position of contraction index k of a K-wide row: the MFMA lane (row, half) consumes k = 2 ks + half for ks = 0, 1, ... -- stored
// as [half][ks] so that four consecutive steps are ONE 16-byte read (the interleaved order costs a 4-byte read per MFMA and operand)
__device__ __forceinline__ int mlpc_pos(int k, int K) { return (k & 1) * (K >> 1) + (k >> 1); }

// This lane's 16-byte pieces of the 32 rows starting at r0 (piece i = lane + 64 u: row i / (Kin / 4), columns 4 (i % (Kin / 4)) ..): coalesced,
// all in flight at once.  Rows past M read as zero.
__device__ __forceinline__ void mlpc_load_rows(const MlpChainP& p, int r0, int lane, int kq, float4 (&xr)[8]) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int i = lane + 64 * u;
        const int r = i >> (p.kshift - 2), c = i & (kq - 1);
        xr[u] = (i < 32 * kq && r0 + r < p.M) ? *reinterpret_cast<const float4*>(p.x + (size_t)(r0 + r) * p.ldx + 4 * c)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Activation of a HIDDEN layer's accumulators (which started at the bias; lane (m, half) holds row m, columns t*32 + 8q + 4 half + e), then
// the next layer's input in place in LDS (k-permuted: two 8-byte stores per four columns) and, when the backward keeps it (KEEP), the
// activation to HBM.  ACT is the layer's activation when the host found one activation for all hidden layers (1 ReLU, 2 leaky ReLU),
// -1: chosen per value from `act` (any other chain).
template <int ACT, bool KEEP>
__device__ __forceinline__ void mlpc_epilogue_hidden(const __attribute__((ext_vector_type(16))) float& acc0, const __attribute__((ext_vector_type(16))) float& acc1,
                                                     float* sX, int m, int half, int N, int Np, int act, float* hidden, int grow, int M) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (t * 32 >= Np) break;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = t * 32 + 8 * q + 4 * half;
            if (n0 >= N) continue;                          // padding columns of a width that is not a multiple of 32 (widths are multiples of 8: n0 < N covers n0 + 3)
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = mlpc_act(t == 0 ? acc0[4 * q + e] : acc1[4 * q + e], ACT < 0 ? act : ACT);
            float* row = sX + m * MLPC_LD + (n0 >> 1);
            *reinterpret_cast<float2*>(row) = make_float2(v[0], v[2]);
            *reinterpret_cast<float2*>(row + (N >> 1)) = make_float2(v[1], v[3]);
            if (KEEP) { if (grow < M) *reinterpret_cast<float4*>(hidden + (size_t)grow * N + n0) = make_float4(v[0], v[1], v[2], v[3]); }
        }
    }
}

// The LAST layer: activation (any of the four, per value: a 4- or 10-wide head is one or two blocks of four columns) -> `out`.
__device__ __forceinline__ void mlpc_epilogue_last(const __attribute__((ext_vector_type(16))) float& acc0, const __attribute__((ext_vector_type(16))) float& acc1,
                                                   int half, int N, int Np, int act, float* out, int ldo, bool out_vec, int grow, int M) {
    if (grow >= M) return;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (t * 32 >= Np) break;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = t * 32 + 8 * q + 4 * half;
            if (n0 >= N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = mlpc_act(t == 0 ? acc0[4 * q + e] : acc1[4 * q + e], act);
            if (out_vec && n0 + 3 < N) *reinterpret_cast<float4*>(out + (size_t)grow * ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; e++) if (n0 + e < N) out[(size_t)grow * ldo + n0 + e] = v[e];
            }
        }
    }
}

// PERSISTENT waves: a workgroup stages ALL layers' weights in LDS once and then every wave walks its own 32-row groups to the end of
// the input with no workgroup barrier at all (its activations are wave-private, the weights read-only) -- restaging the weights per
// 128-row tile with two barriers per layer left the MFMA pipe 30 % busy.
template <int ACT_H, bool KEEP>
__global__ __launch_bounds__(64 * MLPC_WAVES) void k_mlp_chain(MlpChainP p, int wfloats) {
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sWall = smem;                                    // per layer [Np][MLPC_LD], k permuted
    float* sBall = smem + wfloats;                          // [MLPC_MAXL][64]
    float* sXall = sBall + MLPC_MAXL * 64;                  // [waves][32][MLPC_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = 64 * p.waves;
    // Per-layer scalars, read from LDS inside the row-group loop: indexing the by-value argument struct with the (run-time) layer number
    // compiles to scalar loads from the kernarg segment inside the loop; they are read ONCE here, with constant indices, by one thread.
    __shared__ int sMeta[MLPC_MAXL][8];
    if (tid == 0) {
#pragma unroll
        for (int l = 0; l < MLPC_MAXL; l++) {
            const unsigned long long h = (unsigned long long)p.hidden[l];
            sMeta[l][0] = p.N[l]; sMeta[l][1] = p.K[l]; sMeta[l][2] = p.act[l]; sMeta[l][3] = (int)(unsigned)h; sMeta[l][4] = (int)(unsigned)(h >> 32);
        }
    }
    {
        int off = 0;
        for (int l = 0; l < p.nlayers; l++) {
            const int N = p.N[l], K = p.K[l], Np = (N + 31) & ~31;
            // eight loads in flight per thread (one load per trip made the staging a chain of dependent latencies: ~40 us of a 10 k-row launch)
            for (int i0 = tid; i0 < Np * K; i0 += 8 * nthr) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * nthr, n = i / K, k = i - n * K;
                    v[u] = (i < Np * K && n < N) ? p.W[l][(size_t)n * p.ldw[l] + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * nthr, n = i / K, k = i - n * K;
                    if (i < Np * K) sWall[off + n * MLPC_LD + mlpc_pos(k, K)] = v[u];
                }
            }
            if (tid < 64) sBall[l * 64 + tid] = (tid < N && p.b[l]) ? p.b[l][tid] : 0.f;
            off += Np * MLPC_LD;
        }
        if (p.extra) {
            // b_0[n] += sum_e W_0[n][Kin + e] extra[e] (deform_model.py:113-115 concatenates the expanded body pose to every row: the same
            // product for all rows, i.e. a bias) -- a launch of its own before (a [1, 63] x [63, 64] GEMM: 13 us on the frame's critical path).
            // Wave w sums the terms e = w, w + waves, ...; the partial sums are added in wave order.
            float* sP = sXall;                                  // [waves][64] scratch (the activation rows are not in use yet)
            const int n = lane, N0 = p.N[0];
            float a = 0.f;
            if (n < N0) {
                const float* wrow = p.W[0] + (size_t)n * p.ldw[0] + p.Kin;
                for (int e0 = wave; e0 < p.n_extra; e0 += 4 * p.waves) {          // four independent loads in flight
                    float w4[4], x4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int e = e0 + u * p.waves;
                        w4[u] = e < p.n_extra ? wrow[e] : 0.f; x4[u] = e < p.n_extra ? p.extra[e] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) a = fmaf(w4[u], x4[u], a);
                }
            }
            sP[wave * 64 + n] = a;
            __syncthreads();
            if (tid < 64) {
                float t = sBall[tid];
                for (int w = 0; w < p.waves; w++) t += sP[w * 64 + tid];
                sBall[tid] = tid < N0 ? t : 0.f;
            }
        }
    }
    __syncthreads();
    float* sX = sXall + wave * 32 * MLPC_LD;
    const int m = lane & 31, half = lane >> 5;
    const int M = p.M, nlayers = p.nlayers, ldo = p.ldo;
    float* const out = p.out;
    const bool out_vec = !(ldo & 3) && !((uintptr_t)out & 15);
    const int ngroups = (M + 31) >> 5;
    const int kq = p.Kin >> 2;                              // 16-byte pieces per row
    const bool vec = p.kshift >= 0;                         // Kin in {8, 16, 32, 64}, x rows 16-byte aligned (host)
    float4 xr[8];                                           // 32 rows x Kin <= 64 floats = at most 8 pieces per lane
    const int g0 = blockIdx.x * p.waves + wave, gstep = gridDim.x * p.waves;
    if (vec && g0 < ngroups) mlpc_load_rows(p, g0 * 32, lane, kq, xr);
    for (int g = g0; g < ngroups; g += gstep) {
        const int r0 = g * 32;
        __builtin_amdgcn_wave_barrier();
        if (vec) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = lane + 64 * u;
                if (i < 32 * kq) {
                    const int r = i >> (p.kshift - 2), k = (i & (kq - 1)) << 2;        // columns k .. k+3: even ones to [k/2, k/2+1], odd ones behind K/2
                    float* row = sX + r * MLPC_LD + (k >> 1);
                    *reinterpret_cast<float2*>(row) = make_float2(xr[u].x, xr[u].z);
                    *reinterpret_cast<float2*>(row + (p.Kin >> 1)) = make_float2(xr[u].y, xr[u].w);
                }
            }
            if (g + gstep < ngroups) mlpc_load_rows(p, (g + gstep) * 32, lane, kq, xr);   // in flight while the layers below run
        } else {
            for (int i0 = lane; i0 < 32 * p.Kin; i0 += 8 * 64) {        // generic widths: eight 4-byte loads in flight per lane
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + 64 * u, r = i / p.Kin, k = i - r * p.Kin;
                    v[u] = (i < 32 * p.Kin && r0 + r < M) ? p.x[(size_t)(r0 + r) * p.ldx + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + 64 * u, r = i / p.Kin, k = i - r * p.Kin;
                    if (i < 32 * p.Kin) sX[r * MLPC_LD + mlpc_pos(k, p.Kin)] = v[u];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int grow = r0 + m;
        int off = 0;
        for (int l = 0; l < nlayers; l++) {
            const int N = __builtin_amdgcn_readfirstlane(sMeta[l][0]), K = __builtin_amdgcn_readfirstlane(sMeta[l][1]), Np = (N + 31) & ~31;
            const float* sW = sWall + off;
            const float* sB = sBall + l * 64;
            off += Np * MLPC_LD;
            // the accumulators START at the bias (lane (m, half) owns columns t*32 + 8q + 4 half + e): the epilogue then has no LDS read to wait for
            f32x16_t acc0, acc1;
            {
                const float4* pb = reinterpret_cast<const float4*>(sB + 4 * half);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 b0 = pb[2 * q], b1 = pb[8 + 2 * q];
                    acc0[4 * q] = b0.x; acc0[4 * q + 1] = b0.y; acc0[4 * q + 2] = b0.z; acc0[4 * q + 3] = b0.w;
                    acc1[4 * q] = b1.x; acc1[4 * q + 1] = b1.y; acc1[4 * q + 2] = b1.z; acc1[4 * q + 3] = b1.w;
                }
            }
            const float4* px = reinterpret_cast<const float4*>(sX + m * MLPC_LD + half * (K >> 1));
            const float4* pw0 = reinterpret_cast<const float4*>(sW + m * MLPC_LD + half * (K >> 1));
            const float4* pw1 = reinterpret_cast<const float4*>(sW + (32 + m) * MLPC_LD + half * (K >> 1));
            const int nq = K >> 3;                          // 16-byte groups of four MFMA steps (K % 8 == 0)
            if (Np == 64) {                                 // two independent accumulator chains interleaved
#pragma unroll 2
                for (int q = 0; q < nq; q++) {
                    const float4 xv = px[q], w0 = pw0[q], w1 = pw1[q];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, xv.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, xv.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, xv.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, xv.w, acc1, 0, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int q = 0; q < nq; q++) {
                    const float4 xv = px[q], w0 = pw0[q];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc0, 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();                // every lane's reads of this layer's input precede the in-place writes
            const int act = __builtin_amdgcn_readfirstlane(sMeta[l][2]);
            if (l + 1 < nlayers) {
                float* hid = nullptr;
                if (KEEP) hid = reinterpret_cast<float*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(sMeta[l][3]) |
                                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(sMeta[l][4]) << 32));
                mlpc_epilogue_hidden<ACT_H, KEEP>(acc0, acc1, sX, m, half, N, Np, act, hid, grow, M);
            } else {
                mlpc_epilogue_last(acc0, acc1, half, N, Np, act, out, ldo, out_vec, grow, M);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the whole per-Gaussian MLP in ONE launch (+ one reduce): the mirror of k_mlp_chain.  Layer by layer the backward was four
// launches per layer (activation backward + bias gradient, weight-gradient partials, their reduce, the input-gradient GEMM: 32 launches and
// ~0.33 ms for the two networks of a 50 k-Gaussian step, each [M, 64] gradient written to HBM and read back twice).  Here a workgroup
// walks 64-row chunks: the chunk's gradient lives in LDS across all layers (dz_l -> dx -> dz_{l-1} in place), every layer's weights are
// staged in LDS once per workgroup, the saved activations are read ONCE (the output of layer l-1 is both the input of layer l's weight
// gradient and the argument of layer l-1's activation derivative) one layer ahead of their use, and the weight gradients of ALL layers
// stay in registers across the chunks (one 32 x 32 quadrant per wave and layer, exact-f32 MFMA) -- written out once per workgroup and
// summed by k_mlp_chain_bwd_final in a fixed order (deterministic, no atomics).
// LDS images: sG [64][68] = dz of the current layer with the columns of every group of eight permuted (mlpb_pos) so that BOTH uses are
// conflict-free: the weight-gradient MFMA reads a ROW (lanes = columns), the input-gradient MFMA reads four contraction steps of one
// parity as ONE 16-byte piece per lane (lanes = rows, stride 68 words); sX [64][68] = the layer's input rows, natural order.
// ---------------------------------------------------------------------------------------------------------------------
#define MLPB_LD 68
#define MLPB_PART (4096 + 256)      // floats per (workgroup, layer): the 64 x 64 weight-gradient tile + 4 row-quarter bias partials of 64
struct MlpChainBwdP {
    const float* x; int M, Kin, ldx, nlayers;
    const float* W[MLPC_MAXL]; const float* hidden[MLPC_MAXL];      // hidden[l]: the saved OUTPUT of layer l (l < nlayers - 1)
    int ldw[MLPC_MAXL], N[MLPC_MAXL], K[MLPC_MAXL], act[MLPC_MAXL];
    const float* out; int ldo;          // saved output of the last layer: read only when its activation is not the identity
    const float* dy; int lddy;
    float* dx; int lddx;                // may be null
    float* partial;                     // [gridDim.x][nlayers][MLPB_PART]
};
__device__ __forceinline__ int mlpb_pos(int n) { return (n & ~7) | ((n & 1) << 2) | ((n >> 1) & 3); }

// 64 rows x 64 columns of a row-major [M, width] tensor -> 16 registers per thread (element idx = tid + 256 i: row idx >> 6, column idx & 63;
// a wave reads one 256-byte row segment per load), zeros outside the tensor
__device__ __forceinline__ void mlpb_fetch(float (&v)[16], const float* __restrict__ src, int ld, int width, int r0, int M, int tid) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int idx = tid + 256 * i, r = r0 + (idx >> 6), c = idx & 63;
        v[i] = (r < M && c < width) ? src[(size_t)r * ld + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_mlp_chain_bwd(MlpChainBwdP p) {
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sWall = smem;                                    // [nlayers][64][MLPB_LD]: W_l[n][k], zero padded
    float* sG = smem + p.nlayers * 64 * MLPB_LD;
    float* sX = sG + 64 * MLPB_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qr = wave >> 1, qc = wave & 1, half = lane >> 5, l31 = lane & 31;
    for (int l = 0; l < p.nlayers; l++) {
        const int N = p.N[l], K = p.K[l];
        float v[16];                                        // sixteen loads in flight per thread
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = tid + 256 * u, n = i >> 6, k = i & 63;
            v[u] = (n < N && k < K) ? p.W[l][(size_t)n * p.ldw[l] + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = tid + 256 * u;
            sWall[(l * 64 + (i >> 6)) * MLPB_LD + (i & 63)] = v[u];
        }
    }
    f32x16_t accw[MLPC_MAXL];
    float accb[MLPC_MAXL];
#pragma unroll
    for (int l = 0; l < MLPC_MAXL; l++) {
        accb[l] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) accw[l][r] = 0.f;
    }
    const int L = p.nlayers - 1;
    const int nchunks = (p.M + 63) >> 6;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int r0 = chunk * 64;
        float pf[16];
        __syncthreads();                                    // the weights are staged / the previous chunk's images are no longer read
        mlpb_fetch(pf, p.dy, p.lddy, p.N[L], r0, p.M, tid);
        if (p.act[L] != 0) {
            float yo[16];
            mlpb_fetch(yo, p.out, p.ldo, p.N[L], r0, p.M, tid);
#pragma unroll
            for (int i = 0; i < 16; i++) pf[i] *= act_grad_from_output(yo[i], p.act[L]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { const int idx = tid + 256 * i; sG[(idx >> 6) * MLPB_LD + mlpb_pos(idx & 63)] = pf[i]; }
        mlpb_fetch(pf, L > 0 ? p.hidden[L - 1] : p.x, L > 0 ? p.N[L - 1] : p.ldx, p.K[L], r0, p.M, tid);
#pragma unroll
        for (int i = 0; i < 16; i++) { const int idx = tid + 256 * i; sX[(idx >> 6) * MLPB_LD + (idx & 63)] = pf[i]; }
        __syncthreads();
#pragma unroll
        for (int l = MLPC_MAXL - 1; l >= 0; l--) {
            if (l > L) continue;
            const int N = p.N[l], K = p.K[l];
            // the input rows of layer l - 1, in flight while this layer's products run
            if (l > 0) mlpb_fetch(pf, l > 1 ? p.hidden[l - 2] : p.x, l > 1 ? p.N[l - 2] : p.ldx, p.K[l - 1], r0, p.M, tid);
            {   // bias gradient: column tid & 63 over the row quarter tid >> 6
                const float* g = sG + (wave * 16) * MLPB_LD + mlpb_pos(lane);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) s += g[r * MLPB_LD];
                accb[l] += s;
            }
            if (qr * 32 < N && qc * 32 < K) {               // dW[n][k] += sum_m dz[m][n] x[m][k], quadrant (qr, qc)
                const float* pa = sG + half * MLPB_LD + mlpb_pos(qr * 32 + l31);
                const float* pb = sX + half * MLPB_LD + qc * 32 + l31;
#pragma unroll 8
                for (int st = 0; st < 32; st++)
                    accw[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[st * 2 * MLPB_LD], pb[st * 2 * MLPB_LD], accw[l], 0, 0, 0);
            }
            f32x16_t accd;
#pragma unroll
            for (int r = 0; r < 16; r++) accd[r] = 0.f;
            if (qc * 32 < K && (l > 0 || p.dx)) {           // dx[m][k] = sum_n dz[m][n] W[n][k], rows qr, columns qc
                const float4* pA = reinterpret_cast<const float4*>(sG + (qr * 32 + l31) * MLPB_LD + 4 * half);
                const float* pB = sWall + (l * 64 + half) * MLPB_LD + qc * 32 + l31;
                const int nq = (N + 7) >> 3;
#pragma unroll 2
                for (int q = 0; q < nq; q++) {
                    const float4 a = pA[2 * q];
                    const float* b = pB + 8 * q * MLPB_LD;
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[2 * MLPB_LD], accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * MLPB_LD], accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[6 * MLPB_LD], accd, 0, 0, 0);
                }
            }
            if (l > 0) {                                    // dz_{l-1} = dx * act'_{l-1}(output of layer l - 1 = this layer's input)
                const int act = p.act[l - 1];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = qr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    accd[r] *= act_grad_from_output(sX[i * MLPB_LD + qc * 32 + l31], act);
                }
            }
            __syncthreads();                                // every read of this layer's dz / input image is done
            if (l > 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = qr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    sG[i * MLPB_LD + mlpb_pos(qc * 32 + l31)] = accd[r];
                }
#pragma unroll
                for (int i = 0; i < 16; i++) { const int idx = tid + 256 * i; sX[(idx >> 6) * MLPB_LD + (idx & 63)] = pf[i]; }
                __syncthreads();
            } else if (p.dx) {
                const int k = qc * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = r0 + qr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M && k < K) p.dx[(size_t)m * p.lddx + k] = accd[r];
                }
            }
        }
    }
    float* dst = p.partial + (size_t)blockIdx.x * p.nlayers * MLPB_PART;
#pragma unroll
    for (int l = 0; l < MLPC_MAXL; l++) {
        if (l > L) continue;
        float* d = dst + (size_t)l * MLPB_PART;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = qr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            d[n * 64 + qc * 32 + l31] = accw[l][r];
        }
        d[4096 + wave * 64 + lane] = accb[l];
    }
}

struct MlpChainBwdOut {
    float* dw[MLPC_MAXL]; float* db[MLPC_MAXL]; int lddw[MLPC_MAXL], N[MLPC_MAXL], K[MLPC_MAXL];
    int acc[MLPC_MAXL];                 // bit 0: dw[l] += (the caller's gradient slice already holds other contributions), bit 1: db[l] +=
    const float* extra; int n_extra;
};
// grid (260, nlayers): blocks 0..255 sum 16 weight-gradient entries each over the workgroups' partial tiles, blocks 256..259 sixteen bias
// gradients each (a workgroup's four row-quarter partials first) -- 16 strands per output, a strand's loads issued eight at a time (a
// rolled loop was a chain of dependent-latency loads: 30 us per launch for 20 MB), fixed combination order -- and, for the first layer, the
// columns of the vector folded into its bias (dW[:, K + e] = db * extra[e])
__global__ __launch_bounds__(256) void k_mlp_chain_bwd_final(int blocks, int nlayers, const float* __restrict__ partial, MlpChainBwdOut o) {
    __shared__ float part[16][17];
    const int l = blockIdx.y;
    const float* src = partial + (size_t)l * MLPB_PART;
    const size_t stride = (size_t)nlayers * MLPB_PART;
    const int N = o.N[l], K = o.K[l];
    const int oo = threadIdx.x & 15, strand = threadIdx.x >> 4;
    const bool bias = blockIdx.x >= 256;
    const int e = bias ? (blockIdx.x - 256) * 16 + oo : blockIdx.x * 16 + oo;          // bias: column; else entry n * 64 + k
    const float* q = src + (bias ? 4096 + e : e);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b0 = strand; b0 < blocks; b0 += 128) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int b = b0 + 16 * u;
            if (b < blocks) {
                const float* r = q + (size_t)b * stride;
                acc[u] += bias ? (r[0] + r[64]) + (r[128] + r[192]) : r[0];
            }
        }
    }
    part[strand][oo] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (strand != 0) return;
    float t = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) t += part[s_][oo];
    if (!bias) {
        const int n = e >> 6, k = e & 63;
        if (n < N && k < K) { float* d = o.dw[l] + (size_t)n * o.lddw[l] + k; *d = (o.acc[l] & 1) ? *d + t : t; }
        return;
    }
    if (e >= N) return;
    if (o.db[l]) o.db[l][e] = (o.acc[l] & 2) ? o.db[l][e] + t : t;
    if (l == 0 && o.extra)
        for (int x_ = 0; x_ < o.n_extra; x_++) {
            float* d = o.dw[0] + (size_t)e * o.lddw[0] + K + x_;
            *d = (o.acc[0] & 1) ? *d + t * o.extra[x_] : t * o.extra[x_];
        }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// out[r][0:Ca] = a[r], out[r][Ca:Ca+Cb] = b[r]   (channel concat of two NHWC bf16 tensors; Ca, Cb multiples of 8)
__global__ __launch_bounds__(256) void k_concat_channels(long long rows, int Ca, int Cb, const __bf16* __restrict__ a,
                                                         const __bf16* __restrict__ b, __bf16* __restrict__ out) {
    const int C8 = (Ca + Cb) / 8, A8 = Ca / 8;
    const long long n = rows * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        long long r = i / C8; int c = (int)(i - r * C8);
        bf16x8_t v = c < A8 ? *reinterpret_cast<const bf16x8_t*>(a + r * Ca + (long long)c * 8)
                            : *reinterpret_cast<const bf16x8_t*>(b + r * Cb + (long long)(c - A8) * 8);
        *reinterpret_cast<bf16x8_t*>(out + r * (Ca + Cb) + (long long)c * 8) = v;
    }
}

// out = a + b (bf16, n % 8 == 0)
__global__ __launch_bounds__(256) void k_add_bf16(long long n8, const __bf16* __restrict__ a, const __bf16* __restrict__ b,
                                                  __bf16* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        bf16x8_t x = reinterpret_cast<const bf16x8_t*>(a)[i], y = reinterpret_cast<const bf16x8_t*>(b)[i], o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)((float)x[e] + (float)y[e]);
        reinterpret_cast<bf16x8_t*>(out)[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_cast_f32_bf16(long long n, const float* __restrict__ src, __bf16* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (__bf16)src[i];
}

// the same two passes for any activation element type (fp16 / fp32 plans)
template <typename T>
__global__ __launch_bounds__(256) void k_add_t(long long n8, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out) {
    typedef T __attribute__((ext_vector_type(8))) V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        V x = reinterpret_cast<const V*>(a)[i], y = reinterpret_cast<const V*>(b)[i], o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (T)((float)x[e] + (float)y[e]);
        reinterpret_cast<V*>(out)[i] = o;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_cast_f32_t(long long n, const float* __restrict__ src, T* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (T)src[i];
}

// ---- the split-precision f32x format (dwg_xfmt.h): fp32 <-> hi / lo fp16 planes, 8 channels (32 bytes) per thread and trip ----
__global__ __launch_bounds__(256) void k_x_pack(long long n8, const float* __restrict__ src, dwg_xs* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        dwg_x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o.set(e, v[e]);
        o.store(dst + 8 * i);
    }
}
__global__ __launch_bounds__(256) void k_x_unpack(long long n8, const dwg_xs* __restrict__ src, float* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const dwg_x8 x = dwg_x8::load(src + 8 * i);
        reinterpret_cast<float4*>(dst)[2 * i] = make_float4(x.get(0), x.get(1), x.get(2), x.get(3));
        reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(x.get(4), x.get(5), x.get(6), x.get(7));
    }
}
// ---- the f32x VAE encoder's input / output converters in one launch each (round 6; before: 4 / ~14 / 3 element-wise torch launches on the
// step's serial chain) ------------------------------------------------------------------------------------------------------------------
// image [B,3,H,W] fp32 in [0,1] -> x [B,H,W,8] f32x: channels 0..2 = 2 v - 1 (VaeImageProcessor.normalize; 2 v is exact, one rounding as in
// `image * 2.0 - 1.0`), channels 3..7 = 0 (the first convolution's padded input group)
__global__ __launch_bounds__(256) void k_vae_image_pack(long long npix, long long hw, const float* __restrict__ img, dwg_xs* __restrict__ x) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long b = i / hw, r = i - b * hw;
        const float* src = img + b * 3 * hw + r;
        dwg_x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o.set(e, e < 3 ? src[e * hw] * 2.0f - 1.0f : 0.f);
        o.store(x + 8 * i);
    }
}
// d moments [B,8,h,w] fp32 -> dmoments [B,h,w,8] f32x, pre-scaled by the power of two that brings max |g| into [target / 2, target]
// (sd15.VAEEncoderPlan.GRAD_TARGET: the backward is linear, the scale is exact and is taken out again by k_vae_dx_unpack); inv_out =
// 2 * 2^-k (the scale back times d(2 v - 1) / dv).  target <= 0: no pre-scale (k = 0).  ONE workgroup: 32 768 values at B = 1.
__global__ __launch_bounds__(1024) void k_vae_grad_prescale_pack(int B, int hw, const float* __restrict__ g, float target, dwg_xs* __restrict__ dst,
                                                                 float* __restrict__ inv_out) {
    __shared__ float s_max[16];
    __shared__ float s_scale;
    const long long n = (long long)B * 8 * hw;
    float m = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float amax = 0.f;
        for (int w = 0; w < 16; w++) amax = fmaxf(amax, s_max[w]);
        float k = 0.f;
        if (target > 0.f && amax > 0.f) k = fminf(fmaxf(floorf(log2f(target / fmaxf(amax, 1e-30f))), -60.f), 100.f);
        s_scale = ldexpf(1.0f, (int)k);
        inv_out[0] = ldexpf(1.0f, 1 - (int)k);
    }
    __syncthreads();
    const float sc = s_scale;
    const long long npix = (long long)B * hw;
    for (long long i = threadIdx.x; i < npix; i += 1024) {
        const long long b = i / hw, r = i - b * hw;
        const float* src = g + b * 8 * hw + r;
        dwg_x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o.set(e, src[(long long)e * hw] * sc);
        o.store(dst + 8 * i);
    }
}
// dx [B,H,W,8] f32x (channels 0..2) -> d image [B,3,H,W] fp32, times inv[0]
__global__ __launch_bounds__(256) void k_vae_dx_unpack(long long npix, long long hw, const dwg_xs* __restrict__ dx, const float* __restrict__ inv,
                                                       float* __restrict__ out) {
    const float sc = inv[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long b = i / hw, r = i - b * hw;
        const dwg_x8 x = dwg_x8::load(dx + 8 * i);
        float* dst = out + b * 3 * hw + r;
#pragma unroll
        for (int e = 0; e < 3; e++) dst[e * hw] = x.get(e) * sc;
    }
}
// Range telemetry of a stored f32x tensor (round 5): the format saturates at +-65504 instead of overflowing (dwg_x_split) and loses significand
// bits once the hi half goes subnormal (|x| < 6.1e-5) -- silently, in the hot path.  This scan is the cold-path witness: it walks a tensor the
// plan has written and counts  [0] hi halves AT +-65504 (a saturated value, or a legitimate one on the edge: equally worth a warning),
// [1] non-zero values whose hi half is subnormal or zero, [2] non-finite hi halves, [3] max |x| (fp32 bits), [4] elements seen.
__global__ __launch_bounds__(256) void k_x_range_scan(long long n8, const dwg_xs* __restrict__ src, unsigned long long* __restrict__ out) {
    unsigned long long sat = 0, sub = 0, bad = 0;
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const dwg_x8 x = dwg_x8::load(src + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned short hb = __builtin_bit_cast(unsigned short, x.hi[e]), lb = __builtin_bit_cast(unsigned short, x.lo[e]);
            const unsigned ex = (hb >> 10) & 0x1fu, man = hb & 0x3ffu;
            if (ex == 0x1fu) { bad++; continue; }
            if ((hb & 0x7fffu) == 0x7bffu) sat++;
            if (ex == 0u && (man != 0u || (lb & 0x7fffu) != 0u)) sub++;
            amax = fmaxf(amax, fabsf(x.get(e)));
        }
    }
    // wave reduction, then one atomic per wave and counter (cold path)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sat += __shfl_xor((unsigned long long)sat, off); sub += __shfl_xor((unsigned long long)sub, off); bad += __shfl_xor((unsigned long long)bad, off);
        amax = fmaxf(amax, __shfl_xor(amax, off));
    }
    if ((threadIdx.x & 63) == 0) {
        if (sat) atomicAdd(&out[0], sat);
        if (sub) atomicAdd(&out[1], sub);
        if (bad) atomicAdd(&out[2], bad);
        atomicMax(&out[3], (unsigned long long)__float_as_uint(amax));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&out[4], (unsigned long long)n8 * 8ull);
}
__global__ __launch_bounds__(256) void k_add_x(long long n8, const dwg_xs* __restrict__ a, const dwg_xs* __restrict__ b, dwg_xs* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const dwg_x8 x = dwg_x8::load(a + 8 * i), y = dwg_x8::load(b + 8 * i);
        dwg_x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o.set(e, x.get(e) + y.get(e));
        o.store(out + 8 * i);
    }
}
// out[b][c][r] = in[b][r][c] for f32x tensors (the 8-channel groups run along c on the way in, along r on the way out): 64 x 64 logical
// tiles decoded into LDS as floats, re-split for the other axis.  R % 8 == 0, C % 8 == 0.
__global__ __launch_bounds__(256) void k_transpose_x(int R, int C, const dwg_xs* __restrict__ in, long long ld_in, long long bs_in,
                                                     dwg_xs* __restrict__ out, long long ld_out, long long bs_out) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    in += (long long)blockIdx.z * bs_in; out += (long long)blockIdx.z * bs_out;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = tid + i * 256, r = idx >> 3, ch = (idx & 7) * 8;
        if (r0 + r < R && c0 + ch < C) {
            const dwg_x8 v = dwg_x8::load(in + (long long)(r0 + r) * ld_in + c0 + ch);
#pragma unroll
            for (int e = 0; e < 8; e++) tile[r][ch + e] = v.get(e);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = tid + i * 256, c = idx >> 3, rh = (idx & 7) * 8;
        if (c0 + c >= C || r0 + rh >= R) continue;
        dwg_x8 o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.set(k, tile[rh + k][c]);
        o.store(out + (long long)(c0 + c) * ld_out + r0 + rh);
    }
}

static int grid_for(long long n) { long long b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (int)b; }

// out[b, 2i+py, 2j+px, :] = sub[py*2+px][b, i, j, :]  (16-byte pieces; C % 8 == 0)
__global__ __launch_bounds__(256) void k_interleave2x2(int B, int Ho, int Wo, int C8, const uint4* __restrict__ s00,
                                                       const uint4* __restrict__ s01, const uint4* __restrict__ s10,
                                                       const uint4* __restrict__ s11, uint4* __restrict__ out) {
    const long long n = (long long)B * Ho * Wo * C8 * 4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int c = (int)(t % C8);
        long long r = t / C8;                       // output pixel index over [B][2Ho][2Wo]
        const int x = (int)(r % (2 * Wo)); r /= 2 * Wo;
        const int y = (int)(r % (2 * Ho)); const int b = (int)(r / (2 * Ho));
        const uint4* src = (y & 1) ? ((x & 1) ? s11 : s10) : ((x & 1) ? s01 : s00);
        out[t] = src[(((long long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C8 + c];
    }
}


// out[b][c][r] = in[b][r][c] for 2-byte elements: 64 x 64 tiles through LDS, 16-byte reads along c and 16-byte writes along r.
// The LDS tile is [64 rows][64 + 2] halves: a thread's 8 column reads for one output piece step by 33 dwords -> distinct banks.
__global__ __launch_bounds__(256) void k_transpose16(int R, int C, const unsigned short* __restrict__ in, long long ld_in, long long bs_in,
                                                     unsigned short* __restrict__ out, long long ld_out, long long bs_out) {
    __shared__ unsigned short tile[64][66];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    in += (long long)blockIdx.z * bs_in; out += (long long)blockIdx.z * bs_out;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = tid + i * 256, r = idx >> 3, ch = (idx & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < R && c0 + ch < C) v = *reinterpret_cast<const uint4*>(in + (long long)(r0 + r) * ld_in + c0 + ch);   // C % 8 == 0
        unsigned* d = reinterpret_cast<unsigned*>(&tile[r][ch]);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int idx = tid + i * 256, c = idx >> 3, rh = (idx & 7) * 8;
        if (c0 + c >= C || r0 + rh >= R) continue;                       // R % 8 == 0
        unsigned short e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = tile[rh + k][c];
        uint4 v;
        v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
        *reinterpret_cast<uint4*>(out + (long long)(c0 + c) * ld_out + r0 + rh) = v;
    }
}

}  // namespace

extern "C" {

size_t dwg_mlp_wgrad_workspace_floats(int32_t M) {
    int rpb = 256, blocks = dwg_cdiv(M > 0 ? M : 1, rpb);
    return (size_t)blocks * 4096;
}

int dwg_mlp_wgrad(int32_t M, int32_t N, int32_t K, const float* dz, int32_t lddz, const float* x, int32_t ldx, float* dw,
                  int32_t lddw, float* workspace, dwg_stream_t stream) {
    if (M < 0 || N <= 0 || N > 64 || K <= 0 || K > 64 || !dz || !x || !dw || !workspace) return DWG_E_ARG;
    if (M == 0) {
        return DWG_OK;
    }
    const int rpb = 256, blocks = dwg_cdiv(M, rpb);
    DWG_LAUNCH("mlp_wgrad", k_mlp_wgrad_partial, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, N, K, dz, lddz, x, ldx, rpb, workspace);
    DWG_LAUNCH("mlp_wgrad_final", k_mlp_wgrad_final, dim3(256), dim3(256), 0, (hipStream_t)stream, blocks, N, K, (const float*)workspace,
               dw, lddw);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_mlp_chain_forward(int32_t M, int32_t Kin, const float* x, int32_t ldx, int32_t nlayers, const float* const* weights,
                          const int32_t* ldw, const float* const* biases, const int32_t* widths, const int32_t* acts,
                          float* const* hidden, float* out, int32_t ldo, const float* extra, int32_t n_extra, dwg_stream_t stream) {
    if (M < 0 || nlayers < 1 || nlayers > MLPC_MAXL || Kin < 8 || Kin > 64 || (Kin & 7) || !weights || !ldw || !widths || !acts || n_extra < 0 ||
        (n_extra > 0 && !extra))
        return DWG_E_ARG;
    if (M == 0) return DWG_OK;          // an empty batch has no rows to read or write (x / out may be the null pointer of an empty tensor)
    if (!x || !out) return DWG_E_ARG;
    MlpChainP p;
    p.x = x; p.M = M; p.Kin = Kin; p.ldx = ldx; p.nlayers = nlayers; p.out = out; p.ldo = ldo;
    p.extra = n_extra > 0 ? extra : nullptr; p.n_extra = n_extra;
    int k = Kin;
    for (int l = 0; l < MLPC_MAXL; l++) {
        const bool on = l < nlayers;
        if (on && (widths[l] < 1 || widths[l] > 64 || !weights[l] || ldw[l] < k + (l == 0 ? n_extra : 0))) return DWG_E_ARG;
        if (on && l + 1 < nlayers && (widths[l] & 7)) return DWG_E_ARG;       // hidden widths (the next layer's K): multiples of 8
        if (on && acts[l] != 0 && acts[l] != 1 && acts[l] != 2 && acts[l] != 5) return DWG_E_ARG;
        p.W[l] = on ? weights[l] : nullptr; p.b[l] = (on && biases) ? biases[l] : nullptr;
        p.hidden[l] = (on && hidden && l + 1 < nlayers) ? hidden[l] : nullptr;
        p.ldw[l] = on ? ldw[l] : 0; p.N[l] = on ? widths[l] : 0; p.K[l] = on ? k : 0; p.act[l] = on ? acts[l] : 0;
        if (on) k = widths[l];
    }
    int wfloats = 0;
    for (int l = 0; l < nlayers; l++) wfloats += ((widths[l] + 31) & ~31) * MLPC_LD;
    p.kshift = -1;
    if (!(Kin & (Kin - 1)) && !(ldx & 3) && !((uintptr_t)x & 15)) { p.kshift = 0; while ((1 << p.kshift) < Kin) p.kshift++; }
    const size_t lds = ((size_t)wfloats + MLPC_MAXL * 64 + MLPC_WAVES * 32 * MLPC_LD) * sizeof(float);       // <= 150 KiB: one workgroup per CU
    // waves per workgroup: eight share one LDS copy of the weights when there are row groups for every SIMD of the chip (two waves per SIMD
    // hide each other's latencies); a small batch (the 10 k-Gaussian frames: 313 groups) runs four per workgroup -- twice the CUs, one wave per
    // SIMD, so that a group's serial chain of layers has the MFMA pipe to itself (c1: the deformation network's launch 54 us before)
    const int groups = dwg_cdiv(M, 32);
    p.waves = groups <= 1024 ? 4 : MLPC_WAVES;
    const int wgs = dwg_cdiv(groups, p.waves);
    // one activation for all hidden layers (the avatar's two networks: ReLU, leaky ReLU) and whether hidden activations are kept are
    // compile-time properties of the launch; anything else runs the generic instantiation
    int act_h = nlayers > 1 ? acts[0] : 1;
    for (int l = 1; l + 1 < nlayers; l++) if (acts[l] != act_h) act_h = -1;
    if (act_h != 1 && act_h != 2) act_h = -1;
    bool keep = false;
    for (int l = 0; l + 1 < nlayers; l++) keep = keep || p.hidden[l];
    for (int l = 0; l + 1 < nlayers; l++) if (keep && !p.hidden[l]) return DWG_E_ARG;        // all hidden activations or none
    typedef void (*kern_t)(MlpChainP, int);
    static const kern_t kerns[6] = {k_mlp_chain<1, false>, k_mlp_chain<1, true>, k_mlp_chain<2, false>, k_mlp_chain<2, true>,
                                    k_mlp_chain<-1, false>, k_mlp_chain<-1, true>};
    static bool attr_set = false;
    if (!attr_set) {
        for (int i = 0; i < 6; i++) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const kern_t kern = kerns[(act_h == 1 ? 0 : act_h == 2 ? 2 : 4) + (keep ? 1 : 0)];
    DWG_LAUNCH("mlp_chain_fwd", kern, dim3(wgs < 256 ? wgs : 256), dim3(64 * p.waves), lds, (hipStream_t)stream, p, wfloats);    // persistent
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

static int mlp_chain_bwd_blocks(int32_t M) {
    const int chunks = dwg_cdiv(M > 0 ? M : 1, 64);
    return chunks < 256 ? chunks : 256;
}

size_t dwg_mlp_chain_backward_workspace_floats(int32_t M, int32_t nlayers) {
    return (size_t)mlp_chain_bwd_blocks(M) * (size_t)(nlayers > 0 ? nlayers : 1) * MLPB_PART;
}

int dwg_mlp_chain_backward(int32_t M, int32_t Kin, const float* x, int32_t ldx, int32_t nlayers, const float* const* weights,
                           const int32_t* ldw, const int32_t* widths, const int32_t* acts, const float* const* hidden, const float* out,
                           int32_t ldo, const float* dy, int32_t lddy, float* dx, int32_t lddx, float* const* dw, const int32_t* lddw,
                           float* const* db, const int32_t* accumulate, const float* extra, int32_t n_extra, float* workspace,
                           dwg_stream_t stream) {
    if (M < 0 || nlayers < 1 || nlayers > MLPC_MAXL || Kin < 8 || Kin > 64 || (Kin & 7) || !x || !weights || !ldw || !widths || !acts || !dy ||
        !dw || !lddw || !workspace || n_extra < 0 || (n_extra > 0 && !extra))
        return DWG_E_ARG;
    if (nlayers > 1 && !hidden) return DWG_E_ARG;
    MlpChainBwdP p;
    MlpChainBwdOut o;
    p.x = x; p.M = M; p.Kin = Kin; p.ldx = ldx; p.nlayers = nlayers; p.out = out; p.ldo = ldo; p.dy = dy; p.lddy = lddy; p.dx = dx; p.lddx = lddx;
    p.partial = workspace;
    o.extra = n_extra > 0 ? extra : nullptr; o.n_extra = n_extra;
    int k = Kin;
    for (int l = 0; l < MLPC_MAXL; l++) {
        const bool on = l < nlayers;
        if (on && (widths[l] < 1 || widths[l] > 64 || !weights[l] || ldw[l] < k || !dw[l] || lddw[l] < k + (l == 0 ? n_extra : 0))) return DWG_E_ARG;
        if (on && l + 1 < nlayers && ((widths[l] & 7) || !hidden[l])) return DWG_E_ARG;
        if (on && acts[l] != 0 && acts[l] != 1 && acts[l] != 2 && acts[l] != 5) return DWG_E_ARG;
        p.W[l] = on ? weights[l] : nullptr; p.hidden[l] = (on && l + 1 < nlayers) ? hidden[l] : nullptr;
        p.ldw[l] = on ? ldw[l] : 0; p.N[l] = on ? widths[l] : 0; p.K[l] = on ? k : 0; p.act[l] = on ? acts[l] : 0;
        o.dw[l] = on ? dw[l] : nullptr; o.db[l] = (on && db) ? db[l] : nullptr; o.lddw[l] = on ? lddw[l] : 0; o.N[l] = p.N[l]; o.K[l] = p.K[l];
        o.acc[l] = (on && accumulate) ? accumulate[l] : 0;
        if (on) k = widths[l];
    }
    if (acts[nlayers - 1] != 0 && !out) return DWG_E_ARG;
    if (dx && lddx < Kin) return DWG_E_ARG;
    if (M == 0) return DWG_OK;      // the caller zero-fills the gradients of an empty batch
    const size_t lds = (size_t)(nlayers + 2) * 64 * MLPB_LD * sizeof(float);           // <= 139 KiB: one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int blocks = mlp_chain_bwd_blocks(M);
    DWG_LAUNCH("mlp_chain_bwd", k_mlp_chain_bwd, dim3(blocks), dim3(256), lds, (hipStream_t)stream, p);
    DWG_LAUNCH("mlp_chain_bwd_final", k_mlp_chain_bwd_final, dim3(260, nlayers), dim3(256), 0, (hipStream_t)stream, blocks, nlayers,
               (const float*)workspace, o);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_concat_channels(int64_t rows, int32_t Ca, int32_t Cb, const void* a, const void* b, void* out, dwg_stream_t stream) {
    if (rows < 0 || Ca <= 0 || Cb <= 0 || Ca % 8 || Cb % 8 || !a || !b || !out) return DWG_E_ARG;
    if (rows == 0) return DWG_OK;
    DWG_LAUNCH("concat_channels", k_concat_channels, dim3(grid_for(rows * ((Ca + Cb) / 8))), dim3(256), 0, (hipStream_t)stream,
               (long long)rows, Ca, Cb, (const __bf16*)a, (const __bf16*)b, (__bf16*)out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_add_bf16(int64_t n, const void* a, const void* b, void* out, dwg_stream_t stream) {
    if (n < 0 || n % 8 || !a || !b || !out) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    DWG_LAUNCH("add_bf16", k_add_bf16, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const __bf16*)a,
               (const __bf16*)b, (__bf16*)out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_add_dt(int32_t dtype, int64_t n, const void* a, const void* b, void* out, dwg_stream_t stream) {
    if (dtype == DWG_DTYPE_BF16) return dwg_add_bf16(n, a, b, out, stream);
    if (n < 0 || n % 8 || !a || !b || !out) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    if (dtype == DWG_DTYPE_F16)
        DWG_LAUNCH("add_f16", k_add_t<_Float16>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const _Float16*)a,
                   (const _Float16*)b, (_Float16*)out);
    else if (dtype == DWG_DTYPE_F32)
        DWG_LAUNCH("add_f32", k_add_t<float>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const float*)a,
                   (const float*)b, (float*)out);
    else if (dtype == DWG_DTYPE_F32X)
        DWG_LAUNCH("add_f32x", k_add_x, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const dwg_xs*)a,
                   (const dwg_xs*)b, (dwg_xs*)out);
    else return DWG_E_ARG;
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_cast_f32_to_dt(int32_t dtype, int64_t n, const float* src, void* dst, dwg_stream_t stream) {
    if (dtype == DWG_DTYPE_BF16) return dwg_cast_f32_to_bf16(n, src, dst, stream);
    if (n < 0 || !src || !dst) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    if (dtype == DWG_DTYPE_F16)
        DWG_LAUNCH("cast_f32_f16", k_cast_f32_t<_Float16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, src, (_Float16*)dst);
    else if (dtype == DWG_DTYPE_F32)
        DWG_LAUNCH("cast_f32_f32", k_cast_f32_t<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, src, (float*)dst);
    else if (dtype == DWG_DTYPE_F32X) return dwg_xfmt_pack(n, src, dst, stream);
    else return DWG_E_ARG;
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_cast_f32_to_bf16(int64_t n, const float* src, void* dst, dwg_stream_t stream) {
    if (n < 0 || !src || !dst) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    DWG_LAUNCH("cast_f32_bf16", k_cast_f32_bf16, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, src, (__bf16*)dst);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_act_backward_colsum(int32_t M, int32_t N, int32_t act, const float* dy, const float* y, float* dz, float* colsum,
                            dwg_stream_t stream) {
    if (M < 0 || N <= 0 || N > 256 || !dy) return DWG_E_ARG;
    if (M == 0) return DWG_OK;
    int rows_per_block = 128;      // ~800 workgroups at 1e5 rows: the pass is a pure stream, it needs the whole chip
    DWG_LAUNCH("act_bwd_colsum", k_act_bwd_colsum, dim3(dwg_cdiv(M, rows_per_block)), dim3(256), 0, (hipStream_t)stream, M, N,
               act, dy, y, dz, colsum, rows_per_block);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int32_t step, float grad_scale, dwg_stream_t stream) {
    if (n < 0 || step < 1) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return DWG_E_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) return DWG_E_ARG;
    // The step's two scalars in DOUBLE from the fp32 hyper-parameters, rounded once -- the very statements FlatOptimizer.prepare_step makes
    // on the host for the device-resident table of a captured step (dwg_adam_step_dev): an eager step and a replay of the captured one
    // then run k_adam on identical bits (round 5: with the table gradient deterministic, this was the last difference between the two).
    const float stepsize = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    const float bc1 = 1.f;
    float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    lr = stepsize;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    DWG_LAUNCH("adam_step", k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (size_t)n, param, grad, exp_avg,
               exp_avg_sq, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, (const float*)nullptr);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_adam_step_dev(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* hyper, float beta1,
                      float beta2, float eps, dwg_stream_t stream) {
    if (n < 0 || !hyper) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return DWG_E_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) return DWG_E_ARG;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    DWG_LAUNCH("adam_step", k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (size_t)n, param, grad, exp_avg,
               exp_avg_sq, 0.f, beta1, beta2, eps, 1.f, 1.f, 1.f, hyper);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_adam_step_groups_dev(int32_t count, const dwg_adam_group* groups, const float* hyper, dwg_stream_t stream) {
    if (count < 0 || count > DWG_ADAM_MAX_GROUPS || !hyper || (count > 0 && !groups)) return DWG_E_ARG;
    AdamGroups G;
    size_t blocks = 0;
    int used = 0;
    for (int i = 0; i < count; i++) {
        const dwg_adam_group& a = groups[i];
        if (a.n < 0 || a.hyper_row < 0) return DWG_E_ARG;
        if (a.n == 0) continue;
        if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq) return DWG_E_ARG;
        if (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) % 16) return DWG_E_ARG;
        G.g[used++] = a;
        size_t b = ((size_t)a.n / 4 + 255) / 256;
        if (b < 1) b = 1;
        blocks = b > blocks ? b : blocks;
    }
    if (used == 0) return DWG_OK;
    if (blocks > 2048) blocks = 2048;
    DWG_LAUNCH("adam_step", k_adam_groups, dim3((unsigned)blocks, used), dim3(256), 0, (hipStream_t)stream, G, hyper);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_interleave2x2(int32_t B, int32_t Ho, int32_t Wo, int32_t C, const void* s00, const void* s01, const void* s10, const void* s11,
                      void* out, dwg_stream_t stream) {
    if (B < 0 || Ho < 0 || Wo < 0 || C <= 0 || C % 8) return DWG_E_ARG;
    const long long n = (long long)B * Ho * Wo * (C / 8) * 4;
    if (n == 0) return DWG_OK;
    if (!s00 || !s01 || !s10 || !s11 || !out) return DWG_E_ARG;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    DWG_LAUNCH("interleave2x2", k_interleave2x2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B, Ho, Wo, C / 8, (const uint4*)s00,
               (const uint4*)s01, (const uint4*)s10, (const uint4*)s11, (uint4*)out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_xfmt_pack(int64_t n, const float* src, void* dst, dwg_stream_t stream) {
    if (n < 0 || n % 8 || !src || !dst || ((uintptr_t)src | (uintptr_t)dst) % 16) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    DWG_LAUNCH("xfmt_pack", k_x_pack, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), src, (dwg_xs*)dst);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_xfmt_unpack(int64_t n, const void* src, float* dst, dwg_stream_t stream) {
    if (n < 0 || n % 8 || !src || !dst || ((uintptr_t)src | (uintptr_t)dst) % 16) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    DWG_LAUNCH("xfmt_unpack", k_x_unpack, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const dwg_xs*)src, dst);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_vae_image_pack(int32_t B, int32_t H, int32_t W, const float* image_nchw, void* x_xs, dwg_stream_t stream) {
    if (B < 0 || H < 0 || W < 0 || !image_nchw || !x_xs || ((uintptr_t)x_xs % 16)) return DWG_E_ARG;
    const long long hw = (long long)H * W, npix = (long long)B * hw;
    if (npix == 0) return DWG_OK;
    DWG_LAUNCH("vae_image_pack", k_vae_image_pack, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, npix, hw, image_nchw, (dwg_xs*)x_xs);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_vae_grad_prescale_pack(int32_t B, int32_t hw, const float* g_nchw, float target, void* dst_xs, float* inv_out, dwg_stream_t stream) {
    if (B <= 0 || hw <= 0 || !g_nchw || !dst_xs || !inv_out || ((uintptr_t)dst_xs % 16)) return DWG_E_ARG;
    DWG_LAUNCH("vae_grad_prescale_pack", k_vae_grad_prescale_pack, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, hw, g_nchw, target,
               (dwg_xs*)dst_xs, inv_out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_vae_dx_unpack(int32_t B, int32_t H, int32_t W, const void* dx_xs, const float* inv, float* out_nchw, dwg_stream_t stream) {
    if (B < 0 || H < 0 || W < 0 || !dx_xs || !inv || !out_nchw || ((uintptr_t)dx_xs % 16)) return DWG_E_ARG;
    const long long hw = (long long)H * W, npix = (long long)B * hw;
    if (npix == 0) return DWG_OK;
    DWG_LAUNCH("vae_dx_unpack", k_vae_dx_unpack, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, npix, hw, (const dwg_xs*)dx_xs, inv, out_nchw);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_xfmt_range_scan(int64_t n, const void* src, uint64_t* counters5, dwg_stream_t stream) {
    if (n < 0 || n % 8 || !src || !counters5 || ((uintptr_t)src % 16) || ((uintptr_t)counters5 % 8)) return DWG_E_ARG;
    if (n == 0) return DWG_OK;
    DWG_LAUNCH("xfmt_range_scan", k_x_range_scan, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 8), (const dwg_xs*)src,
               (unsigned long long*)counters5);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_transpose_dt(int32_t dtype, int32_t batch, int32_t R, int32_t C, const void* in, int64_t ld_in, int64_t batch_stride_in, void* out,
                     int64_t ld_out, int64_t batch_stride_out, dwg_stream_t stream) {
    if (dtype == DWG_DTYPE_BF16 || dtype == DWG_DTYPE_F16)
        return dwg_transpose_2byte(batch, R, C, in, ld_in, batch_stride_in, out, ld_out, batch_stride_out, stream);
    if (dtype != DWG_DTYPE_F32X) return DWG_E_ARG;
    if (batch < 0 || R < 0 || C < 0 || R % 8 || C % 8 || ld_in % 8 || ld_out % 8 || batch_stride_in % 8 || batch_stride_out % 8) return DWG_E_ARG;
    if (batch == 0 || R == 0 || C == 0) return DWG_OK;
    if (!in || !out || ((uintptr_t)in | (uintptr_t)out) % 16) return DWG_E_ARG;
    DWG_LAUNCH("transpose_x", k_transpose_x, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, R, C,
               (const dwg_xs*)in, (long long)ld_in, (long long)batch_stride_in, (dwg_xs*)out, (long long)ld_out, (long long)batch_stride_out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_transpose_2byte(int32_t batch, int32_t R, int32_t C, const void* in, int64_t ld_in, int64_t batch_stride_in, void* out, int64_t ld_out,
                        int64_t batch_stride_out, dwg_stream_t stream) {
    if (batch < 0 || R < 0 || C < 0 || R % 8 || C % 8 || ld_in % 8 || ld_out % 8 || batch_stride_in % 8 || batch_stride_out % 8) return DWG_E_ARG;
    if (batch == 0 || R == 0 || C == 0) return DWG_OK;
    if (!in || !out || ((uintptr_t)in | (uintptr_t)out) % 16) return DWG_E_ARG;
    DWG_LAUNCH("transpose", k_transpose16, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, R, C,
               (const unsigned short*)in, (long long)ld_in, (long long)batch_stride_in, (unsigned short*)out, (long long)ld_out,
               (long long)batch_stride_out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
