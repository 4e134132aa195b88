// elementwise.hip -- small bandwidth-bound helpers around the GEMM primitive (activation backward + bias gradient,
// fused multi-tensor Adam).  Each is a grid-stride kernel with 16-byte accesses where the layout allows.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_elementwise.h"

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
    __shared__ float part[256];
    const int tid = threadIdx.x;
    const int tpr = N;                       // threads per row (one thread per column)
    const int rows_par = 256 / tpr;          // rows processed in parallel
    const int col = tid % tpr, rsub = tid / tpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
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
__global__ __launch_bounds__(256) void k_adam(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2,
                                              float eps, float bc1, float bc2_sqrt, float grad_scale) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    const float step = lr / bc1;
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
    // 64 outputs per workgroup x 4 strands over the partial tiles; fixed combination order
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, strand = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + o;
    float s = 0.f;
    for (int b = strand; b < blocks; b += 4) s += partial[(size_t)b * 4096 + e];
    part[strand][o] = s;
    __syncthreads();
    if (strand == 0) {
        const int n = e >> 6, k = e & 63;
        if (n < N && k < K) dw[(size_t)n * lddw + k] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
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
    DWG_LAUNCH("mlp_wgrad_final", k_mlp_wgrad_final, dim3(64), dim3(256), 0, (hipStream_t)stream, blocks, N, K, (const float*)workspace,
               dw, lddw);
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
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    DWG_LAUNCH("adam_step", k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (size_t)n, param, grad, exp_avg,
               exp_avg_sq, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale);
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

}  // extern "C"
