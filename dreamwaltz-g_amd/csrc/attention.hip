// attention.hip -- fused (flash-style) multi-head attention forward for gfx950: bf16 in/out, fp32 online softmax.
//
// Used for every attention site of the SD-1.5 UNet / ControlNet (self: N = 4096/1024/256/64 tokens, cross: 77 keys;
// 8 heads of dim 40/80/160) inside boundary B4 (controlnet.py:98-114).  The reference gets the same math from
// diffusers' AttnProcessor2_0 -> F.scaled_dot_product_attention.
//
// Wave64 design (not a warp-shaped tiling):
//   * workgroup = 4 waves, each wave owns 32 queries; K/V tiles of 32 keys are staged once per workgroup in LDS.
//   * S^T = K Q^T is computed "swapped" with v_mfma_f32_32x32x16_bf16 (A = K tile rows, B = Q rows held in VGPRs), so a
//     lane holds ONE query column: its 16 accumulator registers are 16 of the 32 keys of the tile, the other 16 sit in
//     lane^32.  Row max / row sum are therefore 15 in-lane ops + one cross-half exchange -- no LDS round trip.
//   * O^T += V^T P^T reuses the probabilities straight from those registers as the MFMA B operand: the contraction over
//     keys is order-free, so V^T is read from LDS in the SAME key permutation the accumulator layout implies
//     (two 8-byte reads per fragment) instead of shuffling P between lanes.
//   * per-query rescale of O is a per-lane scalar multiply (each lane owns one query column of O^T).
#include "dwg_common.h"
#include <cstdlib>
#include "dwg_prof_internal.h"
#include "../../include/dwg_nn.h"

// 16-bit operand type of this translation unit: attention.hip is the bf16 unit, attention_f16.hip re-includes it with DWG_ATTN_F16_TU
// defined (same kernels on _Float16, v_mfma_f32_32x32x16_f16) for the fp16-storage plans.
#ifdef DWG_ATTN_F16_TU
typedef _Float16 HT;
#define DWG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 HT;
#define DWG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) HT bf16x8;
typedef __attribute__((ext_vector_type(4))) HT bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct AttnP {
    const HT* Q; const HT* K; const HT* V; HT* O;
    int Nq, Nk, H, d;
    long long ldq, ldk, ldv, ldo;          // row strides (elements)
    long long bq, bk, bv, bo;              // per-image strides (elements); head h starts at column h*d
    float scale_log2;                      // softmax scale * log2(e)
};

// DK = head dim padded to a multiple of 16 (contraction of QK^T), DV = padded to a multiple of 32 (rows of O^T).
// LD > 0: the head dim is exactly LD < DV, and the padding row LD of V^T holds ONES: the PV MFMA (which multiplies the padding rows anyway)
// then accumulates the softmax denominator l = sum_k P[q][k] in accumulator row LD -- rescaled with O for free -- and the 16 adds + the
// cross-half exchange of the row sum leave the VALU-bound softmax segment.  (l sums the bf16-rounded P the MFMA uses, not the fp32 one.)
template <int DK, int DV, int LD = 0>
__global__ __launch_bounds__(256, 2) void k_flash_fwd(AttnP p) {   // 2 waves per SIMD = at most 256 registers: the compiler then keeps the O accumulators in
                                                                       // arch VGPRs (MFMA VGPR form).  With 512 allowed it parks them in AGPRs and moves all of them out and back
                                                                       // around the conditional rescale EVERY tile (112 v_accvgpr copies of 352 loop instructions at d = 40)
    constexpr int KT = 32;                 // keys per tile
    constexpr int LDK = DK + 8;            // K tile row stride (bf16), 16-byte aligned rows
    constexpr int LDV = KT + 8;            // V^T tile row stride
    constexpr int NKS = DK / 16, NVB = DV / 32;
    __shared__ __attribute__((aligned(16))) HT sK[KT * LDK];
    __shared__ __attribute__((aligned(16))) HT sVt[DV * LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ql = lane & 31;
    const int img = blockIdx.y / p.H, head = blockIdx.y % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const HT* Q = p.Q + img * p.bq + (long long)head * p.d;
    const HT* K = p.K + img * p.bk + (long long)head * p.d;
    const HT* V = p.V + img * p.bv + (long long)head * p.d;
    HT* O = p.O + img * p.bo + (long long)head * p.d;

    // this lane's query row as MFMA B-operand fragments: element e of step s = Q[q][16 s + 8 half + e]
    bf16x8 qf[NKS];
    {
        const int q = q0 + ql;
#pragma unroll
        for (int s = 0; s < NKS; s++) {
            const int c = 16 * s + 8 * half;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (HT)0.f;
            if (q < p.Nq && c < p.d) v = *reinterpret_cast<const bf16x8*>(Q + (long long)q * p.ldq + c);   // d % 8 == 0
            qf[s] = v;
        }
    }
    f32x16 acc[NVB];
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;

    const int ntiles = (p.Nk + KT - 1) / KT;
    // K / V staging is software-pipelined through registers: the global loads of tile t+1 are issued right after tile t has
    // been written to LDS and complete while tile t is being multiplied (the unpipelined version exposed one L2/HBM round
    // trip per 32-key tile -- 128 of them per workgroup at 4096 keys).
    constexpr int NKC = (KT * (DK / 8) + 255) / 256, NVC = (KT * (DV / 8) + 255) / 256;
    bf16x8 kreg[NKC], vreg[NVC];
    // per-thread staging coordinates are tile-invariant: (key, dc) and the matching global / LDS addresses are computed once
    int kkey[NKC], vkey[NVC];
    bool vone[NVC];
    const HT* kptr[NKC]; const HT* vptr[NVC];
    HT* klds[NKC]; HT* vlds[NVC];
#pragma unroll
    for (int i = 0; i < NKC; i++) {
        const int c = tid + i * 256;
        const int key = c / (DK / 8), dc = (c % (DK / 8)) * 8;
        const bool on = c < KT * (DK / 8) && dc < p.d;
        kkey[i] = on ? key : (1 << 30);                       // disabled lanes never pass the key-range test
        kptr[i] = K + (long long)key * p.ldk + dc;
        klds[i] = c < KT * (DK / 8) ? &sK[key * LDK + dc] : nullptr;
    }
#pragma unroll
    for (int i = 0; i < NVC; i++) {
        const int c = tid + i * 256;
        const int key = c % KT, dc = (c / KT) * 8;            // consecutive threads -> consecutive keys: conflict-light transposed writes
        const bool on = c < KT * (DV / 8) && dc < p.d;
        vkey[i] = on ? key : (1 << 30);
        vone[i] = LD > 0 && c < KT * (DV / 8) && dc == LD;      // the chunk whose first element is row LD of V^T
        vptr[i] = V + (long long)key * p.ldv + dc;
        vlds[i] = c < KT * (DV / 8) ? &sVt[dc * LDV + key] : nullptr;
    }
    auto fetch = [&](int k0) {
        bf16x8 z, one0;
#pragma unroll
        for (int e = 0; e < 8; e++) { z[e] = (HT)0.f; one0[e] = (HT)(e == 0 ? 1.f : 0.f); }
#pragma unroll
        for (int i = 0; i < NKC; i++)
            kreg[i] = (long long)k0 + kkey[i] < p.Nk ? *reinterpret_cast<const bf16x8*>(kptr[i] + (long long)k0 * p.ldk) : z;
#pragma unroll
        for (int i = 0; i < NVC; i++)
            vreg[i] = (long long)k0 + vkey[i] < p.Nk ? *reinterpret_cast<const bf16x8*>(vptr[i] + (long long)k0 * p.ldv) : (vone[i] ? one0 : z);
    };
    if (ntiles > 0) fetch(0);
    for (int t = 0; t < ntiles; t++) {
        const int k0 = t * KT;
        __syncthreads();   // previous tile fully consumed
        // stage K tile [key][d] (zero padded) and V tile transposed [d][key] from the prefetched registers
#pragma unroll
        for (int i = 0; i < NKC; i++)
            if (klds[i]) *reinterpret_cast<bf16x8*>(klds[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < NVC; i++)
            if (vlds[i]) {
#pragma unroll
                for (int e = 0; e < 8; e++) vlds[i][e * LDV] = vreg[i][e];
            }
        __syncthreads();
        if (t + 1 < ntiles) fetch(k0 + KT);      // in flight during the MFMA / softmax work below
        // S^T tile: rows = keys, cols = queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[ql * LDK + 16 * ks + 8 * half]);
            s = DWG_MFMA16(kf, qf[ks], s);
        }
        // online softmax for this lane's query; register r <-> key k0 + (r&3) + 8*(r>>2) + 4*half.
        // The scale (> 0) is folded into the exponent (one fma per element); only the last, partial tile needs key masking;
        // the accumulator rescale is skipped while no lane of the wave has seen a new maximum (the common case after the
        // first tiles).
        if (k0 + KT > p.Nk) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= p.Nk) s[r] = -3.0e38f;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * p.scale_log2);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -m_new));   // raw v_exp_f32: arguments <= 0, underflow -> 0
            if constexpr (LD == 0) rs += s[r];
        }
        if constexpr (LD == 0) rs += __shfl_xor(rs, 32);
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] *= alpha;
        }
        l_run += rs;
        m_run = m_new;
        // P^T as B operand: step st uses registers 8 st .. 8 st + 7 of this lane
        bf16x8 pf[2];
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int e = 0; e < 8; e++) pf[st][e] = (HT)s[8 * st + e];
        // O^T += V^T P^T ; A operand row = dv, elements follow the same key permutation:
        //   e in 0..3 -> key 16 st + 4 half + e ; e in 4..7 -> key 16 st + 8 + 4 half + (e - 4)
#pragma unroll
        for (int j = 0; j < NVB; j++) {
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const HT* vrow = &sVt[(32 * j + ql) * LDV + 16 * st + 4 * half];
                bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow);
                bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + 8);
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                acc[j] = DWG_MFMA16(vf, pf[st], acc[j]);
            }
        }
    }
    // epilogue: O[q][dv] = acc / l ; lane owns query column ql, register r of block j <-> dv = 32 j + (r&3) + 8 (r>>2) + 4 half.
    // Stage through LDS so that rows go out as contiguous 16-byte stores.
    constexpr int LDO = DV + 8;
    __shared__ __attribute__((aligned(16))) HT sOut[4 * 32 * LDO];
    if constexpr (LD > 0) {
        // accumulator row LD: block LD / 32, register r with (r&3) + 8 (r>>2) + 4 half == LD % 32, held by the lanes of that half
        constexpr int W = LD % 32, R = (W & 3) + 4 * (W >> 3), HL = (W >> 2) & 1;
        l_run = __shfl(acc[LD / 32][R], ql + 32 * HL);
    }
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    HT* myO = sOut + wave * 32 * LDO;
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int dv = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
            myO[ql * LDO + dv] = (HT)(acc[j][r] * inv);
        }
    __syncthreads();
    for (int c = lane; c < 32 * (p.d / 8); c += 64) {
        int q = c / (p.d / 8), dc = (c % (p.d / 8)) * 8;
        if (q0 + q < p.Nq)
            *reinterpret_cast<bf16x8*>(O + (long long)(q0 + q) * p.ldo + dc) = *reinterpret_cast<const bf16x8*>(&myO[q * LDO + dc]);
    }
}

}  // namespace

extern "C" {

#ifdef DWG_ATTN_F16_TU
int dwg_attention_forward_f16(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                              const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                              int64_t bo, float scale, dwg_stream_t stream_) {
#else
int dwg_attention_forward_f16(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                              const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                              int64_t bo, float scale, dwg_stream_t stream_);         // attention_f16.hip

int dwg_attention_forward_x(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                            const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                            int64_t bo, float scale, dwg_stream_t stream_);           // attention_x.hip (split-precision operands)

size_t dwg_attention_split_workspace_bytes_x(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d);          // attention_x.hip
int dwg_attention_forward_x_ws(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq, const void* K,
                               int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo, int64_t bo, float scale,
                               void* workspace, size_t workspace_bytes, dwg_stream_t stream_);

size_t dwg_attention_split_workspace_bytes(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d) {
    return dtype == DWG_DTYPE_F32X ? dwg_attention_split_workspace_bytes_x(B, H, Nq, Nk, d) : 0;
}

int dwg_attention_forward_ws(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                             const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                             int64_t bo, float scale, void* workspace, size_t workspace_bytes, dwg_stream_t stream) {
    if (dtype == DWG_DTYPE_F32X)
        return dwg_attention_forward_x_ws(B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, workspace, workspace_bytes, stream);
    return dwg_attention_forward_dt(dtype, B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, stream);
}

int dwg_attention_forward_dt(int32_t dtype, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                             const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                             int64_t bo, float scale, dwg_stream_t stream) {
    if (dtype == DWG_DTYPE_F16) return dwg_attention_forward_f16(B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, stream);
    if (dtype == DWG_DTYPE_F32X) return dwg_attention_forward_x(B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, stream);
    if (dtype != DWG_DTYPE_BF16) return DWG_E_ARG;          // fp32 plans run attention as QK^T -> softmax -> PV on dwg_gemm
    return dwg_attention_forward(B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, stream);
}

int dwg_attention_forward(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq,
                          const void* K, int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo,
                          int64_t bo, float scale, dwg_stream_t stream_) {
#endif
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || d % 8 || d > 160 || !Q || !K || !V || !O) return DWG_E_ARG;
    if ((ldq | ldk | ldv | ldo | bq | bk | bv | bo) % 8) return DWG_E_ARG;   // 16-byte aligned rows
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) % 16) return DWG_E_ARG;
    AttnP p{(const HT*)Q, (const HT*)K, (const HT*)V, (HT*)O, Nq, Nk, H, d, ldq, ldk, ldv, ldo, bq, bk, bv, bo,
            scale * 1.4426950408889634f};
    dim3 grid(dwg_cdiv(Nq, 128), B * H), block(256);
    hipStream_t stream = (hipStream_t)stream_;
    // algorithmic flops of the launch (QK^T and PV on the logical head size; the padded tile columns are not counted)
    const double flops = 4.0 * B * H * (double)Nq * Nk * d;
    static const bool no_lrow = getenv("DWG_ATTN_NO_LROW") != nullptr;       // A/B switch: softmax denominator summed on the VALU
    if (d <= 32) DWG_LAUNCH_W("flash_attn_d32", "k_flash_fwd<32, 32, 0>", flops, (k_flash_fwd<32, 32>), grid, block, 0, stream, p);
    else if (d == 40 && !no_lrow) DWG_LAUNCH_W("flash_attn_d48", "k_flash_fwd<48, 64, 40>", flops, (k_flash_fwd<48, 64, 40>), grid, block, 0, stream, p);
    else if (d <= 48) DWG_LAUNCH_W("flash_attn_d48", "k_flash_fwd<48, 64, 0>", flops, (k_flash_fwd<48, 64>), grid, block, 0, stream, p);
    else if (d <= 64) DWG_LAUNCH_W("flash_attn_d64", "k_flash_fwd<64, 64, 0>", flops, (k_flash_fwd<64, 64>), grid, block, 0, stream, p);
    else if (d == 80 && !no_lrow) DWG_LAUNCH_W("flash_attn_d96", "k_flash_fwd<96, 96, 80>", flops, (k_flash_fwd<96, 96, 80>), grid, block, 0, stream, p);
    else if (d <= 96) DWG_LAUNCH_W("flash_attn_d96", "k_flash_fwd<96, 96, 0>", flops, (k_flash_fwd<96, 96>), grid, block, 0, stream, p);
    else DWG_LAUNCH_W("flash_attn_d160", "k_flash_fwd<160, 160, 0>", flops, (k_flash_fwd<160, 160>), grid, block, 0, stream, p);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
