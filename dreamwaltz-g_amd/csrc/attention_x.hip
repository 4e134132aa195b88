// attention_x.hip -- fused (flash-style) multi-head attention forward on SPLIT-PRECISION ("f32x", dwg_xfmt.h) operands: fp32-grade
// scores, probabilities and outputs at the 16-bit MFMA rate.  Same wave64 design as attention.hip (one lane = one query column of
// S^T = K Q^T, online softmax in registers, O^T += V^T P^T with P taken straight from the accumulator registers), with every product
// formed from three v_mfma_f32_32x32x16_f16:   a b ~= ah bh + 2^-11 (al bh + ah bl)
//   * Q, K, V arrive as hi / lo fp16 planes (32 bytes per 8 channels): the staging pass writes the two planes of a K / V^T tile to
//     separate LDS images, the fragment reads are the bf16 kernel's;
//   * S^T is accumulated in two register sets (main, cross) and joined before the softmax;
//   * P (in (0, 1]) is split in registers: ph = fp16(p), pl = fp16((p - ph) 2^11) -- four VALU instructions per score next to the six
//     of the softmax;
//   * O^T likewise in two accumulator sets, joined in the epilogue, which splits the result into the output's planes.
// Used by the "f32x" denoiser plans for every attention site of the SD-1.5 UNet / ControlNet (boundary B4, controlnet.py:98-114); the
// reference runs these sites in fp32 (configs/__init__.py:236,241) through diffusers' scaled_dot_product_attention.
#include "dwg_common.h"
#include <cstdlib>
#include "dwg_prof_internal.h"
#include "../../include/dwg_nn.h"
#include "dwg_xfmt.h"

namespace {

typedef _Float16 HT;
typedef __attribute__((ext_vector_type(8))) HT h8;
typedef __attribute__((ext_vector_type(4))) HT h4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

struct AttnXP {
    const dwg_xs* Q; const dwg_xs* K; const dwg_xs* V; dwg_xs* O;
    int Nq, Nk, H, d;
    long long ldq, ldk, ldv, ldo;          // row strides (logical elements)
    long long bq, bk, bv, bo;              // per-image strides; head h starts at column h*d
    float scale_log2;                      // softmax scale * log2(e)
    int nsplit;                            // > 1: the keys are split into nsplit ranges (blockIdx.z); a workgroup writes its UNNORMALISED
    float* ws;                             //      partial (O, running max, denominator) to ws[split][image*head][query][dvw] and
    int dvw;                               //      k_flash_merge_x combines the ranges in split order
};

// DK = head dim padded to a multiple of 16, DV = padded to a multiple of 32.  LD > 0: the head dim is exactly LD < DV and row LD of the hi
// plane of V^T holds ONES (lo plane: zeros), so the PV MFMAs accumulate the softmax denominator (of the split P the MFMAs use) in
// accumulator row LD of both sets -- see attention.hip.
template <int DK, int DV, int LD, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flash_fwd_x(AttnXP p) {
    constexpr int KT = 32;
    constexpr int LDK = DK + 8;
    constexpr int LDV = KT + 8;
    constexpr int NKS = DK / 16, NVB = DV / 32;
    constexpr int TILE_HALVES = 2 * KT * LDK + 2 * DV * LDV;               // K hi | K lo | V^T hi | V^T lo
    constexpr int ROWB = DV * 4 + 16;                                      // bytes of one staged output row (f32x) + pad
    constexpr int OUT_BYTES = 4 * 32 * ROWB;
    constexpr bool STAGE = OUT_BYTES <= 56 * 1024;                         // d = 160 (8x8 / 16x16 latents, a few workgroups): rows go out from registers
    constexpr int SMEM = (STAGE && OUT_BYTES > TILE_HALVES * 2) ? OUT_BYTES : TILE_HALVES * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    HT* sKh = reinterpret_cast<HT*>(smem);
    HT* sKl = sKh + KT * LDK;
    HT* sVh = sKl + KT * LDK;
    HT* sVl = sVh + DV * LDV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ql = lane & 31;
    const int img = blockIdx.y / p.H, head = blockIdx.y % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const dwg_xs* Q = p.Q + img * p.bq + (long long)head * p.d;
    const dwg_xs* K = p.K + img * p.bk + (long long)head * p.d;
    const dwg_xs* V = p.V + img * p.bv + (long long)head * p.d;
    dwg_xs* O = p.O + img * p.bo + (long long)head * p.d;

    // this lane's query row as MFMA B-operand fragments, both planes: element e of step s = Q[q][16 s + 8 half + e]
    h8 qh[NKS], qlo[NKS];
    {
        const int q = q0 + ql;
#pragma unroll
        for (int s = 0; s < NKS; s++) {
            const int c = 16 * s + 8 * half;
            dwg_x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) { v.hi[e] = (HT)0.f; v.lo[e] = (HT)0.f; }
            if (q < p.Nq && c < p.d) v = dwg_x8::load(Q + (long long)q * p.ldq + c);      // d % 8 == 0
            qh[s] = v.hi; qlo[s] = v.lo;
        }
    }
    f32x16 acc[NVB], acx[NVB];
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[j][r] = 0.f; acx[j][r] = 0.f; }
    float m_run = -3.0e38f, acc_l = 0.f;

    // this workgroup's key tiles: all of them, or range blockIdx.z of nsplit (small query counts: a serial walk over all keys by the few
    // workgroups there are left most of the chip idle -- 92 us for the 32x32 level's self-attention, 41 us for the 16x16 level's)
    const int ntiles_all = (p.Nk + KT - 1) / KT;
    const int tper = p.nsplit > 1 ? (ntiles_all + p.nsplit - 1) / p.nsplit : ntiles_all;
    const int tbeg = p.nsplit > 1 ? (int)blockIdx.z * tper : 0;
    const int ntiles = min(ntiles_all, tbeg + tper);
    constexpr int NKC = (KT * (DK / 8) + 255) / 256, NVC = (KT * (DV / 8) + 255) / 256;
    dwg_x8 kreg[NKC], vreg[NVC];
    int kkey[NKC], vkey[NVC];
    bool vone[NVC];
    const dwg_xs* kptr[NKC]; const dwg_xs* vptr[NVC];
    int klds[NKC], vlds[NVC];              // LDS offsets in halves (hi plane; the lo plane sits at a constant distance), -1: none
#pragma unroll
    for (int i = 0; i < NKC; i++) {
        const int c = tid + i * 256;
        const int key = c / (DK / 8), dc = (c % (DK / 8)) * 8;
        const bool on = c < KT * (DK / 8) && dc < p.d;
        kkey[i] = on ? key : (1 << 30);
        kptr[i] = K + (long long)key * p.ldk + dc;
        klds[i] = c < KT * (DK / 8) ? key * LDK + dc : -1;
    }
#pragma unroll
    for (int i = 0; i < NVC; i++) {
        const int c = tid + i * 256;
        const int key = c % KT, dc = (c / KT) * 8;
        const bool on = c < KT * (DV / 8) && dc < p.d;
        vkey[i] = on ? key : (1 << 30);
        vone[i] = LD > 0 && c < KT * (DV / 8) && dc == LD;
        vptr[i] = V + (long long)key * p.ldv + dc;
        vlds[i] = c < KT * (DV / 8) ? dc * LDV + key : -1;
    }
    auto fetch = [&](int k0) {
        dwg_x8 z, one0;
#pragma unroll
        for (int e = 0; e < 8; e++) { z.hi[e] = (HT)0.f; z.lo[e] = (HT)0.f; one0.hi[e] = (HT)(e == 0 ? 1.f : 0.f); one0.lo[e] = (HT)0.f; }
#pragma unroll
        for (int i = 0; i < NKC; i++)
            kreg[i] = (long long)k0 + kkey[i] < p.Nk ? dwg_x8::load(kptr[i] + (long long)k0 * p.ldk) : z;
#pragma unroll
        for (int i = 0; i < NVC; i++)
            vreg[i] = (long long)k0 + vkey[i] < p.Nk ? dwg_x8::load(vptr[i] + (long long)k0 * p.ldv) : (vone[i] ? one0 : z);
    };
    // K / V staging is software-pipelined through registers (tile t + 1 in flight while tile t is multiplied) -- except at d = 160, whose two
    // accumulator sets leave no registers for it: there the tile is fetched right before it is staged (8x8 / 16x16 latents: a few tiles)
    constexpr bool PF = DV <= 96;
    if (PF && tbeg < ntiles) fetch(tbeg * KT);
    for (int t = tbeg; t < ntiles; t++) {
        const int k0 = t * KT;
        if constexpr (!PF) fetch(k0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NKC; i++)
            if (klds[i] >= 0) {
                *reinterpret_cast<h8*>(sKh + klds[i]) = kreg[i].hi;
                *reinterpret_cast<h8*>(sKl + klds[i]) = kreg[i].lo;
            }
#pragma unroll
        for (int i = 0; i < NVC; i++)
            if (vlds[i] >= 0) {
#pragma unroll
                for (int e = 0; e < 8; e++) { sVh[vlds[i] + e * LDV] = vreg[i].hi[e]; sVl[vlds[i] + e * LDV] = vreg[i].lo[e]; }
            }
        __syncthreads();
        if (PF && t + 1 < ntiles) fetch(k0 + KT);
        // S^T tile: rows = keys, cols = queries; main and cross sets
        f32x16 s, sx;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = 0.f; sx[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            const h8 kh = *reinterpret_cast<const h8*>(&sKh[ql * LDK + 16 * ks + 8 * half]);
            const h8 kl = *reinterpret_cast<const h8*>(&sKl[ql * LDK + 16 * ks + 8 * half]);
            s = MFMA16(kh, qh[ks], s);
            sx = MFMA16(kl, qh[ks], sx);
            sx = MFMA16(kh, qlo[ks], sx);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = fmaf(sx[r], DWG_X_LO_INV, s[r]);
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
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -m_new));
            if constexpr (LD == 0) rs += s[r];
        }
        if constexpr (LD == 0) rs += __shfl_xor(rs, 32);
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) { acc[j][r] *= alpha; acx[j][r] *= alpha; }
            if constexpr (LD == 0) acc_l *= alpha;
        }
        if constexpr (LD == 0) acc_l += rs;
        m_run = m_new;
        // P^T as B operand, split into its planes: step st uses registers 8 st .. 8 st + 7 of this lane
        h8 ph[2], pl[2];
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float pv = s[8 * st + e];
                const HT a = (HT)pv;
                ph[st][e] = a; pl[st][e] = (HT)((pv - (float)a) * DWG_X_LO_SCALE);
            }
#pragma unroll
        for (int j = 0; j < NVB; j++) {
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const int off = (32 * j + ql) * LDV + 16 * st + 4 * half;
                const h4 a0 = *reinterpret_cast<const h4*>(sVh + off), a1 = *reinterpret_cast<const h4*>(sVh + off + 8);
                const h4 b0 = *reinterpret_cast<const h4*>(sVl + off), b1 = *reinterpret_cast<const h4*>(sVl + off + 8);
                h8 vh, vl;
                vh[0] = a0[0]; vh[1] = a0[1]; vh[2] = a0[2]; vh[3] = a0[3]; vh[4] = a1[0]; vh[5] = a1[1]; vh[6] = a1[2]; vh[7] = a1[3];
                vl[0] = b0[0]; vl[1] = b0[1]; vl[2] = b0[2]; vl[3] = b0[3]; vl[4] = b1[0]; vl[5] = b1[1]; vl[6] = b1[2]; vl[7] = b1[3];
                acc[j] = MFMA16(vh, ph[st], acc[j]);
                acx[j] = MFMA16(vl, ph[st], acx[j]);
                acx[j] = MFMA16(vh, pl[st], acx[j]);
            }
        }
    }
    // epilogue: join the sets, O[q][dv] = acc / l; lane owns query column ql, register r of block j <-> dv = 32 j + (r&3) + 8 (r>>2) + 4 half,
    // i.e. four consecutive channels (4 half .. 4 half + 3) of 8-group 4 j + (r >> 2): split and staged in LDS as f32x rows, then 16-byte stores.
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = fmaf(acx[j][r], DWG_X_LO_INV, acc[j][r]);
    float l_run;
    if constexpr (LD > 0) {
        constexpr int W = LD % 32, R = (W & 3) + 4 * (W >> 3), HL = (W >> 2) & 1;
        l_run = __shfl(acc[LD / 32][R], ql + 32 * HL);
    } else {
        l_run = acc_l;
    }
    if (p.nsplit > 1) {
        // partial result of this key range: unnormalised O (both sets joined), the running maximum (log2 domain, scaled) and the denominator
        const int q = q0 + ql;
        if (q < p.Nq) {
            float* row = p.ws + (((long long)blockIdx.z * gridDim.y + blockIdx.y) * p.Nq + q) * p.dvw;
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
                    *reinterpret_cast<float4*>(row + 32 * j + 8 * r4 + 4 * half) =
                        make_float4(acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]);
            if (half == 0) { row[DV] = m_run; row[DV + 1] = l_run; }
        }
        return;
    }
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    if constexpr (!STAGE) {
        const int q = q0 + ql;
        if (q < p.Nq) {
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const int dv = 32 * j + 8 * r4 + 4 * half;
                    if (dv < p.d) {
                        const float v[4] = {acc[j][4 * r4] * inv, acc[j][4 * r4 + 1] * inv, acc[j][4 * r4 + 2] * inv, acc[j][4 * r4 + 3] * inv};
                        dwg_x_put4(O + (long long)q * p.ldo, dv, v);
                    }
                }
        }
        return;
    }
    __syncthreads();                                   // the K / V tiles are dead: their LDS becomes the output stage
    unsigned char* myO = smem + wave * 32 * ROWB;
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            h4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; e++) { HT a, b; dwg_x_split(acc[j][4 * r4 + e] * inv, a, b); oh[e] = a; ol[e] = b; }
            unsigned char* g = myO + ql * ROWB + (4 * j + r4) * 32 + half * 8;
            *reinterpret_cast<h4*>(g) = oh; *reinterpret_cast<h4*>(g + 16) = ol;
        }
    __syncthreads();
    const int pieces = p.d / 4;                        // 16-byte pieces per row (two per 8-group)
    for (int c = lane; c < 32 * pieces; c += 64) {
        const int q = c / pieces, pc = c % pieces;
        if (q0 + q < p.Nq)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(O + (long long)(q0 + q) * p.ldo) + pc * 16) =
                *reinterpret_cast<const uint4*>(myO + q * ROWB + pc * 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the head sizes 40 / 80 (the 64x64 and 32x32 levels: 4096 / 1024 keys of self-attention, most of the attention time of a step).
// k_flash_fwd_x above spends its time ISSUING, not multiplying: per 32-key tile and wave 21 MFMAs (672 matrix-pipe cycles) stand beside
// ~330 other instructions (16 two-byte LDS stores per staged V chunk for the transposed image, 6 + 5 VALU per score for softmax and the
// hi / lo split of P, two barriers).  This kernel keeps the arithmetic -- S^T = K Q^T in two accumulator sets, online softmax in
// registers, O^T += V^T P^T with the denominator from a ones column -- and removes instructions:
//   * V stays ROW-major in LDS (hi plane | lo plane, 192-byte rows: conflict-free for the transposing read) and the A fragments of the
//     PV products come from ds_read_b64_tr_b16 (gfx950's 4 x 4 transpose read): staging is two 16-byte stores per chunk;
//   * 64 keys per tile and TWO LDS buffers: one barrier per 64 keys instead of two per 32;
//   * the query is scaled by scale * log2(e) once, when it is loaded (re-split: the product's error grows from 3 to 4 units of 2^-22),
//     and the score accumulators START at minus the running maximum, so that after the join fma the register already holds the
//     exponent: no scale, no subtract; the maximum is looked at again only when some score exceeds it (rare after the first tiles);
//   * P is split with v_cvt_pkrtz (two scores per instruction; the lo half absorbs the truncation exactly).
// ---------------------------------------------------------------------------------------------------------------------
typedef __fp16 fh4 __attribute__((__vector_size__(8)));
typedef __fp16 fh2 __attribute__((ext_vector_type(2)));
typedef HT h2 __attribute__((ext_vector_type(2)));

template <bool V> struct BoolK { static constexpr bool value = V; };
template <int D, int KT, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flash_fwd_x2(AttnXP p) {
    constexpr int NKS = (D + 15) / 16, DKP = NKS * 16, LDK = DKP + 8;      // K rows: 16 B x odd -> conflict-free ds_read_b128
    constexpr int NVB = (D + 1 + 31) / 32, LDV = 96;                      // V rows: 192 B -> the four rows x two 16-column halves of a tr read cover 64 banks
    constexpr int KB = KT / 32, DV = 32 * NVB;
    constexpr int PLK = KT * LDK, PLV = KT * LDV, BUF = 2 * PLK + 2 * PLV;   // halves: K hi | K lo | V hi | V lo
    static_assert(DV <= LDV && D % 8 == 0 && KT % 32 == 0, "geometry");
    static_assert((LDK / 8) % 2 == 1, "K row stride");
    constexpr int ROWB = DV * 4 + 16, OUT_BYTES = 4 * 32 * ROWB;
    static_assert(OUT_BYTES <= 2 * BUF * 2, "output stage fits in the tile buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    HT* const sbase = reinterpret_cast<HT*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ql = lane & 31;
    const int img = blockIdx.y / p.H, head = blockIdx.y % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const dwg_xs* Q = p.Q + img * p.bq + (long long)head * D;
    const dwg_xs* K = p.K + img * p.bk + (long long)head * D;
    const dwg_xs* V = p.V + img * p.bv + (long long)head * D;
    dwg_xs* O = p.O + img * p.bo + (long long)head * D;

    // LDS: zeros everywhere (pad columns are multiplied), then the ones column of both V hi planes -- and, where the head size leaves a pad
    // column in the K rows (D = 40 in 48), a ones column there too: the query's element D then carries minus the running maximum INTO the
    // MFMA sum (kept exactly representable in fp16, so that the product is exact)
    constexpr bool KBIAS = DKP > D;
    for (int i = tid; i < 2 * BUF / 8; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < 2 * KT; i += 256) {
        sbase[(i / KT) * BUF + 2 * PLK + (i % KT) * LDV + D] = (HT)1.f;
        if constexpr (KBIAS) sbase[(i / KT) * BUF + (i % KT) * LDK + D] = (HT)1.f;
    }

    // this lane's query row, scaled into the exponent's domain and re-split: B fragments of the S^T products
    h8 qh[NKS], qlo[NKS];
    {
        const int q = q0 + ql;
#pragma unroll
        for (int s = 0; s < NKS; s++) {
            const int c = 16 * s + 8 * half;
            dwg_x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) { v.hi[e] = (HT)0.f; v.lo[e] = (HT)0.f; }
            if (q < p.Nq && c < D) {
                v = dwg_x8::load(Q + (long long)q * p.ldq + c);
#pragma unroll
                for (int e = 0; e < 8; e++) v.set(e, v.get(e) * p.scale_log2);
            }
            qh[s] = v.hi; qlo[s] = v.lo;
        }
    }
    f32x16 acc[NVB], acx[NVB];
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[j][r] = 0.f; acx[j][r] = 0.f; }
    float m_run = -3.0e38f;                 // running maximum (exponent domain), rounded UP to an fp16 value once there is one
    float base = 0.f;                       // what the scores are taken relative to: m_run, or 0 before the first tile

    const int ntiles_all = (p.Nk + KT - 1) / KT;
    const int tper = p.nsplit > 1 ? (ntiles_all + p.nsplit - 1) / p.nsplit : ntiles_all;
    const int tbeg = p.nsplit > 1 ? (int)blockIdx.z * tper : 0;
    const int ntiles = min(ntiles_all, tbeg + tper);

    // staging: chunk c = tid + 256 i of a tile <-> (key, 8-channel group); 32 bytes from global, 16 + 16 into the planes.  Keys past the end
    // are read from the last key's row (finite values; their scores are masked) and chunks past the tile re-read chunk 0 and are not stored.
    constexpr int G8 = D / 8, NCH = KT * G8, NC = (NCH + 255) / 256;
    dwg_x8 kreg[NC], vreg[NC];
    int ckey[NC], ldsk[NC], ldsv[NC];
    const dwg_xs* kptr[NC]; const dwg_xs* vptr[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const int c = tid + i * 256, cc = c < NCH ? c : 0;
        ckey[i] = cc / G8;
        const int off = (cc % G8) * 8;
        ldsk[i] = c < NCH ? ckey[i] * LDK + off : -1;
        ldsv[i] = ckey[i] * LDV + off;
        kptr[i] = K + off; vptr[i] = V + off;
    }
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const long long key = min(k0 + ckey[i], p.Nk - 1);
            kreg[i] = dwg_x8::load(kptr[i] + key * p.ldk);
            vreg[i] = dwg_x8::load(vptr[i] + key * p.ldv);
        }
    };
    // transposing-read address of this lane inside a (16-key step, 32-column block): 16-lane group g = (half, 16-column half), lane i of
    // the group supplies row 4 half + (i >> 2), columns 4 (i & 3) .. + 3; it receives column (i) of rows 4 half .. 4 half + 3
    const int trow = 4 * half + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int toff = trow * LDV + tcol;

    auto tile = [&](int t, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        HT* const sb = sbase + ((t - tbeg) & 1) * BUF;
        HT* const sKh = sb; HT* const sKl = sb + PLK; HT* const sVh = sb + 2 * PLK; HT* const sVl = sVh + PLV;
#pragma unroll
        for (int i = 0; i < NC; i++)
            if (ldsk[i] >= 0) {
                *reinterpret_cast<h8*>(sKh + ldsk[i]) = kreg[i].hi;
                *reinterpret_cast<h8*>(sKl + ldsk[i]) = kreg[i].lo;
                *reinterpret_cast<h8*>(sVh + ldsv[i]) = vreg[i].hi;
                *reinterpret_cast<h8*>(sVl + ldsv[i]) = vreg[i].lo;
            }
        __syncthreads();          // tile t is visible; every wave has left tile t - 1, whose buffer the next iteration overwrites
        if (!TAIL) fetch((t + 1) * KT);
        // S^T (keys x queries) relative to `base`: main and cross sets
        f32x16 s[KB], sx[KB];
        h8 kfh[KB][NKS], kfl[KB][NKS];
        // every K fragment of the tile is requested before the first product (the compiler otherwise sinks each read to its use and the
        // wave sits out the LDS latency a dozen times per tile), then the first 16-key chunk's V fragments
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                kfh[kb][ks] = *reinterpret_cast<const h8*>(&sKh[(32 * kb + ql) * LDK + 16 * ks + 8 * half]);
                kfl[kb][ks] = *reinterpret_cast<const h8*>(&sKl[(32 * kb + ql) * LDK + 16 * ks + 8 * half]);
            }
        h8 vfh[2][NVB], vfl[2][NVB];
        auto vload = [&](int c, int slot) {
#pragma unroll
            for (int j = 0; j < NVB; j++) {
                const int o = 16 * c * LDV + 32 * j + toff;
                typedef __attribute__((address_space(3))) fh4 lds_fh4;
                const fh4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(sVh + o));
                const fh4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(sVh + o + 8 * LDV));
                const fh4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(sVl + o));
                const fh4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(sVl + o + 8 * LDV));
                vfh[slot][j] = __builtin_bit_cast(h8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                vfl[slot][j] = __builtin_bit_cast(h8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * KB * NKS, 0);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) { s[kb][r] = 0.f; sx[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                s[kb] = MFMA16(kfh[kb][ks], qh[ks], s[kb]);
                sx[kb] = MFMA16(kfl[kb][ks], qh[ks], sx[kb]);
                sx[kb] = MFMA16(kfh[kb][ks], qlo[ks], sx[kb]);
            }
        }
        vload(0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s[kb][r] = fmaf(sx[kb][r], DWG_X_LO_INV, s[kb][r]);
                if constexpr (!KBIAS) s[kb][r] -= base;
            }
        if constexpr (TAIL) {
#pragma unroll
            for (int kb = 0; kb < KB; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (t * KT + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half >= p.Nk) s[kb][r] = -3.0e38f;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // the scores are relative to `base`; a new maximum only when one of them is positive (or there was none yet)
        if (__any(mx > 0.f || m_run < -1.0e37f)) {
            const bool up = mx > 0.f || m_run < -1.0e37f;
            float m_new = m_run;
            if (up) {
                const float m = fminf(fmaxf(base + mx, -60000.f), 60000.f);
                m_new = (float)(HT)(m + fabsf(m) * 0.001f + 0.001f);                 // >= m, exactly an fp16 value
            }
            const float shift = m_new - base;                                       // 0 for lanes that keep their maximum
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);              // 0 before the first tile (acc is 0 there)
#pragma unroll
            for (int kb = 0; kb < KB; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++) s[kb][r] -= shift;
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) { acc[j][r] *= alpha; acx[j][r] *= alpha; }
            m_run = m_new; base = m_new;
            if constexpr (KBIAS) { if (half == (D % 16) / 8) qh[D / 16][D % 8] = (HT)(-base); }
        }
        // P^T = 2^s as B operand, split into its planes, in chunks of 16 keys (registers 8 st .. 8 st + 7 of block kb): the chunk's six PV
        // products are independent of the next chunk's exponentials and splits, and the schedule below asks for them side by side
        h8 ph[KB * 2], pl[KB * 2];
        auto pchunk = [&](int c) {
            const int kb = c >> 1, st = c & 1;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s[kb][8 * st + e]), p1 = __builtin_amdgcn_exp2f(s[kb][8 * st + e + 1]);
                const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
                ph[c][e] = hp[0]; ph[c][e + 1] = hp[1];
                pl[c][e] = (HT)((p0 - (float)hp[0]) * DWG_X_LO_SCALE);
                pl[c][e + 1] = (HT)((p1 - (float)hp[1]) * DWG_X_LO_SCALE);
            }
        };
        pchunk(0);
#pragma unroll
        for (int c = 0; c < KB * 2; c++) {
            if (c + 1 < KB * 2) { vload(c + 1, (c + 1) & 1); pchunk(c + 1); }
#pragma unroll
            for (int j = 0; j < NVB; j++) {
                acc[j] = MFMA16(vfh[c & 1][j], ph[c], acc[j]);
                acx[j] = MFMA16(vfl[c & 1][j], ph[c], acx[j]);
                acx[j] = MFMA16(vfh[c & 1][j], pl[c], acx[j]);
            }
            if (c + 1 < KB * 2) {
                // the next chunk's 4 NVB transposing reads first, then 3 NVB MFMAs beside its ~8 exponentials + ~28 VALU
                constexpr int NM = 3 * NVB, VPM = (30 + NM - 1) / NM, TPM = (8 + NM - 1) / NM;
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * NVB, 0);
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, TPM, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
        }
    };
    if (tbeg < ntiles) {
        fetch(tbeg * KT);
        // the last tile of the sequence (ragged key counts: Nk % KT != 0) masks its scores; every tile before it is full
        for (int t = tbeg; t < ntiles - 1; t++) tile(t, BoolK<false>{});
        tile(ntiles - 1, BoolK<true>{});
    }
    // epilogue (as k_flash_fwd_x): join the sets; the denominator is accumulator row D
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = fmaf(acx[j][r], DWG_X_LO_INV, acc[j][r]);
    constexpr int W = D % 32, R = (W & 3) + 4 * (W >> 3), HL = (W >> 2) & 1;
    const float l_run = __shfl(acc[D / 32][R], ql + 32 * HL);
    if (p.nsplit > 1) {
        const int q = q0 + ql;
        if (q < p.Nq) {
            float* row = p.ws + (((long long)blockIdx.z * gridDim.y + blockIdx.y) * p.Nq + q) * p.dvw;
#pragma unroll
            for (int j = 0; j < NVB; j++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
                    *reinterpret_cast<float4*>(row + 32 * j + 8 * r4 + 4 * half) =
                        make_float4(acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]);
            if (half == 0) { row[DV] = m_run; row[DV + 1] = l_run; }
        }
        return;
    }
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    __syncthreads();                                   // the tiles are dead: their LDS becomes the output stage
    unsigned char* myO = smem + wave * 32 * ROWB;
#pragma unroll
    for (int j = 0; j < NVB; j++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            h4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; e++) { HT a, b; dwg_x_split(acc[j][4 * r4 + e] * inv, a, b); oh[e] = a; ol[e] = b; }
            unsigned char* g = myO + ql * ROWB + (4 * j + r4) * 32 + half * 8;
            *reinterpret_cast<h4*>(g) = oh; *reinterpret_cast<h4*>(g + 16) = ol;
        }
    __syncthreads();
    constexpr int pieces = D / 4;                      // 16-byte pieces per row (two per 8-group)
    for (int c = lane; c < 32 * pieces; c += 64) {
        const int q = c / pieces, pc = c % pieces;
        if (q0 + q < p.Nq)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(O + (long long)(q0 + q) * p.ldo) + pc * 16) =
                *reinterpret_cast<const uint4*>(myO + q * ROWB + pc * 16);
    }
}

template <int D, int KT, int MINB>
static void launch_flash_x2(const AttnXP& p, dim3 grid, hipStream_t stream, const char* name, const char* sym, double flops) {
    constexpr int NKS = (D + 15) / 16, LDK = NKS * 16 + 8, LDV = 96;
    constexpr size_t lds = (size_t)2 * (2 * KT * LDK + 2 * KT * LDV) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_fwd_x2<D, KT, MINB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    DWG_LAUNCH_W(name, sym, flops, (k_flash_fwd_x2<D, KT, MINB>), grid, dim3(256), lds, stream, p);
}

// Combines the key ranges of a split attention launch, in range order: O = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s  with  m = max_s m_s.
// One thread per (query row, four channels); the output goes out in the f32x planes.
__global__ __launch_bounds__(256) void k_flash_merge_x(AttnXP p, int BH, int DV) {
    const int groups = p.d >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = idx / groups;
    const int g = (int)(idx - row * groups);
    if (row >= (long long)BH * p.Nq) return;
    const int bh = (int)(row / p.Nq), q = (int)(row - (long long)bh * p.Nq);
    const long long sstride = (long long)BH * p.Nq * p.dvw;
    const float* base = p.ws + row * p.dvw;
    float m = -3.0e38f;
    for (int s2 = 0; s2 < p.nsplit; s2++) m = fmaxf(m, base[s2 * sstride + DV]);
    float L = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < p.nsplit; s2++) {
        const float* r = base + s2 * sstride;
        const float w = __builtin_amdgcn_exp2f(r[DV] - m);
        const float4 v = *reinterpret_cast<const float4*>(r + 4 * g);
        L = fmaf(w, r[DV + 1], L);
        o[0] = fmaf(w, v.x, o[0]); o[1] = fmaf(w, v.y, o[1]); o[2] = fmaf(w, v.z, o[2]); o[3] = fmaf(w, v.w, o[3]);
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    const float out[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
    const int img = bh / p.H, head = bh % p.H;
    dwg_xs* O = p.O + img * p.bo + (long long)head * p.d;
    dwg_x_put4(O + (long long)q * p.ldo, 4 * g, out);
}

// key ranges for a launch: none while the query blocks alone fill the chip or there are few key tiles; else enough ranges for ~512 workgroups,
// at least two 32-key tiles per range, at most eight
static int attn_kt(int d) {               // keys per tile of the kernel that serves head size d
    static const int v2 = getenv("DWG_ATTN_V2") ? atoi(getenv("DWG_ATTN_V2")) : 1;
    return (v2 && (d == 40 || d == 80)) ? 64 : 32;
}
static int attn_splits(int B, int H, int Nq, int Nk, int kt) {
    static const int off = getenv("DWG_ATTN_SPLIT") ? atoi(getenv("DWG_ATTN_SPLIT")) : -1;      // 0 / 1: never; n > 1: force n ranges
    const int base = dwg_cdiv(Nq, 128) * B * H, ntiles = dwg_cdiv(Nk, kt);
    if (off == 0 || off == 1) return 1;
    int sp = off > 1 ? off : (base >= 256 ? 1 : dwg_cdiv(512, base));
    if (sp > 8) sp = 8;
    if (sp > ntiles / 2) sp = ntiles / 2;
    return sp < 2 ? 1 : sp;
}
static int attn_dv(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : (d <= 96 ? 96 : 160)); }

}  // namespace

extern "C" {

size_t dwg_attention_split_workspace_bytes_x(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || d > 160) return 0;
    const int sp = attn_splits(B, H, Nq, Nk, attn_kt(d));
    return sp > 1 ? (size_t)sp * B * H * Nq * (attn_dv(d) + 4) * sizeof(float) : 0;
}

int dwg_attention_forward_x_ws(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq, const void* K,
                               int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo, int64_t bo, float scale,
                               void* workspace, size_t workspace_bytes, dwg_stream_t stream_);

int dwg_attention_forward_x(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq, const void* K,
                            int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo, int64_t bo, float scale,
                            dwg_stream_t stream_) {
    return dwg_attention_forward_x_ws(B, H, Nq, Nk, d, Q, ldq, bq, K, ldk, bk, V, ldv, bv, O, ldo, bo, scale, nullptr, 0, stream_);
}

// `workspace` (dwg_attention_split_workspace_bytes_x bytes, or NULL): with it, launches whose query blocks do not fill the chip split the
// keys over workgroups and a second small launch merges the ranges (fixed order: run-to-run reproducible; the sums differ from the unsplit
// launch's in rounding order only)
int dwg_attention_forward_x_ws(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d, const void* Q, int64_t ldq, int64_t bq, const void* K,
                               int64_t ldk, int64_t bk, const void* V, int64_t ldv, int64_t bv, void* O, int64_t ldo, int64_t bo, float scale,
                               void* workspace, size_t workspace_bytes, dwg_stream_t stream_) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || d % 8 || d > 160 || !Q || !K || !V || !O) return DWG_E_ARG;
    if ((ldq | ldk | ldv | ldo | bq | bk | bv | bo) % 8) return DWG_E_ARG;   // whole 8-channel groups
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) % 16) return DWG_E_ARG;
    AttnXP p{(const dwg_xs*)Q, (const dwg_xs*)K, (const dwg_xs*)V, (dwg_xs*)O, Nq, Nk, H, d, ldq, ldk, ldv, ldo, bq, bk, bv, bo,
             scale * 1.4426950408889634f, 1, nullptr, 0};
    int sp = workspace ? attn_splits(B, H, Nq, Nk, attn_kt(d)) : 1;
    const int DVp = attn_dv(d);
    if (sp > 1 && (((uintptr_t)workspace & 15) || (size_t)sp * B * H * Nq * (DVp + 4) * sizeof(float) > workspace_bytes)) sp = 1;
    if (sp > 1) { p.nsplit = sp; p.ws = reinterpret_cast<float*>(workspace); p.dvw = DVp + 4; }
    dim3 grid(dwg_cdiv(Nq, 128), B * H, sp), block(256);
    hipStream_t stream = (hipStream_t)stream_;
    const double flops = 4.0 * B * H * (double)Nq * Nk * d;     // QK^T and PV on the logical head size, one multiply-add per product
    static const int v2 = getenv("DWG_ATTN_V2") ? atoi(getenv("DWG_ATTN_V2")) : 1;       // 0: the round-4 kernels at every head size
    if (v2 && d == 40) launch_flash_x2<40, 64, 2>(p, grid, stream, "flash_attn_d40", "k_flash_fwd_x2<40, 64, 2>", flops);
    else if (v2 && d == 80) launch_flash_x2<80, 64, 1>(p, grid, stream, "flash_attn_d80", "k_flash_fwd_x2<80, 64, 1>", flops);
    else if (d <= 32) DWG_LAUNCH_W("flash_attn_d32", "k_flash_fwd_x<32, 32, 0, 2>", flops, (k_flash_fwd_x<32, 32, 0, 2>), grid, block, 0, stream, p);
    else if (d == 40) DWG_LAUNCH_W("flash_attn_d48", "k_flash_fwd_x<48, 64, 40, 2>", flops, (k_flash_fwd_x<48, 64, 40, 2>), grid, block, 0, stream, p);
    else if (d <= 48) DWG_LAUNCH_W("flash_attn_d48", "k_flash_fwd_x<48, 64, 0, 2>", flops, (k_flash_fwd_x<48, 64, 0, 2>), grid, block, 0, stream, p);
    else if (d <= 64) DWG_LAUNCH_W("flash_attn_d64", "k_flash_fwd_x<64, 64, 0, 2>", flops, (k_flash_fwd_x<64, 64, 0, 2>), grid, block, 0, stream, p);
    else if (d == 80) DWG_LAUNCH_W("flash_attn_d96", "k_flash_fwd_x<96, 96, 80, 1>", flops, (k_flash_fwd_x<96, 96, 80, 1>), grid, block, 0, stream, p);
    else if (d <= 96) DWG_LAUNCH_W("flash_attn_d96", "k_flash_fwd_x<96, 96, 0, 1>", flops, (k_flash_fwd_x<96, 96, 0, 1>), grid, block, 0, stream, p);
    else DWG_LAUNCH_W("flash_attn_d160", "k_flash_fwd_x<160, 160, 0, 1>", flops, (k_flash_fwd_x<160, 160, 0, 1>), grid, block, 0, stream, p);
    DWG_RETURN_IF_LAUNCH_FAILED();
    if (sp > 1) {
        const long long n = (long long)B * H * Nq * (d / 4);
        DWG_LAUNCH("flash_attn_merge", k_flash_merge_x, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, B * H, DVp);
        DWG_RETURN_IF_LAUNCH_FAILED();
    }
    return DWG_OK;
}

}  // extern "C"
