// gemm.hip -- MFMA GEMM / implicit-GEMM convolution for gfx950 (wave64, v_mfma_f32_32x32x16_bf16 and the exact-f32
// v_mfma_f32_32x32x2_f32), fp32 accumulation, fused epilogue (scale, bias, activation, residual, dtype cast).
//
// One kernel template serves every contraction on the hot path:
//   * the per-Gaussian MLPs (f32-exact): nerf_model.py:12-33, deform_model.py:102-143 -- fwd, dgrad, wgrad (split-K atomics)
//   * UNet / ControlNet / VAE linear layers, attention QK^T and PV (batched, strided per head)        (bf16 in, f32 acc)
//   * conv3x3 / conv1x1 forward and input-gradient as implicit GEMM over NHWC activations             (bf16 in, f32 acc)
//
// C[m][n] = epi( alpha * sum_k A(m,k) * B(n,k) ).  Operand element addresses are fully strided:
//   A(m,k) = A + m*sam + k*sak,  B(n,k) = B + n*sbn + k*sbk   (so NT / TN / NN / TT all map onto one kernel)
// or, for the conv loader, A(m,k) is the im2col view of an NHWC tensor: m=(img,oy,ox), k=(ky,kx,ci), with stride,
// asymmetric padding and input dilation (the latter turns the same loader into the transposed-conv / dgrad gather).
// Tiling: 128 x {128,64} x BK workgroup tile, 4 waves, 32x32 MFMA tiles, operands staged global->VGPR->LDS with the
// next tile's global loads issued before the current tile's MFMAs (one barrier per k-step, two LDS buffers).
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_gemm.h"
#include <cstdlib>

// The 16-bit operand type of this translation unit.  gemm.hip itself is the bf16 (+ exact-f32) unit; gemm_f16.hip re-includes it with
// DWG_GEMM_F16_TU defined: the same kernels on _Float16 operands (v_mfma_f32_32x32x16_f16) for the fp16-storage plans -- the reference's
// autocast storage type (configs/__init__.py:462).  Every kernel lives in an anonymous namespace, so the units do not clash.
// gemm_x.hip re-includes it a third time with DWG_GEMM_X_TU: the split-precision unit (DWG_DTYPE_F32X, dwg_xfmt.h).  There HT is the PHYSICAL
// fp16 half of the hi / lo planes: the loaders see "a 2-byte tensor with 2 K columns" (the entry point doubles K, Cin and every operand stride),
// the k-loops form each product from three MFMAs into two accumulator sets, and the epilogue splits what it stores.
#if defined(DWG_GEMM_X_TU)
#include "dwg_xfmt.h"
typedef _Float16 HT;
#define DWG_DTYPE_HALF DWG_DTYPE_F32X
#define DWG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DWG_HALF_NAME "f32x"
#elif defined(DWG_GEMM_F16_TU)
typedef _Float16 HT;
#define DWG_DTYPE_HALF DWG_DTYPE_F16
#define DWG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DWG_HALF_NAME "f16"
#else
typedef __bf16 HT;
#define DWG_DTYPE_HALF DWG_DTYPE_BF16
#define DWG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DWG_HALF_NAME "bf16"
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) HT bf16x8;      // eight 16-bit operands (the name predates the fp16 unit)
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvP {
    int enabled, Cin, Hin, Win, Hout, Wout, KH, KW, stride, pad_t, pad_l, dil;  // dil = input dilation (dgrad of strided conv)
    int up;          // nearest-neighbour input upsampling factor (1 | 2): Upsample2D fused into the following conv
    int cin1;        // channels [0,cin1) come from A, [cin1,Cin) from A2 (skip-connection concat fused into the conv)
    const void* A2;
};

struct GemmP {
    const void* A; const void* B; void* C; const void* bias; const void* residual;
    int M, N, K;
    long long sam, sak, sbn, sbk, ldc, ldr;
    int nb2;
    long long bA1, bA2, bB1, bB2, bC1, bC2, bR1, bR2;
    int act; float alpha;
    int out_bf16, res_bf16, bias_per_row, splitk, accumulate;
    float* ws;          // split-K slab workspace [splitk][M][N] (nullptr: atomicAdd into C)
    int* cnt;           // per-tile arrival counters (workspace header, zero between launches): the LAST slice of a tile sums the slabs and
                        // stores C inside the GEMM kernel -- no k_splitk_epilogue launch (nullptr: the separate reduce launch)
    int dbg;            // DWG_GEMM_DEBUG (timing experiments, results are garbage): 1 = k_gemm_glds skips LDS reads + MFMAs, 2 = skips the tile loads
    int bias_row_div;   // > 0: bias index = (row / bias_row_div) * bias_ld + col  (per-image channel bias: conv bias + time embedding)
    long long bias_ld;
    ConvP conv;
};

__device__ __forceinline__ float bf2f(HT x) { return (float)x; }
__device__ __forceinline__ HT f2bf(float x) { return (HT)x; }

// Element access to a C / residual tensor of this unit's 16-bit storage type (index in LOGICAL elements).  The split-precision unit stores
// hi / lo halves per 8-channel group (dwg_xfmt.h); the bf16 / fp16 units one 2-byte element.
struct bf16x4_t { HT v[4]; };
#ifdef DWG_GEMM_X_TU
__device__ __forceinline__ float ld_half1(const void* base, long long i) { return dwg_x_get1(base, i); }
__device__ __forceinline__ void st_half1(void* base, long long i, float v) { dwg_x_put1(base, i, v); }
__device__ __forceinline__ void ld_half4(const void* base, long long i, float (&v)[4]) { dwg_x_get4(base, i, v); }
__device__ __forceinline__ void st_half4(void* base, long long i, const float (&v)[4]) { dwg_x_put4(base, i, v); }
#else
__device__ __forceinline__ float ld_half1(const void* base, long long i) { return bf2f(reinterpret_cast<const HT*>(base)[i]); }
__device__ __forceinline__ void st_half1(void* base, long long i, float v) { reinterpret_cast<HT*>(base)[i] = f2bf(v); }
__device__ __forceinline__ void ld_half4(const void* base, long long i, float (&v)[4]) {
    const bf16x4_t r = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const HT*>(base) + i);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = bf2f(r.v[e]);
}
__device__ __forceinline__ void st_half4(void* base, long long i, const float (&v)[4]) {
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; e++) o.v[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x4_t*>(reinterpret_cast<HT*>(base) + i) = o;
}
#endif

// LIGHT: only identity / SiLU are compiled in (the LDS-patch convolution's epilogue: the erf of GELU would cost it registers it
// does not have; the dispatcher sends other activations down the generic kernels)
template <bool LIGHT = false>
__device__ __forceinline__ float apply_act(float v, int act) {
    if constexpr (LIGHT) return act == 3 ? v / (1.f + __expf(-v)) : v;
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: return v > 0.f ? v : 0.01f * v;
        case 3: return v / (1.f + __expf(-v));
        case 4: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));    // libm erf here: the branch-free dwg_erf_fast in this switch makes
                                                                              // LLVM unswitch the epilogue loops and demote the accumulators to scratch
        case 5: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

template <typename T> struct TT;
template <> struct TT<HT> { static constexpr int VEC = 8; static constexpr int BK = 64; static constexpr int PAD = 8; };
template <> struct TT<float> { static constexpr int VEC = 4; static constexpr int BK = 16; static constexpr int PAD = 4; };

// operand load modes
enum { MODE_KVEC = 0, MODE_RVEC = 1, MODE_SCALAR = 2, MODE_CONV = 3 };

template <typename T, int VEC> struct Chunk { T v[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ Chunk<T, VEC> load_vec(const T* p) {
    Chunk<T, VEC> c;
    *reinterpret_cast<uint4*>(&c) = *reinterpret_cast<const uint4*>(p);
    return c;
}
template <typename T, int VEC>
__device__ __forceinline__ Chunk<T, VEC> zero_chunk() {
    Chunk<T, VEC> c;
    *reinterpret_cast<uint4*>(&c) = make_uint4(0u, 0u, 0u, 0u);
    return c;
}

// Loads this thread's share of a [ROWS][BK] operand tile into registers, then stores it to LDS as [ROWS][BK + PAD].
// In the K-contiguous modes every thread owns ONE 16-byte k-column of the tile and NCH rows of it, so everything that
// depends on the row (im2col pixel decomposition, row pointers) is computed once in init() and the k-dependent state
// ((ky, kx, ci) of the im2col view) advances incrementally -- no integer division inside the k-loop.
template <typename T, int ROWS, int MODE>
struct TileLoader {
    static constexpr int VEC = TT<T>::VEC, BK = TT<T>::BK, LDT = BK + TT<T>::PAD;
    static constexpr int CPR = BK / VEC, RPI = 256 / CPR;          // chunks per row, rows per pass (K-contiguous modes)
    static constexpr int CPK = ROWS / VEC, KPI = 256 / CPK;        // chunks per k-row, k-rows per pass (row-contiguous mode)
    static constexpr int NCH = (MODE == MODE_SCALAR) ? (ROWS * BK / 256) : (MODE == MODE_RVEC ? BK / KPI : ROWS / RPI);
    Chunk<T, VEC> regs[(MODE == MODE_SCALAR) ? 1 : NCH];
    T sregs[(MODE == MODE_SCALAR) ? NCH : 1];
    const T* rowptr[(MODE == MODE_KVEC) ? NCH : 1];
    int iy0[(MODE == MODE_CONV) ? NCH : 1], ix0[(MODE == MODE_CONV) ? NCH : 1];
    long long pix0[(MODE == MODE_CONV) ? NCH : 1];
    bool rok[(MODE == MODE_CONV) ? NCH : 1];
    const T* base; long long srow, sk; int nrows, r0, kcur, kend;
    int ci, ky, kx;

    __device__ __forceinline__ void init(const T* base_, long long srow_, long long sk_, int nrows_, int r0_, int kbeg, int kend_,
                                         const ConvP& cv) {
        base = base_; srow = srow_; sk = sk_; nrows = nrows_; r0 = r0_; kend = kend_;
        const int tid = threadIdx.x;
        if (MODE == MODE_KVEC) {
            kcur = kbeg + (tid % CPR) * VEC;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int r = r0 + tid / CPR + i * RPI;
                rowptr[i] = r < nrows ? base + (long long)r * srow : nullptr;
            }
        } else if (MODE == MODE_CONV) {
            kcur = kbeg + (tid % CPR) * VEC;
            const int hw = cv.Hout * cv.Wout;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int r = r0 + tid / CPR + i * RPI;
                rok[i] = r < nrows;
                int rr = rok[i] ? r : 0;
                int img = rr / hw, rem = rr - img * hw;
                int oy = rem / cv.Wout, ox = rem - oy * cv.Wout;
                iy0[i] = oy * cv.stride - cv.pad_t; ix0[i] = ox * cv.stride - cv.pad_l;
                pix0[i] = (long long)img * cv.Hin * cv.Win;
            }
            int tap = kcur / cv.Cin;
            ci = kcur - tap * cv.Cin; ky = tap / cv.KW; kx = tap - ky * cv.KW;
        } else {
            kcur = kbeg;
        }
    }

    __device__ __forceinline__ void advance(const ConvP& cv) {
        kcur += BK;
        if (MODE == MODE_CONV) {
            ci += BK;
            while (ci >= cv.Cin) { ci -= cv.Cin; if (++kx == cv.KW) { kx = 0; ++ky; } }
        }
    }

    __device__ __forceinline__ void load(const ConvP& cv) {
        const int tid = threadIdx.x;
        if (MODE == MODE_KVEC) {
            const bool kok = kcur < kend;
#pragma unroll
            for (int i = 0; i < NCH; i++)
                regs[i] = (kok && rowptr[i]) ? load_vec<T, VEC>(rowptr[i] + kcur) : zero_chunk<T, VEC>();
        } else if (MODE == MODE_CONV) {
            const bool kok = kcur < kend;
            const T* src0 = ci < cv.cin1 ? base : reinterpret_cast<const T*>(cv.A2);
            const int cs = ci < cv.cin1 ? cv.cin1 : (cv.Cin - cv.cin1);
            const int co = ci < cv.cin1 ? ci : ci - cv.cin1;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int iy = iy0[i] + ky, ix = ix0[i] + kx;
                bool ok = kok && rok[i] && iy >= 0 && ix >= 0;
                if (cv.dil > 1) { ok = ok && (iy % cv.dil == 0) && (ix % cv.dil == 0); iy /= cv.dil; ix /= cv.dil; }
                if (cv.up > 1) { iy >>= 1; ix >>= 1; }
                ok = ok && iy < cv.Hin && ix < cv.Win;
                regs[i] = ok ? load_vec<T, VEC>(src0 + (pix0[i] + (long long)iy * cv.Win + ix) * cs + co) : zero_chunk<T, VEC>();
            }
        } else if (MODE == MODE_RVEC) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int kk = tid / CPK + i * KPI, rc = tid % CPK;
                int r = r0 + rc * VEC, k = kcur + kk;
                regs[i] = (r < nrows && k < kend) ? load_vec<T, VEC>(base + (long long)k * sk + r) : zero_chunk<T, VEC>();
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int c = tid + i * 256, row = c / BK, kk = c - row * BK;
                int r = r0 + row, k = kcur + kk;
                sregs[i] = (r < nrows && k < kend) ? base[(long long)r * srow + (long long)k * sk] : (T)0.f;
            }
        }
    }

    __device__ __forceinline__ void store(T* lds) {
        const int tid = threadIdx.x;
        if (MODE == MODE_KVEC || MODE == MODE_CONV) {
#pragma unroll
            for (int i = 0; i < NCH; i++)
                *reinterpret_cast<uint4*>(lds + (tid / CPR + i * RPI) * LDT + (tid % CPR) * VEC) = *reinterpret_cast<uint4*>(&regs[i]);
        } else if (MODE == MODE_RVEC) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int kk = tid / CPK + i * KPI, rc = tid % CPK;
#pragma unroll
                for (int e = 0; e < VEC; e++) lds[(rc * VEC + e) * LDT + kk] = regs[i].v[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                int c = tid + i * 256, row = c / BK, kk = c - row * BK;
                lds[row * LDT + kk] = sregs[i];
            }
        }
    }
};

template <typename T>
__device__ __forceinline__ void mma_tile(const T* sa, const T* sb, int lane, f32x16& acc);

// one 32x32 output tile x one BK slab
template <>
__device__ __forceinline__ void mma_tile<HT>(const HT* sa, const HT* sb, int lane, f32x16& acc) {
    constexpr int LDT = TT<HT>::BK + TT<HT>::PAD;
    const HT* pa = sa + (lane & 31) * LDT + (lane >> 5) * 8;
    const HT* pb = sb + (lane & 31) * LDT + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < TT<HT>::BK / 16; ks++) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
        bf16x8 b = *reinterpret_cast<const bf16x8*>(pb + ks * 16);
        acc = DWG_MFMA16(a, b, acc);
    }
}
template <>
__device__ __forceinline__ void mma_tile<float>(const float* sa, const float* sb, int lane, f32x16& acc) {
    constexpr int LDT = TT<float>::BK + TT<float>::PAD;
    const float* pa = sa + (lane & 31) * LDT + (lane >> 5);
    const float* pb = sb + (lane & 31) * LDT + (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < TT<float>::BK / 2; ks++)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[ks * 2], pb[ks * 2], acc, 0, 0, 0);
}

template <bool LIGHT = false>
__device__ __forceinline__ void epilogue_store(const GemmP& p, float v, int row, int col, long long coff, long long roff) {
    const long long ci = coff + (long long)row * p.ldc + col;
    if (p.bias) v += reinterpret_cast<const float*>(p.bias)[p.bias_row_div > 0 ? (long long)(row / p.bias_row_div) * p.bias_ld + col
                                                                             : (p.bias_per_row ? row : col)];
    v = apply_act<LIGHT>(v, p.act);
    if (p.residual) {
        const long long ri = roff + (long long)row * p.ldr + col;
        v += p.res_bf16 ? ld_half1(p.residual, ri) : reinterpret_cast<const float*>(p.residual)[ri];
    }
    if (p.out_bf16) st_half1(p.C, ci, v);
    else if (p.accumulate) reinterpret_cast<float*>(p.C)[ci] += v;
    else reinterpret_cast<float*>(p.C)[ci] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Transposed accumulator orientation (all MFMA kernels of this file): the MFMA is issued as D' = Btile . Atile^T, so that a lane owns
// ONE output row (lane & 31) and, per group of four accumulator registers, FOUR CONSECUTIVE output columns
// (8*(r>>2) + 4*(lane>>5) + (r&3)).  The epilogue then moves 8-byte bf16x4 / 16-byte float4 pieces (bias, residual, store,
// split-K slab) instead of one 2-byte element per instruction: 4x fewer epilogue instructions and memory requests, which is
// what the small-K layers (5-20 k-steps per tile) spend most of their time on.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool epilogue_vec_ok(const GemmP& p, long long coff, long long roff) {
    bool ok = (p.N & 3) == 0 && (p.ldc & 3) == 0 && (coff & 3) == 0 && ((uintptr_t)p.C & 15) == 0;
    if (p.residual) ok = ok && (p.ldr & 3) == 0 && (roff & 3) == 0 && ((uintptr_t)p.residual & 15) == 0;
    return ok;
}

// four consecutive columns col..col+3 of one row
template <bool LIGHT = false>
__device__ __forceinline__ void epilogue_store4(const GemmP& p, float (&v)[4], int row, int col, long long coff, long long roff, bool vec_ok) {
    if (!vec_ok || col + 3 >= p.N) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (col + e < p.N) epilogue_store<LIGHT>(p, v[e], row, col + e, coff, roff);
        return;
    }
    if (p.bias) {
        const float* b = reinterpret_cast<const float*>(p.bias);
        if (p.bias_row_div > 0) {
            b += (long long)(row / p.bias_row_div) * p.bias_ld + col;
            v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        } else if (p.bias_per_row) {
            const float br = b[row];
            v[0] += br; v[1] += br; v[2] += br; v[3] += br;
        } else {
            v[0] += b[col]; v[1] += b[col + 1]; v[2] += b[col + 2]; v[3] += b[col + 3];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = apply_act<LIGHT>(v[e], p.act);
    if (p.residual) {
        const long long ri = roff + (long long)row * p.ldr + col;
        if (p.res_bf16) {
            float r[4];
            ld_half4(p.residual, ri, r);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += r[e];
        } else {
            float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ri);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
    }
    const long long ci = coff + (long long)row * p.ldc + col;
    if (p.out_bf16) {
        st_half4(p.C, ci, v);
    } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + ci);
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (p.accumulate) { float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        *dst = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Compact epilogue.  The accumulator tile goes registers -> LDS (the operand stages are dead by then) in two 64-row passes, and ONE rolled
// loop -- a single copy of every run-time variant: slab / atomic split-K, GEGLU pair, bias forms, activation, residual, output type --
// takes float4 pieces back out row-major, so that 16 consecutive threads write one contiguous 128- / 256-byte row segment.
// Why: with the variants inside the fully unrolled (tile, register-group) loops every kernel of this file carried 15 000 - 36 000 static
// instructions in ~1 600 - 3 200 basic blocks (120 - 290 KB of code against a 64 KB instruction cache) around a 70 - 115 instruction main
// loop: a ONE-k-step, ONE-workgroup launch took 6.7 us inside a captured graph against 1.5 us for an empty kernel, i.e. ~5 us of
// instruction fetch per launch on ~470 launches per SDS step (tools/shape_sweep.py PROBE=1, DESIGN.md "what bounds the small GEMMs").
// ---------------------------------------------------------------------------------------------------------------------
// row stride 4 (mod 32) banks: conflict-free b128 rows.  One pass over all BM rows when the kernel's operand stages are large enough, else two / four.
template <int BN, int BM = 128> struct EpiLds {
    static constexpr int LDC = BN + 4;
    static constexpr int passes(size_t lds_bytes) {
        return lds_bytes >= (size_t)BM * LDC * 4 ? 1 : (lds_bytes >= (size_t)(BM / 2) * LDC * 4 ? 2 : 4);
    }
    static constexpr size_t bytes(int npass) { return (size_t)(BM / npass) * LDC * 4; }
};

// RowFn: tile-local row (0..BM-1) -> global output row, or -1 (outside the problem).  BM rows, NT threads (128 / 256 for the four-wave
// kernels; 256 x 128 and 128 x 256 tiles on eight waves: round 6)
// 16-byte accesses at DEVICE scope (relaxed agent-scope atomics on the two 8-byte halves: sc1 -- written through / fetched past the
// XCD-private L2): what lets slabs cross XCDs inside one kernel WITHOUT release / acquire fences -- a fence writes back and invalidates
// the whole L2 (measured: the step went from 24 to 42 ms with __threadfence() here).
__device__ __forceinline__ void st_agent4(float* ptr, const float4& v) {
    unsigned long long* u = reinterpret_cast<unsigned long long*>(ptr);
    const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
    const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
    __hip_atomic_store(u, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_agent4(const float4* ptr) {
    const unsigned long long* u = reinterpret_cast<const unsigned long long*>(ptr);
    const unsigned long long lo = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                       __uint_as_float((unsigned)(hi >> 32)));
}

// Split-K without a reduce launch: every slice of a tile has written its slab (above); the workgroup that arrives LAST at the tile's counter
// sums the slabs in slice order and stores C through the same epilogue_store4 as k_splitk_epilogue -- the same additions in the same order,
// so the result does not depend on which slice arrives last (bit-identical to the two-launch path).  The slabs of the other slices were
// written on other XCDs: they are stored and loaded at device scope (st_agent4 / ld_agent4) and every store has completed (vmcnt(0)) before
// the workgroup's arrival is counted.  The counter goes back to zero for the next launch that uses this workspace.
template <int BN, int BM, int NT, typename RowFn>
__device__ __forceinline__ void splitk_reduce_in_kernel(const GemmP& p, float* sC, int n0, int tid, long long coff, long long roff, bool vec_ok,
                                                     RowFn row_of, int tile_id) {
    constexpr int C4 = BN / 4;
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): this thread's slab stores have completed at device scope
    __syncthreads();
    int* flag = reinterpret_cast<int*>(sC);
    if (tid == 0) {
        const int arrived = atomicAdd(p.cnt + tile_id, 1);
        const int last = arrived == p.splitk - 1;
        if (last) atomicExch(p.cnt + tile_id, 0);
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    // PB pieces x four slices in flight per round: one workgroup sums the whole tile, so the round count (memory latencies in a row) is its time
    const long long n4 = ((long long)p.M * p.N) >> 2;
    constexpr int PIECES = BM * C4 / NT, PB = PIECES % 4 == 0 ? 4 : 1;   // (8 x 4 float4 in flight spills the 256-register kernels)
    static_assert(BM * C4 % NT == 0, "whole pieces per thread");
#pragma unroll 1
    for (int k0 = 0; k0 < PIECES; k0 += PB) {
        const float4* src[PB];
        int row[PB], col[PB];
        float4 a[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int idx = tid + (k0 + j) * NT, rl = idx / C4, c4 = idx - rl * C4;
            row[j] = row_of(rl); col[j] = n0 + c4 * 4;
            if (col[j] >= p.N) row[j] = -1;
            src[j] = reinterpret_cast<const float4*>(p.ws + (row[j] < 0 ? 0LL : (long long)row[j] * p.N + col[j]));     // (not stored: any valid piece)
        }
#pragma unroll 1
        for (int s0 = 0; s0 < p.splitk; s0 += 4) {
            float4 u[PB][4];
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (s0 + t < p.splitk) {
#pragma unroll
                    for (int j = 0; j < PB; j++) u[j][t] = ld_agent4(src[j] + (long long)(s0 + t) * n4);
                }
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (s0 + t < p.splitk) {
#pragma unroll
                    for (int j = 0; j < PB; j++) {
                        if (s0 + t == 0) a[j] = u[j][0];
                        else { a[j].x += u[j][t].x; a[j].y += u[j][t].y; a[j].z += u[j][t].z; a[j].w += u[j][t].w; }
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < PB; j++) {
            if (row[j] < 0) continue;
            float v[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
            epilogue_store4(p, v, row[j], col[j], coff, roff, vec_ok);
        }
    }
}

template <int BN, int NPASS, int TM, int TN, bool LIGHT, int BM = 128, int NT = 256, typename RowFn>
__device__ __forceinline__ void tile_epilogue_lds(const GemmP& p, f32x16 (&acc)[TM][TN], float* sC, int n0, int wm, int wn, int lane, int tid,
                                                  int ks_id, long long coff, long long roff, RowFn row_of, int tile_id = 0) {
    constexpr int LDC = EpiLds<BN, BM>::LDC, C4 = BN / 4, PR = BM / NPASS;      // PR rows per pass
    constexpr int BPP = BM / 32 / NPASS;                                         // 32-row blocks per pass
    static_assert(NT % C4 == 0 && (C4 & (C4 - 1)) == 0, "a thread keeps one column group");
    const bool vec_ok = epilogue_vec_ok(p, coff, roff);
    const int lrow = lane & 31, lhalf = (lane >> 5) * 4;
    const bool geglu = p.act == 6;
    const float* bias = reinterpret_cast<const float*>(p.bias);
    const bool fast = vec_ok && !geglu && !p.accumulate && !p.bias_per_row && (p.splitk <= 1 || p.ws) &&
                      (!bias || p.bias_row_div > 0 || ((uintptr_t)bias & 15) == 0) &&
                      (!bias || p.bias_row_div <= 0 || (((uintptr_t)bias & 15) == 0 && (p.bias_ld & 3) == 0)) &&
                      (p.splitk <= 1 || ((uintptr_t)p.ws & 15) == 0);
#pragma unroll 1
    for (int q = 0; q < NPASS; q++) {                   // rows [PR q, PR q + PR) of the tile
        __syncthreads();                                // operand tiles (q = 0) / the previous pass (q = 1) are no longer being read
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int rb = wm * TM + i;                 // 32-row block of this wave (wave-uniform)
            if (NPASS > 1 && rb / BPP != q) continue;
            float* dst = sC + ((NPASS > 1 ? rb % BPP : rb) * 32 + lrow) * LDC + wn * 64 + lhalf;
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
                    *reinterpret_cast<float4*>(dst + j * 32 + 8 * r4) =
                        make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
        }
        __syncthreads();
        if (geglu) {
            // GEGLU pair (diffusers GEGLU: hidden, gate = proj(x).chunk(2); hidden * gelu(gate)).  The projection rows are pre-interleaved in
            // blocks of 32, so columns [64 b, 64 b + 32) of the tile hold `hidden` and [64 b + 32, 64 b + 64) the matching `gate`: the product
            // is formed here and only the half-width result is stored (no [M, 8C] round trip).  N % 64 == 0.
            // a thread keeps ONE group of four output columns for the whole tile: the eight bias values and the column tests are loop invariants
            constexpr int G = C4 / 2, RSTEP = NT / G;
            const int o4 = tid & (G - 1), rl0 = tid / G;
            const int b = o4 >> 3, w = (o4 & 7) * 4;
            const int colp = n0 + b * 64 + w;
            if (colp + 35 < p.N) {
                float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { bh[e] = bias[colp + e]; bg[e] = bias[colp + 32 + e]; }
                }
                const bool vec_store = ((p.ldc | coff) & 3) == 0 && ((uintptr_t)p.C & 7) == 0;
#pragma unroll 2
                for (int rl = rl0; rl < PR; rl += RSTEP) {
                    const int row = row_of(q * PR + rl);
                    if (row < 0) continue;
                    const float4 h = *reinterpret_cast<const float4*>(sC + rl * LDC + b * 64 + w);
                    const float4 g = *reinterpret_cast<const float4*>(sC + rl * LDC + b * 64 + 32 + w);
                    const float hv[4] = {h.x, h.y, h.z, h.w}, gv[4] = {g.x, g.y, g.z, g.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float a = fmaf(hv[e], p.alpha, bh[e]);
                        const float gg = fmaf(gv[e], p.alpha, bg[e]);
                        v[e] = a * 0.5f * gg * (1.f + dwg_erf_fast(gg * 0.70710678118654752f));
                    }
                    const long long ci = coff + (long long)row * p.ldc + (n0 + b * 64) / 2 + w;
                    if (p.out_bf16) {
                        if (vec_store) {
                            st_half4(p.C, ci, v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) st_half1(p.C, ci + e, v[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) reinterpret_cast<float*>(p.C)[ci + e] = v[e];
                    }
                }
            }
            continue;
        }
        if (fast) {
            // The common layers: 16-byte pieces everywhere, N % 4 == 0, plain / slab store.  A thread keeps ONE column group for the whole
            // tile (256 % C4 == 0), so the column bounds test, the per-column bias and every column offset are loop invariants and a piece
            // costs ~20-45 VALU instructions instead of the ~190 of the general loop below.  This matters: with one or two waves per SIMD the
            // prologue + epilogue instruction stream (4 cycles per wave64 VALU instruction) IS the launch time of the 5-20-k-step layers.
            constexpr int RSTEP = NT / C4;
            const int c4 = tid & (C4 - 1), rl0 = tid / C4;
            const int col = n0 + c4 * 4;
            if (col < p.N) {
                const bool col_bias = bias && p.bias_row_div <= 0;
                float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_bias) bc = *reinterpret_cast<const float4*>(bias + col);
                const bool slab = p.splitk > 1;
#pragma unroll 2
                for (int rl = rl0; rl < PR; rl += RSTEP) {
                    const int row = row_of(q * PR + rl);
                    if (row < 0) continue;
                    const float4 a = *reinterpret_cast<const float4*>(sC + rl * LDC + c4 * 4);
                    float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
                    if (slab) {
                        float* dst = p.ws + ((long long)ks_id * p.M + row) * p.N + col;
                        if (p.cnt) st_agent4(dst, make_float4(v[0], v[1], v[2], v[3]));
                        else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        continue;
                    }
                    if (col_bias) { v[0] += bc.x; v[1] += bc.y; v[2] += bc.z; v[3] += bc.w; }
                    else if (bias) {
                        const float4 b = *reinterpret_cast<const float4*>(bias + (long long)(row / p.bias_row_div) * p.bias_ld + col);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (p.act == 3) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = v[e] / (1.f + __expf(-v[e]));
                    } else if (p.act != 0) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = apply_act<LIGHT>(v[e], p.act);
                    }
                    if (p.residual) {
                        const long long ri = roff + (long long)row * p.ldr + col;
                        if (p.res_bf16) {
                            float r[4];
                            ld_half4(p.residual, ri, r);
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] += r[e];
                        } else {
                            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + ri);
                            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                        }
                    }
                    const long long ci = coff + (long long)row * p.ldc + col;
                    if (p.out_bf16) {
                        st_half4(p.C, ci, v);
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + ci) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
            continue;
        }
#pragma unroll 1
        for (int idx = tid; idx < PR * C4; idx += NT) {
            const int rl = idx / C4, c4 = idx - rl * C4;
            const int row = row_of(q * PR + rl), col = n0 + c4 * 4;
            if (row < 0 || col >= p.N) continue;
            const float4 a = *reinterpret_cast<const float4*>(sC + rl * LDC + c4 * 4);
            float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
            if (p.splitk > 1) {
                if (p.ws) {                              // slab, reduced by k_splitk_epilogue
                    float* dst = p.ws + ((long long)ks_id * p.M + row) * p.N + col;
                    if (p.cnt) st_agent4(dst, make_float4(v[0], v[1], v[2], v[3]));       // (counters: N % 4 == 0, host-checked)
                    else if ((p.N & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) if (col + e < p.N) dst[e] = v[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (col + e < p.N) atomicAdd(reinterpret_cast<float*>(p.C) + coff + (long long)row * p.ldc + col + e, v[e]);
                }
                continue;
            }
            epilogue_store4<LIGHT>(p, v, row, col, coff, roff, vec_ok);
        }
    }
    if constexpr (!LIGHT) {
        if (p.splitk > 1 && p.cnt) splitk_reduce_in_kernel<BN, BM, NT>(p, sC, n0, tid, coff, roff, vec_ok, row_of, tile_id);
    }
}

// sums the split-K slabs in slice order and applies the fused epilogue (batch == 1); four consecutive columns per thread
// (float4 slab reads, 8/16-byte bias / residual / output pieces) when N % 4 == 0
__global__ __launch_bounds__(256) void k_splitk_epilogue(GemmP p) {
    const long long n = (long long)p.M * p.N;
    if ((p.N & 3) == 0) {
        const bool vec_ok = epilogue_vec_ok(p, 0, 0);
        const int N4 = p.N >> 2;
        const long long n4 = n >> 2;
        // one launch per split-K GEMM and usually ONE piece per thread: the index arithmetic is the kernel -- a 32-bit (row, column group)
        // division instead of the 64-bit one.
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const unsigned iu = (unsigned)i;                        // M * N / 4 < 2^32 for every shape on the path (checked on the host)
            const int row = (int)(iu / (unsigned)N4), col = (int)(iu - (unsigned)row * (unsigned)N4) * 4;
            const float4* src = reinterpret_cast<const float4*>(p.ws) + i;
            float4 a = src[0];
            // the slices' loads go out four at a time and are added in slice order (a rolled load-add loop is a chain of dependent
            // latencies: 3 - 12 slices x ~0.6 us were most of this kernel's 5.5 us); same sums, same bits
            int sidx = 1;
            for (; sidx + 3 < p.splitk; sidx += 4) {
                const float4 u0 = src[(long long)sidx * n4], u1 = src[(long long)(sidx + 1) * n4], u2 = src[(long long)(sidx + 2) * n4],
                             u3 = src[(long long)(sidx + 3) * n4];
                a.x += u0.x; a.y += u0.y; a.z += u0.z; a.w += u0.w;
                a.x += u1.x; a.y += u1.y; a.z += u1.z; a.w += u1.w;
                a.x += u2.x; a.y += u2.y; a.z += u2.z; a.w += u2.w;
                a.x += u3.x; a.y += u3.y; a.z += u3.z; a.w += u3.w;
            }
            if (sidx + 1 < p.splitk) {
                const float4 u0 = src[(long long)sidx * n4], u1 = src[(long long)(sidx + 1) * n4];
                a.x += u0.x; a.y += u0.y; a.z += u0.z; a.w += u0.w;
                a.x += u1.x; a.y += u1.y; a.z += u1.z; a.w += u1.w;
                sidx += 2;
            }
            if (sidx < p.splitk) {
                const float4 u = src[(long long)sidx * n4];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
            float v[4] = {a.x, a.y, a.z, a.w};
            epilogue_store4(p, v, row, col, 0, 0, vec_ok);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splitk; s++) v += p.ws[(long long)s * n + i];
        int row = (int)(i / p.N), col = (int)(i - (long long)row * p.N);
        epilogue_store(p, v, row, col, 0, 0);
    }
}

// the reduce launch of a split-K product whose kernel did not reduce in place (no counters: GemmP::cnt)
static void launch_splitk_epilogue(const GemmP& p, hipStream_t stream) {
    if (!(p.splitk > 1 && p.ws) || p.cnt) return;
    long long n = (long long)p.M * p.N;
    if ((p.N & 3) == 0) n >>= 2;                 // four columns per thread
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
    DWG_LAUNCH("splitk_epilogue", k_splitk_epilogue, dim3(blocks), dim3(256), 0, stream, p);
}
// the in-kernel reduce needs one counter per output tile (the workspace header holds DWG_GEMM_WS_COUNTERS) and four-column pieces
static GemmP with_counters(const GemmP& p, long long tiles) {
    GemmP q = p;
    if (!(q.cnt && q.splitk > 1 && q.ws && (q.N & 3) == 0 && tiles <= DWG_GEMM_WS_COUNTERS)) q.cnt = nullptr;
    return q;
}

template <typename T, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256, (sizeof(T) == 4 || BN == 64) ? 2 : 1) void k_gemm(GemmP p) {   // <= 256 registers where they suffice: accumulators
                                                     // in arch VGPRs (see k_gemm_glds); the register-staged 16-bit 128 x 128 tile needs more
    constexpr int BM = 128, BK = TT<T>::BK, LDT = BK + TT<T>::PAD;
    constexpr int WN = BN / 64;            // waves along N (2 for BN=128, 1 for BN=64)
    constexpr int WM = 4 / WN;             // waves along M
    constexpr int TM = BM / WM / 32;       // 32x32 tiles per wave along M (2 | 1)
    constexpr int TN = 2;                  // each wave spans 64 columns
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sA = reinterpret_cast<T*>(smem_raw);            // [2][BM][LDT]
    T* sB = sA + 2 * BM * LDT;                         // [2][BN][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int ntn = (p.N + BN - 1) / BN;
    const int n0 = (blockIdx.y % ntn) * BN;
    const int ks_id = blockIdx.y / ntn;                // split-K slice
    const int z = blockIdx.z, z1 = z / p.nb2, z2 = z - z1 * p.nb2;
    const T* A = reinterpret_cast<const T*>(p.A) + z1 * p.bA1 + z2 * p.bA2;
    const T* B = reinterpret_cast<const T*>(p.B) + z1 * p.bB1 + z2 * p.bB2;
    // contraction range of this slice (multiple of BK so that vector loads stay aligned)
    int kbeg = 0, kend = p.K;
    if (p.splitk > 1) {
        int per = ((p.K + p.splitk - 1) / p.splitk + BK - 1) / BK * BK;
        kbeg = ks_id * per; kend = min(p.K, kbeg + per);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    TileLoader<T, BM, AMODE> la;
    TileLoader<T, BN, BMODE> lb;
    la.init(A, p.sam, p.sak, p.M, m0, kbeg, kend, p.conv);
    lb.init(B, p.sbn, p.sbk, p.N, n0, kbeg, kend, p.conv);
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        la.load(p.conv); lb.load(p.conv);
        la.store(sA); lb.store(sB);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.advance(p.conv); lb.advance(p.conv);
            la.load(p.conv); lb.load(p.conv);
        }
        const T* a = sA + cur * BM * LDT + (wm * TM * 32) * LDT;
        const T* b = sB + cur * BN * LDT + (wn * 64) * LDT;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) mma_tile<T>(b + j * 32 * LDT, a + i * 32 * LDT, lane, acc[i][j]);   // transposed: see tile_epilogue_t
        if (kt + 1 < nk) { la.store(sA + (cur ^ 1) * BM * LDT); lb.store(sB + (cur ^ 1) * BN * LDT); }
        __syncthreads();
    }
    constexpr int NPASS = EpiLds<BN>::passes((size_t)2 * (BM + BN) * LDT * sizeof(T));
    static_assert((size_t)2 * (BM + BN) * LDT * sizeof(T) >= EpiLds<BN>::bytes(NPASS), "epilogue staging fits in the operand stages");
    tile_epilogue_lds<BN, NPASS, TM, TN, false>(p, acc, reinterpret_cast<float*>(smem_raw), n0, wm, wn, lane, tid, ks_id, z1 * p.bC1 + z2 * p.bC2,
                                         z1 * p.bR1 + z2 * p.bR2, [&](int rl) { const int r = m0 + rl; return r < p.M ? r : -1; },
                                         (int)blockIdx.x * ntn + (int)(blockIdx.y % ntn));
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 fast path: operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write pass).
// The LDS image of a tile is lane-linear per wave-instruction (8 rows x 128 B), so bank conflicts are avoided by an XOR
// swizzle applied on the SOURCE side: LDS slot (row, c) holds logical 16-byte chunk c ^ swz(row), swz(row) = (row >> 1) & 7;
// readers apply the same XOR.  ds_read_b128 is serviced in 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) over
// 64 banks: a group's 16 fragment rows cover every residue mod 16, and (row & 1, swz(row)) then selects 16 distinct
// 16-byte bank slots -> conflict-free (row & 7 as the XOR would pair them up 2-way).
// Halo / out-of-range chunks are sourced from a 16-byte zero page.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned char g_zero16[16];

template <int ROWS, bool CONV, int NW = 4>
struct GldsLoader {
    static constexpr int NJ = ROWS / (8 * NW);     // wave-instructions per wave per tile (each covers 8 rows); even
    const HT* rowptr[CONV ? 1 : NJ];
    int iy0[CONV ? NJ : 1], ix0[CONV ? NJ : 1];
    long long pix0[CONV ? NJ : 1];
    bool rok[CONV ? NJ : 1];
    const HT* base;
    // Two k-positions per lane: the logical chunk a lane fetches is slot ^ swz(row) with swz(row) = (row >> 1) & 7, and
    // row = (wave*NJ + j)*8 + sub, so it depends on the parity of j (bit 2 of the XOR) -- [0] even j, [1] odd j.
    int kc[2], kend, ci[2], ky[2], kx[2];

    __device__ __forceinline__ void init(const HT* base_, long long srow, int nrows, int r0, int kbeg, int kend_, const ConvP& cv) {
        static_assert(NJ % 2 == 0, "row-block parity must equal j parity");
        base = base_; kend = kend_;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 3;
        const int lg0 = (lane & 7) ^ (sub >> 1);
        kc[0] = kbeg + lg0 * 8; kc[1] = kbeg + (lg0 ^ 4) * 8;
        if (!CONV) {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                int r = r0 + (wave * NJ + j) * 8 + sub;
                rowptr[j] = r < nrows ? base + (long long)r * srow : nullptr;
            }
        } else {
            const int hw = cv.Hout * cv.Wout;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                int r = r0 + (wave * NJ + j) * 8 + sub;
                rok[j] = r < nrows;
                int rr = rok[j] ? r : 0;
                int img = rr / hw, rem = rr - img * hw;
                int oy = rem / cv.Wout, ox = rem - oy * cv.Wout;
                iy0[j] = oy * cv.stride - cv.pad_t; ix0[j] = ox * cv.stride - cv.pad_l;
                pix0[j] = (long long)img * cv.Hin * cv.Win;
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                int tap = kc[q] / cv.Cin;
                ci[q] = kc[q] - tap * cv.Cin; ky[q] = tap / cv.KW; kx[q] = tap - ky[q] * cv.KW;
            }
        }
    }
    __device__ __forceinline__ void advance(const ConvP& cv) {
        kc[0] += 64; kc[1] += 64;
        if (CONV) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                ci[q] += 64;
                while (ci[q] >= cv.Cin) { ci[q] -= cv.Cin; if (++kx[q] == cv.KW) { kx[q] = 0; ++ky[q]; } }
            }
        }
    }
    // lds_tile: byte address of this operand's tile in LDS (workgroup-uniform)
    __device__ __forceinline__ void issue(unsigned char* lds_tile, const ConvP& cv) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const HT* zero = reinterpret_cast<const HT*>(g_zero16);
        const HT* src0[2] = {base, base}; int cs[2] = {0, 0}, co[2] = {0, 0};
        bool kok[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            kok[q] = kc[q] < kend;
            if (CONV) {
                src0[q] = ci[q] < cv.cin1 ? base : reinterpret_cast<const HT*>(cv.A2);
                cs[q] = ci[q] < cv.cin1 ? cv.cin1 : (cv.Cin - cv.cin1);
                co[q] = ci[q] < cv.cin1 ? ci[q] : ci[q] - cv.cin1;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int q = j & 1;
            const HT* src;
            if (!CONV) {
                src = (kok[q] && rowptr[j]) ? rowptr[j] + kc[q] : zero;
            } else {
                int iy = iy0[j] + ky[q], ix = ix0[j] + kx[q];
                bool ok = kok[q] && rok[j] && iy >= 0 && ix >= 0;
                if (cv.dil > 1) { ok = ok && (iy % cv.dil == 0) && (ix % cv.dil == 0); iy /= cv.dil; ix /= cv.dil; }
                if (cv.up > 1) { iy >>= 1; ix >>= 1; }
                ok = ok && iy < cv.Hin && ix < cv.Win;
                src = ok ? src0[q] + (pix0[j] + (long long)iy * cv.Win + ix) * cs[q] + co[q] : zero;
            }
            unsigned char* dst = lds_tile + ((wave * NJ + j) * 8) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
};

// Convolution A-loader for the common case (Cin % 64 == 0, no input dilation / upsampling / concat): the generic loader redoes
// the whole im2col index computation (~70 VALU instructions per MFMA in the 128x64-tile kernel, which made that kernel
// instruction-issue-bound at ~10 % MFMA utilisation).  Here everything lane-dependent is hoisted out of the k-loop:
//   * a 64-wide k-tile lies inside ONE tap (Cin % 64 == 0), so every lane of the workgroup fetches the same (ky, kx, ci) --
//     the tap offset ((ky*Win + kx)*Cin + ci) is wave-uniform (scalar registers);
//   * per row: a base pointer at (iy0, ix0) [may lie outside the image, only dereferenced when valid] that already contains the
//     lane's swizzled chunk offset, and a KH*KW-bit validity mask (row in range, tap inside the image).
// Per k-step and row this leaves: test one mask bit, one 64-bit add, one select.
template <int ROWS, int NW = 4, bool CAT = false>
struct GldsConvFast {
    static constexpr int NJ = ROWS / (8 * NW);
    const unsigned char* pb[NJ];
    const unsigned char* pb2[CAT ? NJ : 1];    // CAT (round 6): the skip-connection concat fused into the conv -- channels [0, cin1) from A, the rest
                                               // from A2, both multiples of 64: a k-tile lies inside one tap AND one source (wave-uniform choice)
    unsigned int mask[NJ];
    int kcur, kend, tap, ci, ky, kx;       // wave-uniform

    __device__ __forceinline__ void init(const HT* base, long long, int nrows, int r0, int kbeg, int kend_, const ConvP& cv) {
        static_assert(NJ % 2 == 0, "row-block parity must equal j parity");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 3;
        const int lg0 = (lane & 7) ^ (sub >> 1);
        kcur = kbeg; kend = kend_;
        tap = kbeg / cv.Cin; ci = kbeg - tap * cv.Cin; ky = tap / cv.KW; kx = tap - ky * cv.KW;
        const int hw = cv.Hout * cv.Wout;
        const int cs1 = CAT ? cv.cin1 : cv.Cin;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int r = r0 + (wave * NJ + j) * 8 + sub;
            const bool rok = r < nrows;
            const int rr = rok ? r : 0;
            const int img = rr / hw, rem = rr - img * hw;
            const int oy = rem / cv.Wout, ox = rem - oy * cv.Wout;
            const int iy0 = oy * cv.stride - cv.pad_t, ix0 = ox * cv.stride - cv.pad_l;
            unsigned int m = 0;
            for (int t = 0, y = 0; y < cv.KH; y++)
                for (int x = 0; x < cv.KW; x++, t++) {
                    const int iy = iy0 + y, ix = ix0 + x;
                    if (rok && iy >= 0 && iy < cv.Hin && ix >= 0 && ix < cv.Win) m |= 1u << t;
                }
            mask[j] = m;
            const long long pix = ((long long)img * cv.Hin + iy0) * cv.Win + ix0;
            pb[j] = reinterpret_cast<const unsigned char*>(base) + (pix * cs1 + (lg0 ^ ((j & 1) << 2)) * 8) * 2;
            if (CAT) pb2[j] = reinterpret_cast<const unsigned char*>(cv.A2) + (pix * (cv.Cin - cv.cin1) + (lg0 ^ ((j & 1) << 2)) * 8) * 2;
        }
    }
    __device__ __forceinline__ void advance(const ConvP& cv) {
        // branch-free (scalar selects): a branch here would cut the k-step's basic block in two and keep the scheduler from placing the
        // tile loads between the MFMAs
        kcur += 64; ci += 64;
        const int w = ci >= cv.Cin ? 1 : 0;
        ci = w ? 0 : ci; tap += w; kx += w;
        const int w2 = kx == cv.KW ? 1 : 0;
        kx = w2 ? 0 : kx; ky += w2;
    }
    __device__ __forceinline__ void issue(unsigned char* lds_tile, const ConvP& cv) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned char* zero = g_zero16;
        const bool second = CAT && ci >= cv.cin1;                                       // scalar
        const int cs = CAT ? (second ? cv.Cin - cv.cin1 : cv.cin1) : cv.Cin;
        const long long uoff = (((long long)ky * cv.Win + kx) * cs + (second ? ci - cv.cin1 : ci)) * 2;      // scalar
        const unsigned int bit = kcur < kend ? (1u << tap) : 0u;                        // scalar (tiles past kend: zero page)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned char* src = (mask[j] & bit) ? ((CAT && second) ? pb2[j] : pb[j]) + uoff : zero;
            unsigned char* dst = lds_tile + ((wave * NJ + j) * 8) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
};

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt unconstrained): gfx9 encoding vmcnt = simm16[15:14] : simm16[3:0]
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// S-stage software pipeline: S-1 k-tiles are in flight (direct-to-LDS loads) while one is being multiplied; one barrier per
// k-step.  Tiles past kend are still issued (from a 16-byte zero line) so that the outstanding-load count the s_waitcnt
// relies on is a compile-time constant.
template <int BM, int AKIND, int NW = 4> struct ALoaderOf { typedef GldsLoader<BM, false, NW> type; };
template <int BM, int NW> struct ALoaderOf<BM, 1, NW> { typedef GldsLoader<BM, true, NW> type; };
template <int BM, int NW> struct ALoaderOf<BM, 2, NW> { typedef GldsConvFast<BM, NW, false> type; };
template <int BM, int NW> struct ALoaderOf<BM, 3, NW> { typedef GldsConvFast<BM, NW, true> type; };

// AKIND: 0 = plain rows (linear layers), 1 = generic implicit-GEMM convolution, 2 = convolution fast path (GldsConvFast), 3 = the fast path
// over two concatenated sources (A | A2)
// Tile geometry: 128 x 64 (four waves of 32 x 64), 128 x 128 (four waves of 64 x 64) and -- round 6 -- 256 x 128 / 128 x 256 on EIGHT waves of
// 64 x 64 (512 threads, one workgroup per CU, three 48-KiB stages).  Why the big tiles: the mid / small-M layers of the denoiser are bound by
// the operand bytes a CU can keep in flight global -> LDS (Little's law: LDS capacity / load latency), not by the MFMA pipe; a tile with twice
// the area does twice the multiply-adds per byte in flight.
template <int BM, int BN> struct GldsGeom {
    static constexpr int WN = BN / 64, WM = (BM == 128 && BN == 64) ? 4 : BM / 64, NW = WM * WN, NT = NW * 64, TM = BM / WM / 32, TN = 2;
};

// DBG (timing experiments only, results are garbage; DWG_GEMM_DEBUG=n routes the conv-fast 128 x 64 and 256 x 128 launches here):
//   1 = no fragment reads, no MFMAs (loads + barriers only)   2 = no tile loads (LDS reads + MFMAs + barriers)   3 = MFMAs only
template <int BM, int BN, int AKIND, int S, int DBG = 0>
__device__ __forceinline__ void gemm_glds_body(const GemmP& p) {
    typedef GldsGeom<BM, BN> Geo;
    constexpr int WN = Geo::WN, WM = Geo::WM, NW = Geo::NW, NT = Geo::NT, TM = Geo::TM, TN = Geo::TN;
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];     // [2][A tile | B tile], rows of 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: the dispatcher hands consecutive workgroup ids to the 8 XCDs round-robin, each with a private
    // L2.  Remap so that every XCD owns a CONTIGUOUS run of tiles, n fastest: the n-tiles of one A panel (and, for wide N,
    // neighbouring B panels) are then co-resident on one L2 instead of being fetched from HBM / Infinity Cache by 8 of them.
    const int gm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = id % 8, loc = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;     // bijective for any nwg
    }
    const int tile = id % (gm * ntn), ks_id = id / (gm * ntn);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int z = blockIdx.z, z1 = z / p.nb2, z2 = z - z1 * p.nb2;
    const HT* A = reinterpret_cast<const HT*>(p.A) + z1 * p.bA1 + z2 * p.bA2;
    const HT* B = reinterpret_cast<const HT*>(p.B) + z1 * p.bB1 + z2 * p.bB2;
    int kbeg = 0, kend = p.K;
    if (p.splitk > 1) {
        int per = ((p.K + p.splitk - 1) / p.splitk + 63) / 64 * 64;
        kbeg = ks_id * per; kend = min(p.K, kbeg + per);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#ifdef DWG_GEMM_X_TU
    f32x16 acx[TM][TN];                 // the two cross products al bh + ah bl (lo planes carry a factor 2^11: dwg_xfmt.h)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acx[i][j][r] = 0.f;
#endif
    typename ALoaderOf<BM, AKIND, NW>::type la;
    GldsLoader<BN, false, NW> lb;
    la.init(A, p.sam, p.M, m0, kbeg, kend, p.conv);
    lb.init(B, p.sbn, p.N, n0, kbeg, kend, p.conv);
    const int nk = DBG == 6 || DBG == 7 ? 0 : (kend > kbeg ? (kend - kbeg + 63) / 64 : 0);
    constexpr int LPT = (BM + BN) * 128 / (NT * 16);    // direct-to-LDS loads per thread per stage
#pragma unroll
    for (int s = 0; s < S - 1; s++) {
        if (DBG == 7) break;
        la.issue(smem_raw + s * STAGE, p.conv); lb.issue(smem_raw + s * STAGE + ABYTES, p.conv);
        la.advance(p.conv); lb.advance(p.conv);
    }
    // fragment addressing: row r, logical chunk cl -> byte r*128 + ((cl ^ swz(r)) << 4); here swz(r) == (lane >> 1) & 7
    const int frow = lane & 31, fx = (lane >> 1) & 7, fh = lane >> 5;
    // ---- main loop (round 6: software-pipelined inside the wave).  Rounds 2-5 ran, per k-step and wave: barrier -> ~55 address instructions
    // + the tile's direct-to-LDS loads -> [ds_read fragments -> s_waitcnt lgkmcnt(0) -> MFMAs] per 16-k slab -- the LDS latency exposed four
    // times per k-step and the load issue in front of the MFMAs instead of between them: the MFMA pipe idled ~3/4 of the time (24-27 % of
    // its rate on every mid / small-M layer, whatever the tile).  Now the fragments of slab s + 1 (the first slab of tile kt + 1 included:
    // the barrier sits in the MIDDLE of the k-step, before that read) are in flight while slab s is multiplied, and the next tile's loads are
    // issued in the shadow of the first slab's MFMAs.  Two fragment sets: 64 more registers in the 64 x 64 f32x wave tile (still < 256).
#ifdef DWG_GEMM_X_TU
    constexpr int KS = 2;               // 16-k MFMA slabs per 64-half k-step: a 128-byte row holds 32 logical k as [h0 l0 h1 l1 h2 l2 h3 l3]
    struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
#else
    constexpr int KS = 4;
    struct Frag { bf16x8 af[TM], bf[TN]; };
#endif
    auto load_frags = [&](int stage, int ks, Frag& f) {
        const unsigned char* ta = smem_raw + stage * STAGE + (wm * TM * 32 + frow) * 128;
        const unsigned char* tb = smem_raw + stage * STAGE + ABYTES + (wn * 64 + frow) * 128;
#ifdef DWG_GEMM_X_TU
        // MFMA slab ks (16 k): lanes 0-31 take group 2 ks, lanes 32-63 group 2 ks + 1 -- hi chunk 4 ks + 2 fh, lo chunk right behind it
        const int offh = (((ks * 4 + fh * 2) ^ fx) << 4), offl = (((ks * 4 + fh * 2 + 1) ^ fx) << 4);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(ta + i * 32 * 128 + offh);
            f.al[i] = *reinterpret_cast<const bf16x8*>(ta + i * 32 * 128 + offl);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + offh);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + offl);
        }
#else
        const int off = (((ks * 2 + fh) ^ fx) << 4);
#pragma unroll
        for (int i = 0; i < TM; i++) f.af[i] = *reinterpret_cast<const bf16x8*>(ta + i * 32 * 128 + off);
#pragma unroll
        for (int j = 0; j < TN; j++) f.bf[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + off);
#endif
    };
    auto mma = [&](const Frag& f) {
#ifdef DWG_GEMM_X_TU
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(f.bh[j], f.ah[i], acc[i][j]);   // transposed: see tile_epilogue_t
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(f.bl[j], f.ah[i], acx[i][j]);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(f.bh[j], f.al[i], acx[i][j]);
#else
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(f.bf[j], f.af[i], acc[i][j]);   // transposed: see tile_epilogue_t
#endif
    };
    // vmcnt <= N and lgkmcnt == 0 in one s_waitcnt: the loads of the tile read next have landed, and this wave's own fragment reads of the
    // stage that is refilled after the barrier have completed (they are consumed only AFTER the barrier now)
    auto wait_tile_and_lds = [&]() {
        constexpr int N = (S - 2) * LPT;
        __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
    };
    Frag F[2];
    int cur = 0, nxt = S - 1;
    if (nk > 0) {
        wait_vmcnt<(S - 2) * LPT>();        // this thread's loads of tile 0 have landed ...
        __builtin_amdgcn_s_barrier();       // ... and everybody's
        load_frags(0, 0, F[0]);
        __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0): the loop is entered with nothing pending (see the wait at its end)
    }
    // MFMAs per slab and wave: the issue of the next tile's loads is spread over the first slab's (igrouplp hints: one MFMA, then a load and
    // its address arithmetic in the MFMA's shadow)
#ifdef DWG_GEMM_X_TU
    constexpr int MPS = 3 * TM * TN;
#else
    constexpr int MPS = TM * TN;
#endif
    for (int kt = 0; kt < nk; kt++) {
        const int cnext = cur + 1 == S ? 0 : cur + 1;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + 1 < KS) {
                if (DBG != 1 && DBG != 3) load_frags(cur, ks + 1, F[(ks + 1) & 1]);
            } else {
                wait_tile_and_lds();
                __builtin_amdgcn_s_barrier();       // tile kt + 1 is in LDS for everybody; everybody is done reading tile kt
                if (DBG != 1 && DBG != 3) load_frags(cnext, 0, F[0]);         // (past the last tile: a stage of zero-page / stale rows, never multiplied)
            }
            __builtin_amdgcn_sched_barrier(0);      // the fragment reads stay AHEAD of the MFMAs that hide their latency
            if (ks == 0) {
                // tile kt + S - 1 goes into the stage every wave finished reading before the barrier inside the PREVIOUS k-step
                if (DBG != 2 && DBG != 3) { la.issue(smem_raw + nxt * STAGE, p.conv); lb.issue(smem_raw + nxt * STAGE + ABYTES, p.conv); }
                la.advance(p.conv); lb.advance(p.conv);
            }
            if (DBG != 1) mma(F[ks & 1]);
            if (ks == 0) {
#pragma unroll
                for (int m = 0; m < MPS; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, (LPT * 10 + MPS - 1) / MPS, 0); // VALU / SALU of a load's address
                    __builtin_amdgcn_sched_group_barrier(0x020, (LPT + MPS - 1) / MPS, 0);      // the load(s)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // nothing of this wave's is pending in LDS when the next k-step starts (the reads above went out a slab's MFMAs ago: free), so the
        // compiler's counter bookkeeping does not have to wait for the NEXT slab's reads before the first MFMA of the loop
        __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0) only
        cur = cnext; nxt = nxt + 1 == S ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();                        // drain the zero-line tail loads before LDS is handed back
#ifdef DWG_GEMM_X_TU
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = fmaf(acx[i][j][r], DWG_X_LO_INV, acc[i][j][r]);
#endif
    constexpr int NPASS = EpiLds<BN, BM>::passes((size_t)S * STAGE);
    static_assert((size_t)S * STAGE >= EpiLds<BN, BM>::bytes(NPASS), "epilogue staging fits in the operand stages");
    if (DBG == 5) { if (acc[0][0][0] == 12345.f) reinterpret_cast<float*>(p.C)[0] = 1.f; return; }      // timing: no epilogue
    tile_epilogue_lds<BN, NPASS, TM, TN, false, BM, NT>(p, acc, reinterpret_cast<float*>(smem_raw), n0, wm, wn, lane, tid, ks_id, z1 * p.bC1 + z2 * p.bC2,
                                         z1 * p.bR1 + z2 * p.bR2, [&](int rl) { const int r = m0 + rl; return r < p.M ? r : -1; }, tile);
}

template <int BN, int AKIND, int S>
__global__ __launch_bounds__(256, 2) void k_gemm_glds(GemmP p) {     // 2 waves / SIMD = the LDS-bound occupancy anyway; with <= 256 registers the
                                                                      // accumulators stay in arch VGPRs (no v_accvgpr copies in prologue / epilogue)
    gemm_glds_body<128, BN, AKIND, S>(p);
}

// the eight-wave tiles (256 x 128 | 128 x 256): one workgroup per CU = 2 waves / SIMD
template <int BM, int BN, int AKIND, int S>
__global__ __launch_bounds__(512, 2) void k_gemm_glds8(GemmP p) {
    static_assert(GldsGeom<BM, BN>::NT == 512, "eight waves");
    gemm_glds_body<BM, BN, AKIND, S>(p);
}
#ifdef DWG_GEMM_X_TU
template <int DBG> __global__ __launch_bounds__(256, 2) void k_gemm_dbg4(GemmP p) { gemm_glds_body<128, 64, 2, 3, DBG>(p); }
template <int DBG> __global__ __launch_bounds__(512, 2) void k_gemm_dbg8(GemmP p) { gemm_glds_body<256, 128, 2, 3, DBG>(p); }
template <int DBG> static void launch_dbg(const GemmP& p, bool big, hipStream_t stream) {
    if (big) {
        const size_t lds = (size_t)3 * 384 * 128;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_dbg8<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid(((p.M + 255) / 256) * ((p.N + 127) / 128) * (p.splitk > 1 ? p.splitk : 1));
        hipLaunchKernelGGL((k_gemm_dbg8<DBG>), grid, dim3(512), lds, stream, p);
    } else {
        const size_t lds = (size_t)3 * 192 * 128;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_dbg4<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid(((p.M + 127) / 128) * ((p.N + 63) / 64) * (p.splitk > 1 ? p.splitk : 1));
        hipLaunchKernelGGL((k_gemm_dbg4<DBG>), grid, dim3(256), lds, stream, p);
    }
}
#endif

// Algorithmic flops of one launch for the profiler table: 2*M*N*K, with the zero taps of an input-dilated (strided-conv
// dgrad) convolution not counted.
static double gemm_flops(const GemmP& p, int batch) {
    double f = 2.0 * p.M * p.N * p.K * batch;
#ifdef DWG_GEMM_X_TU
    f *= 0.5;           // p.K counts physical halves (hi + lo plane); the three MFMAs per product are ONE algorithmic multiply-add
#endif
    if (p.conv.enabled && p.conv.dil > 1) f /= (double)(p.conv.dil * p.conv.dil);
    return f;
}

template <int BN, int AKIND, int S>
static void launch_glds_s(const GemmP& p_, int batch, hipStream_t stream, const char* name) {
    const GemmP p = with_counters(p_, (long long)((p_.M + 127) / 128) * ((p_.N + BN - 1) / BN));
    size_t lds = (size_t)S * (128 + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_glds<BN, AKIND, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(((p.M + 127) / 128) * ((p.N + BN - 1) / BN) * (p.splitk > 1 ? p.splitk : 1), 1, batch);
    // profiler symbols spelled as rocprofv3 prints the instantiation: k_gemm_glds<BN, AKIND, S>
    static const char* const sym[2][2][4] = {{{"k_gemm_glds<64, 0, 2>", "k_gemm_glds<64, 1, 2>", "k_gemm_glds<64, 2, 2>", "k_gemm_glds<64, 3, 2>"},
                                              {"k_gemm_glds<64, 0, 3>", "k_gemm_glds<64, 1, 3>", "k_gemm_glds<64, 2, 3>", "k_gemm_glds<64, 3, 3>"}},
                                             {{"k_gemm_glds<128, 0, 2>", "k_gemm_glds<128, 1, 2>", "k_gemm_glds<128, 2, 2>", "k_gemm_glds<128, 3, 2>"},
                                              {"k_gemm_glds<128, 0, 3>", "k_gemm_glds<128, 1, 3>", "k_gemm_glds<128, 2, 3>", "k_gemm_glds<128, 3, 3>"}}};
    DWG_LAUNCH_W(name, sym[BN == 128][S == 3][AKIND], gemm_flops(p, batch), (k_gemm_glds<BN, AKIND, S>), grid, dim3(256), lds, stream, p);
    launch_splitk_epilogue(p, stream);
}

// The eight-wave tiles: BM x BN = 256 x 128 | 128 x 256, 512 threads, S stages of 48 KiB (S = 3: 144 of the CU's 160 KiB)
template <int BM, int BN, int AKIND, int S>
static void launch_glds8(const GemmP& p_, int batch, hipStream_t stream, const char* name) {
    const GemmP p = with_counters(p_, (long long)((p_.M + BM - 1) / BM) * ((p_.N + BN - 1) / BN));
    const size_t lds = (size_t)S * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_glds8<BM, BN, AKIND, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.splitk > 1 ? p.splitk : 1), 1, batch);
    // (as rocprofv3 prints the instantiation: k_gemm_glds8<BM, BN, AKIND, S>)
    static const char* const sym[2][2][4] = {
        {{"k_gemm_glds8<256, 128, 0, 2>", "k_gemm_glds8<256, 128, 1, 2>", "k_gemm_glds8<256, 128, 2, 2>", "k_gemm_glds8<256, 128, 3, 2>"},
         {"k_gemm_glds8<128, 256, 0, 2>", "k_gemm_glds8<128, 256, 1, 2>", "k_gemm_glds8<128, 256, 2, 2>", "k_gemm_glds8<128, 256, 3, 2>"}},
        {{"k_gemm_glds8<256, 128, 0, 3>", "k_gemm_glds8<256, 128, 1, 3>", "k_gemm_glds8<256, 128, 2, 3>", "k_gemm_glds8<256, 128, 3, 3>"},
         {"k_gemm_glds8<128, 256, 0, 3>", "k_gemm_glds8<128, 256, 1, 3>", "k_gemm_glds8<128, 256, 2, 3>", "k_gemm_glds8<128, 256, 3, 3>"}}};
    DWG_LAUNCH_W(name, sym[S == 3][BN == 256][AKIND], gemm_flops(p, batch), (k_gemm_glds8<BM, BN, AKIND, S>), grid, dim3(512), lds, stream, p);
    launch_splitk_epilogue(p, stream);
}
template <int AKIND>
static void launch_big(const GemmP& p, int bm, int batch, hipStream_t stream, const char* name) {
    static const int stages = getenv("DWG_GEMM_BIG_STAGES") ? atoi(getenv("DWG_GEMM_BIG_STAGES")) : 3;
    if (bm == 256) { if (stages >= 3) launch_glds8<256, 128, AKIND, 3>(p, batch, stream, name); else launch_glds8<256, 128, AKIND, 2>(p, batch, stream, name); }
    else { if (stages >= 3) launch_glds8<128, 256, AKIND, 3>(p, batch, stream, name); else launch_glds8<128, 256, AKIND, 2>(p, batch, stream, name); }
}

// Pipeline depth.  Measured on MI355X at HEAD of round 2 (bench.py, DWG_GEMM_STAGES=2|3, average launch): the 128x64 tile gains from a
// third stage (72 KiB of LDS still leaves the 2 workgroups per CU these small-M launches have anyway): conv fast path 24.6 -> 23.5 us,
// plain rows 17.3 -> 15.6 us; the 128x128 tile loses (96 KiB -> ONE workgroup per CU: 44 -> 59 us, 122 -> 209 us), and so does the
// generic im2col loader (38 -> 41 us: its per-stage index arithmetic is the cost there).  Default: 3 stages for the narrow tile with
// the plain-row / fast-conv loaders, 2 otherwise; DWG_GEMM_STAGES forces one depth everywhere.
template <int BN, int AKIND>
static void launch_glds(const GemmP& p, int batch, hipStream_t stream, const char* name) {
    static const int forced = getenv("DWG_GEMM_STAGES") ? atoi(getenv("DWG_GEMM_STAGES")) : 0;
    const bool three = forced ? forced >= 3 : (BN == 64 && AKIND != 1);
    if (three) launch_glds_s<BN, AKIND, 3>(p, batch, stream, name);
    else launch_glds_s<BN, AKIND, 2>(p, batch, stream, name);
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch.
// The im2col view re-reads every input pixel nine times (once per tap) through L2 -> LDS, and that operand stream -- not
// the MFMA pipe -- bounds the implicit GEMM at these shapes.  Here a workgroup owns an 8 x 16 output-pixel tile: for each
// 64-channel slab it brings the (8+2) x (16+2) halo patch into LDS ONCE (direct-to-LDS loads, same source-side XOR swizzle,
// halo / out-of-image pixels from the zero page) and forms all nine taps from it; only the weight slab streams per tap.
// L2->LDS bytes per flop drop by ~40 % (A: 16 KiB/tap -> 23 KiB/9 taps).
// ---------------------------------------------------------------------------------------------------------------------
template <int BN, bool SPLIT>      // SPLIT: split-K over 64-channel slabs (its own instantiation: the plain one keeps its register budget)
__global__ __launch_bounds__(256, 2) void k_conv3x3_patch(GemmP p) {   // 2 waves per SIMD: LDS allows 2 workgroups per CU anyway
    constexpr int PH = 8, PW = 16, HP = PH + 2, WP = PW + 2, NPIX = HP * WP;     // 180 patch pixels, 128 B each
    constexpr int NPI = (NPIX + 7) / 8;                                          // 23 wave-instructions per patch
    constexpr int WN = BN / 64, WM = 4 / WN, TM = 128 / WM / 32, TN = 2;
    constexpr int PBYTES = NPI * 8 * 128, BBYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];     // [2 patches][2 weight tiles]
    unsigned char* sP = smem_raw;
    unsigned char* sB = smem_raw + 2 * PBYTES;
    const ConvP& cv = p.conv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (cv.Wout + PW - 1) / PW, tiles_y = (cv.Hout + PH - 1) / PH;
    const int nimg = p.M / (cv.Hout * cv.Wout);
    const int gm = nimg * tiles_y * tiles_x, ntn = (p.N + BN - 1) / BN;
    int id = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = id % 8, loc = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int ks_id = 0;                                        // split-K slice: a contiguous range of 64-channel slabs
    if constexpr (SPLIT) { ks_id = id / (gm * ntn); id -= ks_id * gm * ntn; }
    const int mt = id / ntn, n0 = (id % ntn) * BN;
    const int img = mt / (tiles_y * tiles_x), trem = mt % (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * PH, x0 = (trem % tiles_x) * PW;
    const HT* X = reinterpret_cast<const HT*>(p.A) + (long long)img * cv.Hin * cv.Win * cv.Cin;
    const HT* Wt = reinterpret_cast<const HT*>(p.B);
    const HT* zero = reinterpret_cast<const HT*>(g_zero16);
    const int sub = lane >> 3, lg0 = (lane & 7) ^ (sub >> 1);      // logical chunk of an even 8-row block; odd: ^ 4

    // patch loader state: this wave issues patch instructions wave, wave+4, ... ; lane -> patch pixel (inst*8 + sub)
    auto issue_patch = [&](int cc, unsigned char* dst) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        for (int inst = wv; inst < NPI; inst += 4) {
            const int pi = inst * 8 + sub;
            const int py = pi / WP, px = pi - py * WP;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = pi < NPIX && iy >= 0 && iy < cv.Hin && ix >= 0 && ix < cv.Win;
            const int logical = lg0 ^ ((inst & 1) << 2);
            const HT* src = ok ? X + ((long long)iy * cv.Win + ix) * cv.Cin + cc * 64 + logical * 8 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + inst * 8 * 128), 16, 0, 0);
        }
    };
    // weight slab loader: rows n0 .. n0+BN of Wt[Cout][9*Cin], columns (tap*Cin + cc*64) .. +64.  Row pointers (with the lane's
    // swizzled chunk folded in) are fixed for the whole kernel; per step only a wave-uniform column offset is added.
    constexpr int NJW = BN / 32;
    const HT* wrow[NJW];
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int j = 0; j < NJW; j++) {
            const int r = n0 + (wv * NJW + j) * 8 + sub;
            wrow[j] = r < p.N ? Wt + (long long)r * p.sbn + (lg0 ^ ((j & 1) << 2)) * 8 : nullptr;
        }
    }
    auto issue_w = [&](int cc, int tap, unsigned char* dst) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const long long kofs = (long long)tap * cv.Cin + cc * 64;          // wave-uniform
#pragma unroll
        for (int j = 0; j < NJW; j++) {
            const HT* src = wrow[j] ? wrow[j] + kofs : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (wv * NJW + j) * 8 * 128), 16, 0, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#ifdef DWG_GEMM_X_TU
    f32x16 acx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acx[i][j][r] = 0.f;
#endif
    // this lane's output pixels (rows of the A operand): r = (wm*TM + i)*32 + (lane & 31) -> (oy, ox) = (r >> 4, r & 15)
    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int r = (wm * TM + i) * 32 + (lane & 31);
        pbase[i] = (r >> 4) * WP + (r & 15);
    }
    int cc0 = 0, ncc = cv.Cin / 64;
    if constexpr (SPLIT) {
        const int per = (ncc + p.splitk - 1) / p.splitk;
        cc0 = ks_id * per; ncc = min(ncc, cc0 + per);
    }
    const int frow = lane & 31, fx = (lane >> 1) & 7, fh = lane >> 5;
    if constexpr (SPLIT) {
        if (cc0 < ncc) { issue_patch(cc0, sP + (cc0 & 1) * PBYTES); issue_w(cc0, 0, sB + ((cc0 & 1) ? BBYTES : 0)); }
    } else {
        issue_patch(0, sP);
        issue_w(0, 0, sB);
    }
    // 9 taps unrolled: the tap's patch offset is an immediate and there is no step -> (slab, tap) division in the loop
    for (int cc = cc0; cc < ncc; cc++) {
        const unsigned char* pa = sP + (cc & 1) * PBYTES;
#ifdef DWG_GEMM_X_TU
        // two accumulator sets leave no room for the 9 x TM x 4 loop-invariant fragment addresses the compiler otherwise hoists out of this loop
        // (304 bytes of scratch per lane in the 128-wide instantiation): launder the pixel bases so they are recomputed next to their reads
#pragma unroll
        for (int i = 0; i < TM; i++) asm volatile("" : "+v"(pbase[i]));
#endif
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int par = (cc + tap) & 1;                     // step parity: 9 steps per slab
            __syncthreads();                                    // everything issued so far has landed; older buffers are free
            if (tap < 8) issue_w(cc, tap + 1, sB + (par ^ 1) * BBYTES);
            else if (cc + 1 < ncc) issue_w(cc + 1, 0, sB + (par ^ 1) * BBYTES);
            if (tap == 0 && cc + 1 < ncc) issue_patch(cc + 1, sP + ((cc + 1) & 1) * PBYTES);   // a full slab ahead
            const unsigned char* tb = sB + par * BBYTES + (wn * 64 + frow) * 128;
            const int toff = (tap / 3) * WP + (tap % 3);
#ifdef DWG_GEMM_X_TU
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int cl = ks * 4 + fh * 2;              // hi chunk; the lo chunk is cl + 1 (see k_gemm_glds)
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int pi = pbase[i] + toff, sw = (pi >> 1) & 7;
                    ah[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + ((cl ^ sw) << 4));
                    al[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + (((cl + 1) ^ sw) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + ((cl ^ fx) << 4));
                    bl[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + (((cl + 1) ^ fx) << 4));
                }
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(bh[j], ah[i], acc[i][j]);   // transposed
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(bl[j], ah[i], acx[i][j]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(bh[j], al[i], acx[i][j]);
            }
#else
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int cl = ks * 2 + fh;
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int pi = pbase[i] + toff;
                    af[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + ((cl ^ ((pi >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + ((cl ^ fx) << 4));
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(bf[j], af[i], acc[i][j]);   // transposed
            }
#endif
        }
    }
#ifdef DWG_GEMM_X_TU
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = fmaf(acx[i][j][r], DWG_X_LO_INV, acc[i][j][r]);
#endif
    // epilogue: tile-local row rr <-> output pixel (y0 + rr / 16, x0 + rr % 16) of image img
    constexpr int NPASS = EpiLds<BN>::passes((size_t)2 * PBYTES + 2 * BBYTES);
    static_assert((size_t)2 * PBYTES + 2 * BBYTES >= EpiLds<BN>::bytes(NPASS), "epilogue staging fits in the patch / weight buffers");
    tile_epilogue_lds<BN, NPASS, TM, TN, !SPLIT>(p, acc, reinterpret_cast<float*>(smem_raw), n0, wm, wn, lane, tid, ks_id, 0, 0, [&](int rr) {
        const int y = y0 + (rr >> 4), x = x0 + (rr & 15);
        return (y < cv.Hout && x < cv.Wout) ? (img * cv.Hout + y) * cv.Wout + x : -1;
    }, id);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the LDS-patch convolution with a PIPELINED step.  k_conv3x3_patch above runs, per (64-channel slab, tap): __syncthreads() [= wait
// for EVERY outstanding load: the next tap's weight tile was issued one step earlier, so its latency is exposed each step] -> issue -> read
// fragments -> wait -> MFMAs.  Here (same data layout, same arithmetic, same results):
//   * NBS weight stages: the tile of step g + NBS - 1 is issued in step g and awaited by COUNT (s_waitcnt vmcnt(N), N a compile-time
//     constant per unrolled tap: every wave issues the same number of loads per step -- the last patch instruction is duplicated where the
//     patch does not divide evenly -- and steps past the end load from the zero page);
//   * the fragments of the next 16-k slab -- of the next tap, of the next slab's first tap -- are read while the current one is
//     multiplied; the barrier sits in the middle of a step, after the last slab's fragment reads have completed;
//   * the step's loads are issued between its first slab's MFMAs;
//   * NW = 8: a 16 x 16-pixel tile (256 rows) x BN = 128 on eight waves -- half the weight-tile traffic per multiply-add of the 8 x 16 tile,
//     one workgroup per CU -- for the layers whose grid still fills the chip (the VAE's, the 32^2 / 16^2 levels with split-K).
// ---------------------------------------------------------------------------------------------------------------------
template <bool V> struct BoolC { static constexpr bool value = V; };
template <int BN, int NW> struct Patch2Geom {
    static constexpr int PH = NW == 8 ? 16 : 8, PW = 16, HP = PH + 2, WP = PW + 2, NPIX = HP * WP, BM = PH * PW;
    static constexpr int NPI = (NPIX + 7) / 8;                  // wave-instructions per patch (8 pixels of 128 B each)
    static constexpr int PPW = (NPI + NW - 1) / NW;             // patch loads per thread and slab (the tail duplicates the wave's last one)
    static constexpr int WN = BN / 64, WM = NW / WN, TM = BM / WM / 32, TN = 2, NT = NW * 64;
    static constexpr int NJW = BN / (8 * NW);                   // weight-tile loads per thread and step
    static constexpr int PBYTES = NPI * 8 * 128, BBYTES = BN * 128;
};

template <int BN, bool SPLIT, int NW, int NBS>
__global__ __launch_bounds__(NW * 64, 2) void k_conv3x3_patch2(GemmP p) {
    typedef Patch2Geom<BN, NW> G;
    constexpr int PH = G::PH, PW = G::PW, WP = G::WP, NPIX = G::NPIX, NPI = G::NPI, PPW = G::PPW;
    constexpr int WN = G::WN, WM = G::WM, TM = G::TM, TN = G::TN, NT = G::NT, NJW = G::NJW, BM = G::BM;
    constexpr int PBYTES = G::PBYTES, BBYTES = G::BBYTES;
    static_assert(NJW >= 1 && (NJW % 2 == 0 || NJW == 1), "row-block parity");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];     // [2 patches][NBS weight tiles]
    unsigned char* sP = smem_raw;
    unsigned char* sB = smem_raw + 2 * PBYTES;
    const ConvP& cv = p.conv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (cv.Wout + PW - 1) / PW, tiles_y = (cv.Hout + PH - 1) / PH;
    const int nimg = p.M / (cv.Hout * cv.Wout);
    const int gm = nimg * tiles_y * tiles_x, ntn = (p.N + BN - 1) / BN;
    int id = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = id % 8, loc = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int ks_id = 0;
    if constexpr (SPLIT) { ks_id = id / (gm * ntn); id -= ks_id * gm * ntn; }
    const int mt = id / ntn, n0 = (id % ntn) * BN;
    const int img = mt / (tiles_y * tiles_x), trem = mt % (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * PH, x0 = (trem % tiles_x) * PW;
    const HT* X = reinterpret_cast<const HT*>(p.A) + (long long)img * cv.Hin * cv.Win * cv.Cin;
    const HT* Wt = reinterpret_cast<const HT*>(p.B);
    const HT* zero = reinterpret_cast<const HT*>(g_zero16);
    const int sub = lane >> 3, lg0 = (lane & 7) ^ (sub >> 1);      // logical chunk of an even 8-row block; odd: ^ 4

    // patch loader: this wave issues patch instructions wv, wv + NW, ... (PPW of them: the tail repeats the wave's last valid one -- same
    // source, same destination -- so that every wave's outstanding-load count is the same); lane -> patch pixel inst * 8 + sub.
    // Per-pixel source offsets (in elements, without the slab's channel offset) are loop invariants.
    // (the per-pixel source offsets are recomputed at every slab's tap 0 -- once per nine steps, in the MFMAs' shadow -- instead of living
    // in 2 x PPW registers the 64 x 64 wave tile does not have)
    auto issue_patch = [&](int cc, bool valid, unsigned char* dst) {
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            int inst = wv + q * NW;
            if (inst >= NPI) inst -= NW;                       // duplicate of the previous one
            const int pi = inst * 8 + sub;
            const int py = pi / WP, px = pi - py * WP;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = valid && pi < NPIX && iy >= 0 && iy < cv.Hin && ix >= 0 && ix < cv.Win;
            const HT* src = ok ? X + ((long long)iy * cv.Win + ix) * cv.Cin + (lg0 ^ ((inst & 1) << 2)) * 8 + cc * 64 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + inst * 8 * 128), 16, 0, 0);
        }
    };
    const HT* wrow[NJW];
#pragma unroll
    for (int j = 0; j < NJW; j++) {
        const int r = n0 + (wv * NJW + j) * 8 + sub;
        wrow[j] = r < p.N ? Wt + (long long)r * p.sbn + (lg0 ^ (((wv * NJW + j) & 1) << 2)) * 8 : nullptr;
    }
    auto issue_w = [&](int cc, int tap, bool valid, unsigned char* dst) {
        const long long kofs = (long long)tap * cv.Cin + cc * 64;          // wave-uniform
#pragma unroll
        for (int j = 0; j < NJW; j++) {
            const HT* src = (valid && wrow[j]) ? wrow[j] + kofs : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (wv * NJW + j) * 8 * 128), 16, 0, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#ifdef DWG_GEMM_X_TU
    f32x16 acx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acx[i][j][r] = 0.f;
    constexpr int KS = 2, MPS = 3 * TM * TN;
    struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
#else
    constexpr int KS = 4, MPS = TM * TN;
    struct Frag { bf16x8 af[TM], bf[TN]; };
#endif
    // this lane's output pixels (rows of the A operand): r = (wm*TM + i)*32 + (lane & 31) -> (oy, ox) = (r / PW, r % PW)
    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int r = (wm * TM + i) * 32 + (lane & 31);
        pbase[i] = (r / PW) * WP + (r % PW);
    }
    int cc0 = 0, ncc = cv.Cin / 64;
    if constexpr (SPLIT) {
        const int per = (ncc + p.splitk - 1) / p.splitk;
        cc0 = ks_id * per; ncc = min(ncc, cc0 + per);
    }
    const int frow = lane & 31, fx = (lane >> 1) & 7, fh = lane >> 5;
    const int nsteps = max(0, ncc - cc0) * 9;
    // step g (0-based over this workgroup's slabs): slab cc0 + g / 9, tap g % 9, weight stage g % NBS, patch buffer (g / 9) & 1
    auto load_frags = [&](const unsigned char* pa, int toff, const unsigned char* tbs, int ks, Frag& f) {
        const unsigned char* tb = tbs + (wn * 64 + frow) * 128;
#ifdef DWG_GEMM_X_TU
        const int cl = ks * 4 + fh * 2;                  // hi chunk; the lo chunk is cl + 1 (see gemm_glds_body)
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int pi = pbase[i] + toff, sw = (pi >> 1) & 7;
            f.ah[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + ((cl ^ sw) << 4));
            f.al[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + (((cl + 1) ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + ((cl ^ fx) << 4));
            f.bl[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + (((cl + 1) ^ fx) << 4));
        }
#else
        const int cl = ks * 2 + fh;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int pi = pbase[i] + toff;
            f.af[i] = *reinterpret_cast<const bf16x8*>(pa + pi * 128 + ((cl ^ ((pi >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; j++) f.bf[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + ((cl ^ fx) << 4));
#endif
    };
    auto mma = [&](const Frag& f) {
#ifdef DWG_GEMM_X_TU
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(f.bh[j], f.ah[i], acc[i][j]);   // transposed
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(f.bl[j], f.ah[i], acx[i][j]);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acx[i][j] = DWG_MFMA16(f.bh[j], f.al[i], acx[i][j]);
#else
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = DWG_MFMA16(f.bf[j], f.af[i], acc[i][j]);   // transposed
#endif
    };
    // prologue: the first slab's patch, then the weight tiles of steps 0 .. NBS - 2
    issue_patch(cc0, nsteps > 0, sP + (cc0 & 1) * PBYTES);
#pragma unroll
    for (int g = 0; g < NBS - 1; g++) issue_w(cc0 + g / 9, g % 9, g < nsteps, sB + g * BBYTES);
    Frag F[2];
    if (nsteps > 0) {
        wait_vmcnt<(NBS - 2) * NJW>();          // the patch and the weight tile of step 0 have landed (loads return in order)
        __builtin_amdgcn_s_barrier();
        load_frags(sP + (cc0 & 1) * PBYTES, 0, sB, 0, F[0]);
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): the loop is entered with nothing pending in LDS
    }
    int bst = 0;                                 // weight stage of the current step
    // One step = one tap of one slab.  Tap 0 (which also issues the next slab's patch) is its own copy of the body; taps 1 .. 8 run as a ROLLED
    // loop -- with all nine unrolled the compiler keeps the taps' 72 fragment addresses live across the slab and the 64 x 64 f32x wave tile
    // (128 accumulator + 64 fragment registers) spills.
    auto step = [&](auto is_tap0, int tap, int cc, const unsigned char* pa, const unsigned char* pan) {
        constexpr bool TAP0 = decltype(is_tap0)::value;
        const int ty = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
        const int toff = ty * WP + (tap - 3 * ty);
        const int bnext = bst + 1 == NBS ? 0 : bst + 1;
        const int bfill = bst == 0 ? NBS - 1 : bst - 1;                    // the stage of step g - 1 == the stage of step g + NBS - 1
        const unsigned char* tb = sB + bst * BBYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + 1 < KS) {
                load_frags(pa, toff, tb, ks + 1, F[(ks + 1) & 1]);
            } else {
                // the weight tile of step g + 1 (and, at tap 8, the next slab's patch: issued nine steps ago) has landed; loads issued
                // after it: the tiles of steps g + 2 .. g + NBS - 1, and a patch iff one was issued in steps g - NBS + 2 .. g
                if (TAP0 || tap <= NBS - 2) {
                    constexpr int N = (NBS - 2) * NJW + PPW;
                    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
                } else {
                    constexpr int N = (NBS - 2) * NJW;
                    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
                }
                __builtin_amdgcn_s_barrier();
                const int ntap = tap == 8 ? 0 : tap + 1;
                const int nty = ntap >= 6 ? 2 : (ntap >= 3 ? 1 : 0);
                load_frags(tap == 8 ? pan : pa, nty * WP + (ntap - 3 * nty), sB + bnext * BBYTES, 0, F[0]);   // (past the last step: stale rows, never multiplied)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                // step g + NBS - 1's weight tile (tap + NBS - 1 wraps into the next slab), then -- at tap 0 -- the next slab's patch
                const int ft = tap + NBS - 1, fcc = cc + (ft >= 9 ? 1 : 0), ftap = ft >= 9 ? ft - 9 : ft;
                issue_w(fcc, ftap, fcc < ncc, sB + bfill * BBYTES);
                if (TAP0) issue_patch(cc + 1, cc + 1 < ncc, const_cast<unsigned char*>(pan));
            }
            mma(F[ks & 1]);
            if (ks == 0) {
                constexpr int NL = NJW + (TAP0 ? PPW : 0);
#pragma unroll
                for (int m = 0; m < MPS; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, (NL * 8 + MPS - 1) / MPS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, (NL + MPS - 1) / MPS, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): see gemm_glds_body
        bst = bnext;
    };
    for (int cc = cc0; cc < ncc; cc++) {
        const unsigned char* pa = sP + (cc & 1) * PBYTES;
        const unsigned char* pan = sP + ((cc + 1) & 1) * PBYTES;
        step(BoolC<true>{}, 0, cc, pa, pan);
#pragma unroll 1
        for (int tap = 1; tap < 9; tap++) step(BoolC<false>{}, tap, cc, pa, pan);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifdef DWG_GEMM_X_TU
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = fmaf(acx[i][j][r], DWG_X_LO_INV, acc[i][j][r]);
#endif
    constexpr int NPASS = EpiLds<BN, BM>::passes((size_t)2 * PBYTES + NBS * BBYTES);
    static_assert((size_t)2 * PBYTES + NBS * BBYTES >= EpiLds<BN, BM>::bytes(NPASS), "epilogue staging fits in the patch / weight buffers");
    tile_epilogue_lds<BN, NPASS, TM, TN, !SPLIT, BM, NT>(p, acc, reinterpret_cast<float*>(smem_raw), n0, wm, wn, lane, tid, ks_id, 0, 0, [&](int rr) {
        const int y = y0 + rr / PW, x = x0 + rr % PW;
        return (y < cv.Hout && x < cv.Wout) ? (img * cv.Hout + y) * cv.Wout + x : -1;
    }, id);
}

template <int BN, int NW, int NBS>
static void launch_conv3x3_patch2(const GemmP& p_, hipStream_t stream, const char* name) {
    typedef Patch2Geom<BN, NW> G;
    const size_t lds = (size_t)2 * G::PBYTES + (size_t)NBS * G::BBYTES;
    const ConvP& cv = p_.conv;
    const int nimg = p_.M / (cv.Hout * cv.Wout);
    const int gm = nimg * ((cv.Hout + G::PH - 1) / G::PH) * ((cv.Wout + G::PW - 1) / G::PW);
    const GemmP p = with_counters(p_, (long long)gm * ((p_.N + BN - 1) / BN));
    dim3 grid(gm * ((p.N + BN - 1) / BN) * (p.splitk > 1 ? p.splitk : 1));
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_patch2<BN, false, NW, NBS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_patch2<BN, true, NW, NBS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // profiler symbols spelled as rocprofv3 prints the instantiation: k_conv3x3_patch2<BN, SPLIT, NW, NBS> (bench.py joins the PMC traffic on it)
    static_assert((NW == 8 && BN == 128 && NBS == 4) || (NW == 4 && BN == 64 && NBS == 4) || (NW == 4 && BN == 128 && NBS == 2), "symbol table below");
    const char* sym = NW == 8 ? (p.splitk > 1 && p.ws ? "k_conv3x3_patch2<128, true, 8, 4>" : "k_conv3x3_patch2<128, false, 8, 4>")
                              : (BN == 64 ? (p.splitk > 1 && p.ws ? "k_conv3x3_patch2<64, true, 4, 4>" : "k_conv3x3_patch2<64, false, 4, 4>")
                                          : (p.splitk > 1 && p.ws ? "k_conv3x3_patch2<128, true, 4, 2>" : "k_conv3x3_patch2<128, false, 4, 2>"));
    if (p.splitk > 1 && p.ws)
        DWG_LAUNCH_W(name, sym, gemm_flops(p, 1), (k_conv3x3_patch2<BN, true, NW, NBS>), grid, dim3(NW * 64), lds, stream, p);
    else
        DWG_LAUNCH_W(name, sym, gemm_flops(p, 1), (k_conv3x3_patch2<BN, false, NW, NBS>), grid, dim3(NW * 64), lds, stream, p);
    launch_splitk_epilogue(p, stream);
}

template <int BN>
static void launch_conv3x3_patch(const GemmP& p_, hipStream_t stream, const char* name) {
    constexpr int NPI = (10 * 18 + 7) / 8;
    const size_t lds = (size_t)2 * NPI * 8 * 128 + (size_t)2 * BN * 128;
    const ConvP& cv = p_.conv;
    const int nimg = p_.M / (cv.Hout * cv.Wout);
    const int gm = nimg * ((cv.Hout + 7) / 8) * ((cv.Wout + 15) / 16);
    const GemmP p = with_counters(p_, (long long)gm * ((p_.N + BN - 1) / BN));
    dim3 grid(gm * ((p.N + BN - 1) / BN) * (p.splitk > 1 ? p.splitk : 1));
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_patch<BN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_patch<BN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.splitk > 1 && p.ws)
        DWG_LAUNCH_W(name, (BN == 64 ? "k_conv3x3_patch<64, true>" : "k_conv3x3_patch<128, true>"), gemm_flops(p, 1),
                     (k_conv3x3_patch<BN, true>), grid, dim3(256), lds, stream, p);
    else
        DWG_LAUNCH_W(name, (BN == 64 ? "k_conv3x3_patch<64, false>" : "k_conv3x3_patch<128, false>"), gemm_flops(p, 1), (k_conv3x3_patch<BN, false>), grid,
                     dim3(256), lds, stream, p);
    launch_splitk_epilogue(p, stream);
}

template <typename T, int BN, int AMODE, int BMODE>
static void launch(const GemmP& p_, int batch, hipStream_t stream, const char* name) {
    const GemmP p = with_counters(p_, (long long)((p_.M + 127) / 128) * ((p_.N + BN - 1) / BN));
    constexpr int LDT = TT<T>::BK + TT<T>::PAD;
    size_t lds = (size_t)2 * (128 + BN) * LDT * sizeof(T);
    dim3 grid((p.M + 127) / 128, ((p.N + BN - 1) / BN) * (p.splitk > 1 ? p.splitk : 1), batch);
    static bool attr_set = false;   // > 64 KiB of dynamic LDS needs an explicit opt-in (one per instantiation)
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<T, BN, AMODE, BMODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    DWG_LAUNCH_W(name, (sizeof(T) == 2 ? "k_gemm<" DWG_HALF_NAME ">" : "k_gemm<f32>"), gemm_flops(p, batch), (k_gemm<T, BN, AMODE, BMODE>), grid,
                 dim3(256), lds, stream, p);
    launch_splitk_epilogue(p, stream);
}

template <typename T, int BN, int AMODE>
static void dispatch_b(const GemmP& p, int bmode, int batch, hipStream_t s, const char* name) {
    if (bmode == MODE_KVEC) launch<T, BN, AMODE, MODE_KVEC>(p, batch, s, name);
    else if (bmode == MODE_RVEC) launch<T, BN, AMODE, MODE_RVEC>(p, batch, s, name);
    else launch<T, BN, AMODE, MODE_SCALAR>(p, batch, s, name);
}
template <typename T, int BN>
static void dispatch_a(const GemmP& p, int amode, int bmode, int batch, hipStream_t s, const char* name) {
    if (amode == MODE_KVEC) dispatch_b<T, BN, MODE_KVEC>(p, bmode, batch, s, name);
    else if (amode == MODE_RVEC) dispatch_b<T, BN, MODE_RVEC>(p, bmode, batch, s, name);
    else if (amode == MODE_CONV) dispatch_b<T, BN, MODE_CONV>(p, bmode, batch, s, name);
    else dispatch_b<T, BN, MODE_SCALAR>(p, bmode, batch, s, name);
}

static int batch_of(const dwg_gemm_desc* d) { return d->batch1 * d->batch2; }
static int tile_bn(const dwg_gemm_desc* d) {
    if (d->N <= 64) return 64;
    {
        long long blocks128 = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch_of(d);
        if (blocks128 < 256 && getenv("DWG_GEMM_NO_NARROW") == nullptr) return 64;
    }
    return 128;
}

// split-K factor for shapes that cannot fill 256 CUs with 128 x BN output tiles (small-M layers: 8x8 / 16x16 latents)
static int auto_splitk(int M, int N, int K, int bn, int bk) {
    // round 6: 256 workgroups of >= 12 k-steps (was 512 of >= 8): a weight-streaming slice is bound by 128-byte row pieces out of HBM, and
    // longer slices stream better than more of them (8 x 8-latent convolution 33.4 -> 28.2 us; the step 25.33 -> 25.07 ms)
    static const int target = getenv("DWG_SPLITK_TARGET") ? atoi(getenv("DWG_SPLITK_TARGET")) : 256;
    static const int nosplit = getenv("DWG_SPLITK_NOSPLIT") ? atoi(getenv("DWG_SPLITK_NOSPLIT")) : 384;
    static const int minsteps = getenv("DWG_SPLITK_MINSTEPS") ? atoi(getenv("DWG_SPLITK_MINSTEPS")) : 12;
    long long blocks = (long long)((M + 127) / 128) * ((N + bn - 1) / bn);
    if (blocks >= nosplit || K < 2 * minsteps * bk) return 1;
    if (2.0 * M * N * K < 4.0e8) return 1;      // tiny products are launch-bound: a second (reduce) launch costs more than it buys
    long long sk = target / blocks;             // aim at ~1 workgroup per CU
    long long kmax = K / (minsteps * bk);       // keep >= 12 k-steps per slice
    if (sk > kmax) sk = kmax;
    if (sk > 16) sk = 16;
    return sk >= 2 ? (int)sk : 1;
}

// Round 6: the eight-wave tiles.  Returns BM (256: the 256 x 128 tile, 128: the 128 x 256 tile) or 0 (keep the four-wave tiles), and the
// split-K factor that fills the chip with ONE workgroup per CU.  Shape-only, so that the workspace query and the launch agree.
//   * only where the k-loop is long enough to pay for the larger prologue / epilogue (K >= DWG_GEMM_BIG_MINK physical halves);
//   * M >= 192: 256 x 128 when N fills whole 128-wide tiles (or is large); M <= 128: 128 x 256 for N >= 256;
//   * layers with >= DWG_GEMM_BIG_MAXTILES big tiles already fill the chip with small tiles at full rate (the VAE): left alone.
static int big_tile(int M, int N, int K, int* splitk_out) {
    static const int mode = getenv("DWG_GEMM_BIG") ? atoi(getenv("DWG_GEMM_BIG")) : 1;
    static const int mink = getenv("DWG_GEMM_BIG_MINK") ? atoi(getenv("DWG_GEMM_BIG_MINK")) : 1280;
    static const int skmax = getenv("DWG_GEMM_BIG_SKMAX") ? atoi(getenv("DWG_GEMM_BIG_SKMAX")) : 24;
    static const int maxtiles = getenv("DWG_GEMM_BIG_MAXTILES") ? atoi(getenv("DWG_GEMM_BIG_MAXTILES")) : 2048;
    static const int minsteps = getenv("DWG_GEMM_BIG_MINSTEPS") ? atoi(getenv("DWG_GEMM_BIG_MINSTEPS")) : 8;
    static const int tall_only = getenv("DWG_GEMM_BIG_TALL_ONLY") ? atoi(getenv("DWG_GEMM_BIG_TALL_ONLY")) : 0;
    if (splitk_out) *splitk_out = 1;
    if (mode == 0 || K < mink) return 0;
    int bm = 0;
    if (M >= 192) { if (N >= 128 && (N % 128 == 0 || N >= 640)) bm = 256; }
    else if (N >= 256 && !tall_only) bm = 128;
    if (!bm) return 0;
    const int bn = bm == 256 ? 128 : 256;
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tiles > maxtiles) return 0;
    static const int target = getenv("DWG_GEMM_BIG_TARGET") ? atoi(getenv("DWG_GEMM_BIG_TARGET")) : 256;     // one workgroup per CU
    static const int minwg = getenv("DWG_GEMM_BIG_MINWG") ? atoi(getenv("DWG_GEMM_BIG_MINWG")) : 200;
    long long sk = 1;
    if (tiles < target) {
        sk = target / tiles;            // FLOOR: tiles x sk must not exceed the CU count -- a 257th workgroup is a second round of the whole launch
        const long long kmax = K / (minsteps * 64);
        if (sk > kmax) sk = kmax;
        if (sk > skmax) sk = skmax;
        if (sk < 1) sk = 1;
    }
    // a grid that leaves a fifth of the CUs idle, or spills a few workgroups into a second round: the small tiles' finer grain wins
    const long long wgs = tiles * sk, rounds = (wgs + target - 1) / target;
    if (mode < 2 && (wgs < minwg || wgs * 100 < rounds * target * 80)) return 0;
    if (splitk_out) *splitk_out = (int)sk;
    return bm;
}

template <typename T>
static int pick_mode(const void* base, long long srow, long long sk, int nrows, int K, const long long* boffs, int nboffs) {
    constexpr int VEC = TT<T>::VEC;
    bool aligned = ((uintptr_t)base % 16) == 0;
    for (int i = 0; i < nboffs; i++) aligned = aligned && (boffs[i] % VEC == 0);
    if (sk == 1 && aligned && K % VEC == 0 && srow % VEC == 0) return MODE_KVEC;
    if (srow == 1 && aligned && nrows % VEC == 0 && sk % VEC == 0) return MODE_RVEC;
    return MODE_SCALAR;
}

}  // namespace

#ifdef DWG_GEMM_X_TU
// The split-precision unit takes the LOGICAL descriptor (strides and K in fp32-sized elements, dwg_gemm.h) and hands the kernels the PHYSICAL
// one: a 2-byte tensor with 2 K columns.  C / residual indexing stays logical (st_half4 / ld_half4 know the layout).
static bool x_physical(const dwg_gemm_desc* d, dwg_gemm_desc* o) {
    if (!d) return false;
    *o = *d;
    if (d->dtype != DWG_DTYPE_F32X) return false;
    if (d->K % 8 || d->a_k_stride != 1 || d->b_k_stride != 1) return false;                    // both operands K-contiguous, whole 8-groups
    if ((d->a_row_stride | d->b_row_stride | d->a_batch1_stride | d->a_batch2_stride | d->b_batch1_stride | d->b_batch2_stride) % 8) return false;
    if (d->out_dtype == DWG_DTYPE_F32X && ((d->ldc | d->c_batch1_stride | d->c_batch2_stride) % 8 || ((uintptr_t)d->C % 16))) return false;
    if (d->residual && d->residual_dtype == DWG_DTYPE_F32X &&
        (((d->ldr ? d->ldr : d->ldc) | d->r_batch1_stride | d->r_batch2_stride) % 8 || ((uintptr_t)d->residual % 16))) return false;
    if (d->act == DWG_ACT_GEGLU_PAIR && d->out_dtype == DWG_DTYPE_F32X && d->N % 16) return false;
    if (d->conv_enabled && (d->conv_cin % 8 || (d->A2 && d->conv_cin1 % 8))) return false;
    o->K = 2 * d->K;
    o->a_row_stride = 2 * d->a_row_stride; o->b_row_stride = 2 * d->b_row_stride;
    o->a_batch1_stride = 2 * d->a_batch1_stride; o->a_batch2_stride = 2 * d->a_batch2_stride;
    o->b_batch1_stride = 2 * d->b_batch1_stride; o->b_batch2_stride = 2 * d->b_batch2_stride;
    o->conv_cin = 2 * d->conv_cin; o->conv_cin1 = 2 * d->conv_cin1;
    return true;
}
#endif

extern "C" {

#if defined(DWG_GEMM_X_TU)
#define DWG_GEMM_FN dwg_gemm_x
#define DWG_GEMM_WS_FN dwg_gemm_workspace_bytes_x
#elif defined(DWG_GEMM_F16_TU)
#define DWG_GEMM_FN dwg_gemm_f16
#define DWG_GEMM_WS_FN dwg_gemm_workspace_bytes_f16
#else
#define DWG_GEMM_FN dwg_gemm
#define DWG_GEMM_WS_FN dwg_gemm_workspace_bytes
// the fp16-operand unit (gemm_f16.hip) and the split-precision unit (gemm_x.hip): same descriptor, dtype == DWG_DTYPE_F16 / DWG_DTYPE_F32X
size_t dwg_gemm_workspace_bytes_f16(const dwg_gemm_desc* d);
int dwg_gemm_f16(const dwg_gemm_desc* d, dwg_stream_t stream);
size_t dwg_gemm_workspace_bytes_x(const dwg_gemm_desc* d);
int dwg_gemm_x(const dwg_gemm_desc* d, dwg_stream_t stream);
#endif

size_t DWG_GEMM_WS_FN(const dwg_gemm_desc* d) {
#if !defined(DWG_GEMM_F16_TU) && !defined(DWG_GEMM_X_TU)
    if (d && d->dtype == DWG_DTYPE_F16) return dwg_gemm_workspace_bytes_f16(d);
    if (d && d->dtype == DWG_DTYPE_F32X) return dwg_gemm_workspace_bytes_x(d);
#endif
#ifdef DWG_GEMM_X_TU
    dwg_gemm_desc xd;
    if (!x_physical(d, &xd)) return 0;
    d = &xd;
#endif
    if (!d || d->batch1 * d->batch2 != 1 || d->M <= 0 || d->N <= 0) return 0;
    if ((long long)d->M * d->N >= (1LL << 33)) return 0;       // k_splitk_epilogue indexes the float4 pieces of C with 32 bits
    const int bn = tile_bn(d), bk = d->dtype == DWG_DTYPE_HALF ? TT<HT>::BK : TT<float>::BK;
    int sk = d->splitk > 1 ? d->splitk : (d->splitk == 0 && d->act != DWG_ACT_GEGLU_PAIR ? auto_splitk(d->M, d->N, d->K, bn, bk) : 1);
    if (d->dtype == DWG_DTYPE_HALF && d->splitk == 0 && d->act != DWG_ACT_GEGLU_PAIR) {      // the eight-wave tiles split deeper (one workgroup per CU)
        int bsk = 1;
        if (big_tile(d->M, d->N, d->K, &bsk) && bsk > sk) sk = bsk;
    }
    return sk > 1 ? (size_t)sk * d->M * d->N * sizeof(float) + DWG_GEMM_WS_HEADER_BYTES : 0;     // slabs + the tile-counter header
}

int DWG_GEMM_FN(const dwg_gemm_desc* d, dwg_stream_t stream_) {
#if !defined(DWG_GEMM_F16_TU) && !defined(DWG_GEMM_X_TU)
    if (d && d->dtype == DWG_DTYPE_F16) return dwg_gemm_f16(d, stream_);
    if (d && d->dtype == DWG_DTYPE_F32X) return dwg_gemm_x(d, stream_);
#endif
#ifdef DWG_GEMM_X_TU
    dwg_gemm_desc xd;
    if (!x_physical(d, &xd)) return DWG_E_ARG;
    d = &xd;
#endif
    if (!d || !d->A || !d->B || !d->C) return DWG_E_ARG;
    if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch1 < 1 || d->batch2 < 1) return DWG_E_ARG;
    if (d->M == 0 || d->N == 0) return DWG_OK;
    if (d->dtype != DWG_DTYPE_F32 && d->dtype != DWG_DTYPE_HALF) return DWG_E_ARG;
    if (d->splitk > 1 && !d->workspace && (d->out_dtype != DWG_DTYPE_F32 || d->bias || d->residual || d->act)) return DWG_E_ARG;
    if (d->accumulate && d->out_dtype != DWG_DTYPE_F32) return DWG_E_ARG;
    if (d->act == DWG_ACT_GEGLU_PAIR && (d->N % 64 != 0 || d->residual || d->splitk > 1 || d->bias_per_row || d->bias_row_div)) return DWG_E_ARG;
    // workspace_counters: the first DWG_GEMM_WS_HEADER_BYTES of the workspace are tile counters (zero between launches), the slabs follow
    dwg_gemm_desc wd = *d;
    int* ws_counters = nullptr;
    if (wd.workspace && wd.workspace_counters) {
        if (wd.workspace_bytes <= DWG_GEMM_WS_HEADER_BYTES || ((uintptr_t)wd.workspace & 15)) return DWG_E_ARG;
        ws_counters = reinterpret_cast<int*>(wd.workspace);
        wd.workspace = reinterpret_cast<char*>(wd.workspace) + DWG_GEMM_WS_HEADER_BYTES;
        wd.workspace_bytes -= DWG_GEMM_WS_HEADER_BYTES;
    }
    d = &wd;
    GemmP p;
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.residual = d->residual;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.sam = d->a_row_stride; p.sak = d->a_k_stride; p.sbn = d->b_row_stride; p.sbk = d->b_k_stride;
    p.ldc = d->ldc; p.ldr = d->ldr ? d->ldr : d->ldc;
    p.nb2 = d->batch2;
    p.bA1 = d->a_batch1_stride; p.bA2 = d->a_batch2_stride; p.bB1 = d->b_batch1_stride; p.bB2 = d->b_batch2_stride;
    p.bC1 = d->c_batch1_stride; p.bC2 = d->c_batch2_stride; p.bR1 = d->r_batch1_stride; p.bR2 = d->r_batch2_stride;
    p.act = d->act; p.alpha = d->alpha;
    p.out_bf16 = d->out_dtype == DWG_DTYPE_HALF; p.res_bf16 = d->residual_dtype == DWG_DTYPE_HALF;
    p.bias_per_row = d->bias_per_row; p.splitk = d->splitk > 1 ? d->splitk : 1; p.accumulate = d->accumulate;
    p.ws = nullptr; p.cnt = ws_counters;
    static const int dbg = getenv("DWG_GEMM_DEBUG") ? atoi(getenv("DWG_GEMM_DEBUG")) : 0;
    p.dbg = dbg;
    {
        const int bn = tile_bn(d), bk = d->dtype == DWG_DTYPE_HALF ? TT<HT>::BK : TT<float>::BK;
        if (d->workspace && d->batch1 * d->batch2 == 1 && (long long)d->M * d->N < (1LL << 33)) {
            int sk = d->splitk > 1 ? d->splitk : (d->splitk == 0 && d->act != DWG_ACT_GEGLU_PAIR ? auto_splitk(d->M, d->N, d->K, bn, bk) : 1);
            while (sk > 1 && (size_t)sk * d->M * d->N * sizeof(float) > d->workspace_bytes) sk--;
            p.splitk = sk;
            if (sk > 1) p.ws = reinterpret_cast<float*>(d->workspace);
        }
    }
    // A 1 x 1 / stride 1 / unpadded convolution over NHWC rows IS the plain product of the [M, Cin] row matrix: it takes the plain-row loader
    // (no per-stage pixel arithmetic), DWG_CONV1X1_PLAIN=0: the convolution loader as before
    static const bool conv1x1_plain = !(getenv("DWG_CONV1X1_PLAIN") && atoi(getenv("DWG_CONV1X1_PLAIN")) == 0);
    const bool plain1x1 = conv1x1_plain && d->conv_enabled && d->conv_kh == 1 && d->conv_kw == 1 && d->conv_stride == 1 && d->conv_pad_t == 0 &&
                          d->conv_pad_l == 0 && d->conv_in_dilation <= 1 && d->conv_in_upsample <= 1 && !d->A2 && d->conv_hout == d->conv_hin &&
                          d->conv_wout == d->conv_win && d->K == d->conv_cin && d->batch1 * d->batch2 == 1;
    if (plain1x1) { p.sam = d->conv_cin; p.sak = 1; }
    p.conv.enabled = d->conv_enabled && !plain1x1;
    p.conv.Cin = d->conv_cin; p.conv.Hin = d->conv_hin; p.conv.Win = d->conv_win; p.conv.Hout = d->conv_hout;
    p.conv.Wout = d->conv_wout; p.conv.KH = d->conv_kh; p.conv.KW = d->conv_kw; p.conv.stride = d->conv_stride;
    p.conv.pad_t = d->conv_pad_t; p.conv.pad_l = d->conv_pad_l; p.conv.dil = d->conv_in_dilation > 1 ? d->conv_in_dilation : 1;
    p.conv.up = d->conv_in_upsample > 1 ? d->conv_in_upsample : 1;
    p.conv.A2 = d->A2; p.conv.cin1 = d->A2 ? d->conv_cin1 : d->conv_cin;
    p.bias_row_div = d->bias_row_div; p.bias_ld = d->bias_ld > 0 ? d->bias_ld : d->N;
    const int batch = d->batch1 * d->batch2;
    hipStream_t stream = (hipStream_t)stream_;
    const char* name = d->name ? d->name : (d->conv_enabled ? "conv_igemm" : "gemm");
    bool narrow = d->N <= 64;
    if (!narrow) {
        // mid-size layers (e.g. 64x64 latents, N = 320): 128x128 tiles give < 1 workgroup per CU and waste the last n-tile;
        // 128x64 tiles double the workgroup count (3 resident per CU) -- latency hiding beats operand reuse there
        long long blocks128 = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch_of(d);
        if (blocks128 < 256 && getenv("DWG_GEMM_NO_NARROW") == nullptr) narrow = true;
    }
    if (d->dtype == DWG_DTYPE_HALF) {
        typedef HT T;
        int amode, bmode;
        long long ao[2] = {p.bA1, p.bA2}, bo[2] = {p.bB1, p.bB2};
        if (d->conv_enabled && !plain1x1) {
            if (d->conv_cin % 8 != 0 || ((uintptr_t)d->A % 16) != 0) return DWG_E_ARG;
            if (d->A2 && (d->conv_cin1 % 8 != 0 || d->conv_cin1 <= 0 || d->conv_cin1 >= d->conv_cin || ((uintptr_t)d->A2 % 16) != 0))
                return DWG_E_ARG;
            if (d->K != d->conv_kh * d->conv_kw * d->conv_cin) return DWG_E_ARG;
            amode = MODE_CONV;
        } else amode = pick_mode<T>(p.A, p.sam, p.sak, p.M, p.K, ao, 2);
        bmode = pick_mode<T>(p.B, p.sbn, p.sbk, p.N, p.K, bo, 2);
        const bool glds_ok = bmode == MODE_KVEC && (amode == MODE_KVEC || amode == MODE_CONV) && !d->force_register_staging;
        // LDS-patch convolution: the halo patch of a 64-channel slab is loaded once for all nine taps (2.3x less L2 -> LDS traffic
        // per flop than the im2col loader, which is what bounds these layers).  Below 64x64 latents it needs split-K over the
        // channel slabs to fill the chip; DWG_CONV_PATCH_MINM = smallest M it is used for (experiment switch).
        const int patch_min_m = getenv("DWG_CONV_PATCH_MINM") ? atoi(getenv("DWG_CONV_PATCH_MINM")) : 8192;
        static const int patch2 = getenv("DWG_CONV_PATCH2") ? atoi(getenv("DWG_CONV_PATCH2")) : 1;            // 0: the round-2 kernel
        static const int patch2_low = getenv("DWG_CONV_PATCH2_LOWRES") ? atoi(getenv("DWG_CONV_PATCH2_LOWRES")) : 1;
        const bool patch_geom = glds_ok && amode == MODE_CONV && batch == 1 && d->conv_kh == 3 && d->conv_kw == 3 &&
                                d->conv_stride == 1 && p.conv.dil == 1 && p.conv.up == 1 && d->conv_pad_t == 1 && d->conv_pad_l == 1 &&
                                !d->A2 && d->conv_cin % 64 == 0 && d->conv_hout == d->conv_hin && d->conv_wout == d->conv_win &&
                                d->conv_wout >= 16 && d->conv_hout >= 8 && getenv("DWG_CONV_NO_PATCH") == nullptr;
        bool patch_ok = patch_geom && d->M >= patch_min_m && (d->act == 0 || d->act == 3 || p.splitk > 1);
        // Round 6: the pipelined patch kernel on EIGHT waves (16 x 16-pixel tile x 128 columns) where whole 16 x 16 tiles cover the image and the
        // grid -- with split-K over the 64-channel slabs when it has to -- still gives >= 200 workgroups; this also brings the 32^2 / 16^2
        // levels of the denoiser (M < 8192, split-K) onto the patch kernel
        int p2_nw = 0, p2_sk = 1;
        if (patch2 && patch_geom && d->splitk <= 1 && d->conv_hout % 16 == 0 && d->conv_wout % 16 == 0 && (d->N % 128 == 0 || d->N >= 512) &&
            (d->M >= patch_min_m || patch2_low)) {
            const long long tiles8 = (long long)(d->M / 256) * ((d->N + 127) / 128);
            const int ncc = d->conv_cin / 64;
            long long sk = 1;
            if (tiles8 < 200 && d->workspace && d->splitk == 0 && (long long)d->M * d->N < (1LL << 33)) {
                sk = 256 / tiles8;
                if (sk > ncc / 2) sk = ncc / 2;
                while (sk > 1 && (size_t)sk * d->M * d->N * sizeof(float) > d->workspace_bytes) sk--;
                if (sk < 1) sk = 1;
                const long long per = (ncc + sk - 1) / sk;
                sk = (ncc + per - 1) / per;                          // no empty slices
            }
            const long long wgs = tiles8 * sk, rounds = (wgs + 255) / 256;
            static const int nw8 = getenv("DWG_CONV_PATCH2_NW8") ? atoi(getenv("DWG_CONV_PATCH2_NW8")) : 1;   // 0: never, 1: auto, 2: whenever legal
            // one workgroup per CU exposes every tile's prologue (first patch from HBM) and epilogue (128 KB of stores): with few steps per
            // tile (the VAE's 512^2 / 256^2 layers: 36 - 72) the four-wave kernel's two workgroups per CU overlap them; DWG_CONV_PATCH2_NW8_MINSTEPS
            static const int minsteps8 = getenv("DWG_CONV_PATCH2_NW8_MINSTEPS") ? atoi(getenv("DWG_CONV_PATCH2_NW8_MINSTEPS")) : 100;
            const long long steps = (long long)((ncc + sk - 1) / sk) * 9;
            const bool worth = nw8 == 2 || (nw8 == 1 && (sk > 1 || steps >= minsteps8 || d->M < patch_min_m));
            if (worth && wgs >= 200 && wgs * 100 >= rounds * 256 * 80 && (d->act == 0 || d->act == 3 || sk > 1)) { p2_nw = 8; p2_sk = (int)sk; patch_ok = true; }
        }
        if (patch_ok && !p2_nw && patch2) p2_nw = 4;
        if (patch_ok && p2_nw != 8 && p.splitk > 1) {
            // re-derive the slice count for this kernel's tiling: (8x16 pixel tiles) x (N / BN) workgroups, >= 2 slabs per slice
            const int gm = (d->M / (d->conv_hout * d->conv_wout)) * ((d->conv_hout + 7) / 8) * ((d->conv_wout + 15) / 16);
            const int blocks = gm * ((d->N + (narrow ? 64 : 128) - 1) / (narrow ? 64 : 128));
            int sk = blocks >= 256 ? 1 : (384 + blocks - 1) / blocks;
            const int ncc = d->conv_cin / 64;
            if (sk > ncc / 2) sk = ncc / 2;
            if (sk > p.splitk) sk = p.splitk;             // the workspace was sized for p.splitk slabs
            if (sk < 2) { sk = 1; p.ws = nullptr; }
            p.splitk = sk;
        }
        if (p2_nw == 8) { p.splitk = p2_sk; p.ws = p2_sk > 1 ? reinterpret_cast<float*>(d->workspace) : nullptr; }
        // the eight-wave tiles (round 6): long-K layers that do not take the LDS-patch kernel
        int big_sk = 1;
        int big_bm = (glds_ok && !patch_ok && d->splitk <= 1) ? big_tile(d->M, d->N, d->K, &big_sk) : 0;
        if (big_bm) {
            if (d->act == DWG_ACT_GEGLU_PAIR || !d->workspace || batch != 1 || d->splitk == 1 || (long long)d->M * d->N >= (1LL << 33)) big_sk = 1;
            while (big_sk > 1 && (size_t)big_sk * d->M * d->N * sizeof(float) > d->workspace_bytes) big_sk--;
            const int bbn = big_bm == 256 ? 128 : 256;
            const long long tiles = (long long)((d->M + big_bm - 1) / big_bm) * ((d->N + bbn - 1) / bbn) * batch;
            static const int force = getenv("DWG_GEMM_BIG") ? atoi(getenv("DWG_GEMM_BIG")) : 1;
            const long long wgs = tiles * big_sk, rounds = (wgs + 255) / 256;
            if (force < 2 && (wgs < 200 || wgs * 100 < rounds * 256 * 80)) big_bm = 0;      // (an unsplittable launch that would leave CUs idle)
        }
        if (patch_ok && p2_nw == 8) {
            launch_conv3x3_patch2<128, 8, 4>(p, stream, name);
        } else if (patch_ok && p2_nw == 4) {
            if (narrow) launch_conv3x3_patch2<64, 4, 4>(p, stream, name); else launch_conv3x3_patch2<128, 4, 2>(p, stream, name);
        } else if (patch_ok) {
            if (narrow) launch_conv3x3_patch<64>(p, stream, name); else launch_conv3x3_patch<128>(p, stream, name);
        } else if (glds_ok) {
            static const bool no_fast = getenv("DWG_CONV_NO_FAST") != nullptr;
            static const bool no_cat_fast = getenv("DWG_CONV_NO_CAT_FAST") != nullptr;
            const bool fast_geom = amode == MODE_CONV && d->conv_cin % 64 == 0 && p.conv.dil == 1 && p.conv.up == 1 && d->conv_kh * d->conv_kw <= 32 && !no_fast;
            const int akind = amode != MODE_CONV ? 0
                              : (fast_geom && !d->A2 ? 2
                                 : (fast_geom && d->A2 && d->conv_cin1 % 64 == 0 && (d->conv_cin - d->conv_cin1) % 64 == 0 && !no_cat_fast ? 3 : 1));
#ifdef DWG_GEMM_X_TU
            if (dbg && akind == 2 && ((big_bm == 256) || (!big_bm && narrow))) {         // timing experiments (no epilogue launch: garbage anyway)
                if (big_bm) { p.splitk = big_sk; p.ws = big_sk > 1 ? reinterpret_cast<float*>(d->workspace) : nullptr; }
                if (dbg == 1) launch_dbg<1>(p, big_bm != 0, stream); else if (dbg == 2) launch_dbg<2>(p, big_bm != 0, stream);
                else if (dbg == 3) launch_dbg<3>(p, big_bm != 0, stream); else if (dbg == 5) launch_dbg<5>(p, big_bm != 0, stream);
                else if (dbg == 6) launch_dbg<6>(p, big_bm != 0, stream); else if (dbg == 7) launch_dbg<7>(p, big_bm != 0, stream);
                else launch_dbg<0>(p, big_bm != 0, stream);
                return DWG_OK;
            }
#endif
            if (big_bm) {
                p.splitk = big_sk; p.ws = big_sk > 1 ? reinterpret_cast<float*>(d->workspace) : nullptr;
                if (akind == 3) launch_big<3>(p, big_bm, batch, stream, name);
                else if (akind == 2) launch_big<2>(p, big_bm, batch, stream, name);
                else if (akind == 1) launch_big<1>(p, big_bm, batch, stream, name);
                else launch_big<0>(p, big_bm, batch, stream, name);
            } else if (narrow) {
                if (akind == 3) launch_glds<64, 3>(p, batch, stream, name);
                else if (akind == 2) launch_glds<64, 2>(p, batch, stream, name);
                else if (akind == 1) launch_glds<64, 1>(p, batch, stream, name);
                else launch_glds<64, 0>(p, batch, stream, name);
            } else {
                if (akind == 3) launch_glds<128, 3>(p, batch, stream, name);
                else if (akind == 2) launch_glds<128, 2>(p, batch, stream, name);
                else if (akind == 1) launch_glds<128, 1>(p, batch, stream, name);
                else launch_glds<128, 0>(p, batch, stream, name);
            }
        }
#ifdef DWG_GEMM_X_TU
        else return DWG_E_ARG;     // split-precision operands exist only for the direct-to-LDS kernels (K-contiguous, 16-byte aligned rows)
#else
        else if (narrow) dispatch_a<T, 64>(p, amode, bmode, batch, stream, name);
        else dispatch_a<T, 128>(p, amode, bmode, batch, stream, name);
#endif
    } else {
#if defined(DWG_GEMM_F16_TU) || defined(DWG_GEMM_X_TU)
        return DWG_E_ARG;          // the exact-f32 kernels live in the bf16 unit
#else
        // exact-f32 path (v_mfma_f32_32x32x2_f32): the avatar's MLPs, and every layer of the fp32 denoiser / VAE plans -- the precision the
        // reference runs the 3DGS stage in (configs/__init__.py:236,241).  Register-staged generic kernel, incl. the im2col loader.
        typedef float T;
        long long ao[2] = {p.bA1, p.bA2}, bo[2] = {p.bB1, p.bB2};
        int amode;
        if (d->conv_enabled && !plain1x1) {
            if (d->conv_cin % 4 != 0 || ((uintptr_t)d->A % 16) != 0) return DWG_E_ARG;
            if (d->A2 && (d->conv_cin1 % 4 != 0 || d->conv_cin1 <= 0 || d->conv_cin1 >= d->conv_cin || ((uintptr_t)d->A2 % 16) != 0))
                return DWG_E_ARG;
            if (d->K != d->conv_kh * d->conv_kw * d->conv_cin) return DWG_E_ARG;
            amode = MODE_CONV;
        } else amode = pick_mode<T>(p.A, p.sam, p.sak, p.M, p.K, ao, 2);
        int bmode = pick_mode<T>(p.B, p.sbn, p.sbk, p.N, p.K, bo, 2);
        if (narrow) dispatch_a<T, 64>(p, amode, bmode, batch, stream, name);
        else dispatch_a<T, 128>(p, amode, bmode, batch, stream, name);
#endif
    }
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
