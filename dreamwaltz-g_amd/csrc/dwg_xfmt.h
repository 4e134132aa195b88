// dwg_xfmt.h -- the "f32x" activation / weight format of the split-precision plans (DWG_DTYPE_F32X, include/dwg_types.h).
//
// Why it exists.  The reference runs the guidance stage (VAE encoder inside autograd, ControlNet + UNet) in fp32
// (configs/__init__.py:236,241; scripts/train_w_expr.sh never passes --optim.fp16).  Exact-f32 MFMA peaks at 157 TFLOP/s on MI355X, the
// 16-bit MFMA at 2.5 PFLOP/s.  An fp32 value x is therefore kept as TWO fp16 halves
//        hi = fp16(x)  (round to nearest even),      lo = fp16((x - hi) * 2^11)          x ~= hi + lo * 2^-11
// (22 significand bits for 6.1e-5 <= |x| <= 65504, graceful below: fp16 subnormals are honoured by v_mfma_f32_32x32x16_f16 and by
// v_cvt_f16_f32 on gfx950 -- tools/probe_mfma_denorm.hip), and a product of two such numbers is formed from THREE 16-bit MFMAs with fp32
// accumulation:  a b ~= ah bh + 2^-11 (al bh + ah bl)   (the dropped al bl term is 2^-22 relative).
//
// Layout.  The split is done ONCE, by the producer of a tensor (GEMM epilogue, norm layer, input packer), not on every LDS fill of a
// consumer: a tensor of logical shape [rows, C] (C % 8 == 0, channels innermost) is stored in C*4 bytes per row -- exactly the bytes of
// the fp32 tensor -- as C/8 groups of 32 bytes: the eight hi halves of channels 8g .. 8g+7 (16 bytes), then their eight lo halves
// (16 bytes).  Every 16-byte chunk is an MFMA-ready fragment of eight consecutive k values, so the direct-to-LDS loaders of gemm.hip move
// the tensor unchanged ("a 2-byte tensor with 2C columns") and the k-loop carries no conversion instruction; channel slices at multiples
// of 8 (heads of 40 / 80 / 160, q | k | v column blocks, skip concatenations) stay valid sub-tensors.
// On the Python side such a tensor is an int32 tensor of the logical shape (dreamwaltz_g_amd/xfmt.py packs / unpacks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DWG_X_LO_SCALE 2048.0f
#define DWG_X_LO_INV 4.8828125e-4f      /* 2^-11 */
#define DWG_X_MAX 65504.0f

struct dwg_xs { uint32_t bits; };       // one LOGICAL element's worth of storage (4 bytes): pointer arithmetic in logical elements

typedef _Float16 dwg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dwg_h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dwg_x_split(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_fminf(__builtin_fmaxf(x, -DWG_X_MAX), DWG_X_MAX);      // saturate instead of inf (NaN propagates)
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * DWG_X_LO_SCALE);
}
__device__ __forceinline__ float dwg_x_join(_Float16 hi, _Float16 lo) { return __builtin_fmaf((float)lo, DWG_X_LO_INV, (float)hi); }

// eight channels (one 32-byte group); p must be 16-byte aligned and point at the group's first logical element
struct dwg_x8 {
    dwg_h8 hi, lo;
    __device__ __forceinline__ static dwg_x8 load(const dwg_xs* p) {
        dwg_x8 r;
        r.hi = *reinterpret_cast<const dwg_h8*>(p);
        r.lo = *reinterpret_cast<const dwg_h8*>(reinterpret_cast<const unsigned char*>(p) + 16);
        return r;
    }
    __device__ __forceinline__ void store(dwg_xs* p) const {
        *reinterpret_cast<dwg_h8*>(p) = hi;
        *reinterpret_cast<dwg_h8*>(reinterpret_cast<unsigned char*>(p) + 16) = lo;
    }
    __device__ __forceinline__ float get(int e) const { return dwg_x_join(hi[e], lo[e]); }
    __device__ __forceinline__ void set(int e, float v) { _Float16 h, l; dwg_x_split(v, h, l); hi[e] = h; lo[e] = l; }
};

// byte offset of logical element i (any i) inside a tensor whose rows start at multiples of 8 elements: hi half; the lo half is +16
__device__ __forceinline__ long long dwg_x_byte(long long i) { return ((i >> 3) << 5) + ((i & 7) << 1); }

__device__ __forceinline__ float dwg_x_get1(const void* base, long long i) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(base) + dwg_x_byte(i);
    return dwg_x_join(*reinterpret_cast<const _Float16*>(p), *reinterpret_cast<const _Float16*>(p + 16));
}
__device__ __forceinline__ void dwg_x_put1(void* base, long long i, float v) {
    unsigned char* p = reinterpret_cast<unsigned char*>(base) + dwg_x_byte(i);
    _Float16 h, l; dwg_x_split(v, h, l);
    *reinterpret_cast<_Float16*>(p) = h; *reinterpret_cast<_Float16*>(p + 16) = l;
}
// four consecutive elements i .. i+3, i % 4 == 0 (half a group): two 8-byte accesses
__device__ __forceinline__ void dwg_x_get4(const void* base, long long i, float (&v)[4]) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(base) + dwg_x_byte(i);
    const dwg_h4 h = *reinterpret_cast<const dwg_h4*>(p), l = *reinterpret_cast<const dwg_h4*>(p + 16);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = dwg_x_join(h[e], l[e]);
}
__device__ __forceinline__ void dwg_x_put4(void* base, long long i, const float (&v)[4]) {
    unsigned char* p = reinterpret_cast<unsigned char*>(base) + dwg_x_byte(i);
    dwg_h4 h, l;
#pragma unroll
    for (int e = 0; e < 4; e++) { _Float16 a, b; dwg_x_split(v[e], a, b); h[e] = a; l[e] = b; }
    *reinterpret_cast<dwg_h4*>(p) = h; *reinterpret_cast<dwg_h4*>(p + 16) = l;
}
