// probe: does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs, and does v_cvt_f16_f32 produce them?  (decides the lo-plane scaling of the
// split-precision "f32x" plans: an unscaled residual plane is subnormal for |x| < 0.125)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ void k(float a, float b, float* out, unsigned short* bits) {
    h8 A, B;
    _Float16 ha = (_Float16)a, hb = (_Float16)b;
    for (int e = 0; e < 8; e++) { A[e] = ha; B[e] = hb; }
    f16v c; for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)ha; out[2] = (float)ha * (float)hb; bits[0] = __builtin_bit_cast(unsigned short, ha); }
}
int main() {
    float* out; unsigned short* bits;
    hipMalloc(&out, 16); hipMalloc(&bits, 4);
    const float as[] = {1.0f, 9.5367431640625e-07f /*2^-20*/, 3.0517578125e-05f /*2^-15*/, 6.103515625e-05f /*2^-14 min normal*/, 5.9604644775390625e-08f /*2^-24*/};
    for (float a : as) {
        for (float b : {1.0f, 9.5367431640625e-07f}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, out, bits);
            float h[3]; unsigned short hb;
            hipMemcpy(h, out, 12, hipMemcpyDeviceToHost); hipMemcpy(&hb, bits, 2, hipMemcpyDeviceToHost);
            printf("a=%.6e b=%.6e  f16(a) bits=0x%04x back=%.6e  mfma=%.9e expected=%.9e ratio=%.6f\n", a, b, hb, h[1], h[0], 16.0 * h[2],
                   h[2] != 0 ? h[0] / (16.0 * h[2]) : -1.0);
        }
    }
    return 0;
}
