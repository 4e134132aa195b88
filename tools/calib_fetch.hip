// calib_fetch.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for access patterns with a KNOWN byte count (round 5; verdict round 4,
// weak point 10).  MI355X_MICROARCH.md calibrates one pattern -- wide coalesced streaming reads (16 B / lane) are reported at HALF their
// bytes -- and says the others are uncalibrated; the rasterizer's kernels are 16-byte GATHERS, so the bench line carried a corrected and an
// uncorrected number for them.  Every kernel below reads from a 1 GiB buffer (beyond the 256 MiB Infinity Cache) exactly once per launch:
//   stream16   every lane one coalesced 16-byte load            useful bytes = n * 16
//   stream4    every lane one coalesced 4-byte load             useful bytes = n * 4
//   gather16   every lane one 16-byte load at a RANDOM row      useful bytes = n * 16, one 64-byte sector each = n * 64 moved
//   gather48   every lane three 16-byte loads of a RANDOM 48-byte row (the splat records' shape)   useful n * 48, sectors n * 64 .. 128
// Build + run (tools/calib_fetch.sh):  hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch && rocprofv3 --pmc FETCH_SIZE ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void stream16(const float4* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = x[i];
    if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = v.x;
}
__global__ void stream4(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    if (v == 12345.678f) out[0] = v;
}
__device__ __forceinline__ size_t rnd_row(size_t i, size_t rows) {       // a bijection-ish scatter of the lane index over the rows
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return (size_t)(h % rows);
}
__global__ void gather16(const float4* __restrict__ x, float* __restrict__ out, size_t n, size_t rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = x[rnd_row(i, rows)];
    if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = v.x;
}
__global__ void gather48(const float4* __restrict__ x, float* __restrict__ out, size_t n, size_t rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = rnd_row(i, rows / 3) * 3;
    const float4 a = x[r], b = x[r + 1], c = x[r + 2];
    if (a.x + b.y + c.z == 12345.678f) out[0] = a.x;
}

int main() {
    const size_t bytes = 1ull << 30, rows = bytes / 16;
    float4* x; float* out;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(x, 0, bytes);
    const size_t n16 = rows, n4 = bytes / 4 / 4, ng = 16u << 20;          // stream4 reads a quarter of the buffer
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, x, out, n16);
        hipLaunchKernelGGL(stream4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float*)x, out, n4);
        hipLaunchKernelGGL(gather16, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, 0, x, out, ng, rows);
        hipLaunchKernelGGL(gather48, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, 0, x, out, ng, rows);
    }
    hipDeviceSynchronize();
    printf("{\"stream16_useful\": %zu, \"stream4_useful\": %zu, \"gather16_useful\": %zu, \"gather16_sectors64\": %zu, \"gather48_useful\": %zu}\n",
           n16 * 16, n4 * 4, ng * 16, ng * 64, ng * 48);
    return 0;
}
