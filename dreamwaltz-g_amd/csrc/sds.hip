// sds.hip -- the latent algebra around the denoiser call of an SDS step (include/dwg_sds.h): posterior sample (+ backward), add_noise,
// classifier-free combination -> SDS gradient.  Each was 6 - 12 element-wise torch launches on 16 K-element tensors (launch latency, not
// bandwidth: ~45 launches per step between the VAE and the denoiser); here each is one grid-stride launch, 16 bytes per lane.
#include "dwg_common.h"
#include <cfloat>
#include "dwg_prof_internal.h"
#include "../../include/dwg_sds.h"

namespace {

__device__ __forceinline__ float clamp_logvar(float lv) { return fminf(fmaxf(lv, -30.f), 20.f); }

// one thread = four consecutive elements of one image's 4-channel latent (n % 4 == 0)
__global__ __launch_bounds__(256) void k_sds_posterior(int V, long long n4, const float4* __restrict__ moments, const float4* __restrict__ noise,
                                                       float scale, float4* __restrict__ latents) {
    const long long total = (long long)V * n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / n4, j = i - v * n4;
        const float4 m = moments[v * 2 * n4 + j], lv = moments[v * 2 * n4 + n4 + j], e = noise[i];
        float4 o;
        o.x = (m.x + expf(0.5f * clamp_logvar(lv.x)) * e.x) * scale; o.y = (m.y + expf(0.5f * clamp_logvar(lv.y)) * e.y) * scale;
        o.z = (m.z + expf(0.5f * clamp_logvar(lv.z)) * e.z) * scale; o.w = (m.w + expf(0.5f * clamp_logvar(lv.w)) * e.w) * scale;
        latents[i] = o;
    }
}

__device__ __forceinline__ float dlogvar(float lv, float e, float g, float scale) {
    return (lv >= -30.f && lv <= 20.f) ? g * scale * e * (0.5f * expf(0.5f * lv)) : 0.f;       // torch.clamp passes the gradient on [min, max]
}

__global__ __launch_bounds__(256) void k_sds_posterior_bwd(int V, long long n4, const float4* __restrict__ moments, const float4* __restrict__ noise,
                                                           float scale, const float4* __restrict__ g, float4* __restrict__ gm) {
    const long long total = (long long)V * n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / n4, j = i - v * n4;
        const float4 lv = moments[v * 2 * n4 + n4 + j], e = noise[i], gg = g[i];
        gm[v * 2 * n4 + j] = make_float4(gg.x * scale, gg.y * scale, gg.z * scale, gg.w * scale);
        gm[v * 2 * n4 + n4 + j] = make_float4(dlogvar(lv.x, e.x, gg.x, scale), dlogvar(lv.y, e.y, gg.y, scale), dlogvar(lv.z, e.z, gg.z, scale),
                                              dlogvar(lv.w, e.w, gg.w, scale));
    }
}

__device__ __forceinline__ float acp_at(const float* __restrict__ acp, int n_steps, const int64_t* __restrict__ t, long long v) {
    long long k = t[v];
    k = k < 0 ? 0 : (k >= n_steps ? n_steps - 1 : k);
    return acp[k];
}

__global__ __launch_bounds__(256) void k_sds_add_noise(int V, long long n4, const float4* __restrict__ lat, const float4* __restrict__ noise,
                                                       const float* __restrict__ acp, int n_steps, const int64_t* __restrict__ t,
                                                       float4* __restrict__ out) {
    const long long total = (long long)V * n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float a = acp_at(acp, n_steps, t, i / n4);
        const float sa = sqrtf(a), sb = sqrtf(1.f - a);
        const float4 x = lat[i], e = noise[i];
        out[i] = make_float4(sa * x.x + sb * e.x, sa * x.y + sb * e.y, sa * x.z + sb * e.z, sa * x.w + sb * e.w);
    }
}

__device__ __forceinline__ float nan_to_num_f(float v) {
    if (v != v) return 0.f;
    if (v > FLT_MAX) return FLT_MAX;
    if (v < -FLT_MAX) return -FLT_MAX;
    return v;
}

__global__ __launch_bounds__(256) void k_sds_gradient(int V, long long n4, const float4* __restrict__ eps, const float4* __restrict__ noise,
                                                      const float* __restrict__ acp, int n_steps, const int64_t* __restrict__ t, float s,
                                                      int weight_type, int ntn, float4* __restrict__ grad, float4* __restrict__ npred) {
    const long long total = (long long)V * n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float w = 1.f;
        if (weight_type != DWG_SDS_WEIGHT_NONE) {
            const float a = acp_at(acp, n_steps, t, i / n4);
            w = weight_type == DWG_SDS_WEIGHT_DREAMFUSION ? 1.f - a
                : (weight_type == DWG_SDS_WEIGHT_LATENT_NERF ? (1.f - a) * sqrtf(a) : sqrtf((1.f - a) / a));
        }
        const float4 u = eps[i], c = eps[total + i], e = noise[i];
        float4 p = make_float4(u.x + s * (c.x - u.x), u.y + s * (c.y - u.y), u.z + s * (c.z - u.z), u.w + s * (c.w - u.w));
        float4 g = make_float4(p.x - e.x, p.y - e.y, p.z - e.z, p.w - e.w);
        if (weight_type != DWG_SDS_WEIGHT_NONE) { g.x *= w; g.y *= w; g.z *= w; g.w *= w; }
        if (ntn) { g.x = nan_to_num_f(g.x); g.y = nan_to_num_f(g.y); g.z = nan_to_num_f(g.z); g.w = nan_to_num_f(g.w); }
        grad[i] = g;
        if (npred) npred[i] = p;
    }
}

inline int blocks_for(long long total) { long long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
inline bool bad16(const void* p) { return ((uintptr_t)p & 15) != 0; }

}  // namespace

extern "C" {

int dwg_sds_posterior_sample(int32_t V, int64_t n, const float* moments, const float* noise, float scale, float* latents, dwg_stream_t stream) {
    if (V < 0 || n < 0 || (n & 3)) return DWG_E_ARG;
    if (V == 0 || n == 0) return DWG_OK;
    if (!moments || !noise || !latents || bad16(moments) || bad16(noise) || bad16(latents)) return DWG_E_ARG;
    DWG_LAUNCH("sds_posterior", k_sds_posterior, dim3(blocks_for((long long)V * (n / 4))), dim3(256), 0, (hipStream_t)stream, V, (long long)(n / 4),
               (const float4*)moments, (const float4*)noise, scale, (float4*)latents);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_sds_posterior_sample_backward(int32_t V, int64_t n, const float* moments, const float* noise, float scale, const float* g_latents,
                                      float* g_moments, dwg_stream_t stream) {
    if (V < 0 || n < 0 || (n & 3)) return DWG_E_ARG;
    if (V == 0 || n == 0) return DWG_OK;
    if (!moments || !noise || !g_latents || !g_moments || bad16(moments) || bad16(noise) || bad16(g_latents) || bad16(g_moments)) return DWG_E_ARG;
    DWG_LAUNCH("sds_posterior_bwd", k_sds_posterior_bwd, dim3(blocks_for((long long)V * (n / 4))), dim3(256), 0, (hipStream_t)stream, V,
               (long long)(n / 4), (const float4*)moments, (const float4*)noise, scale, (const float4*)g_latents, (float4*)g_moments);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_sds_add_noise(int32_t V, int64_t n, const float* latents, const float* noise, const float* alphas_cumprod, int32_t n_steps,
                      const int64_t* timestep, float* out, dwg_stream_t stream) {
    if (V < 0 || n < 0 || (n & 3) || n_steps < 1) return DWG_E_ARG;
    if (V == 0 || n == 0) return DWG_OK;
    if (!latents || !noise || !alphas_cumprod || !timestep || !out || bad16(latents) || bad16(noise) || bad16(out)) return DWG_E_ARG;
    DWG_LAUNCH("sds_add_noise", k_sds_add_noise, dim3(blocks_for((long long)V * (n / 4))), dim3(256), 0, (hipStream_t)stream, V, (long long)(n / 4),
               (const float4*)latents, (const float4*)noise, alphas_cumprod, n_steps, timestep, (float4*)out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_sds_gradient(int32_t V, int64_t n, const float* eps, const float* noise, const float* alphas_cumprod, int32_t n_steps,
                     const int64_t* timestep, float guidance_scale, int32_t weight_type, int32_t nan_to_num, float* gradients, float* noise_pred,
                     dwg_stream_t stream) {
    if (V < 0 || n < 0 || (n & 3) || weight_type < 0 || weight_type > 3) return DWG_E_ARG;
    if (V == 0 || n == 0) return DWG_OK;
    if (!eps || !noise || !gradients || bad16(eps) || bad16(noise) || bad16(gradients) || (noise_pred && bad16(noise_pred))) return DWG_E_ARG;
    if (weight_type != DWG_SDS_WEIGHT_NONE && (!alphas_cumprod || !timestep || n_steps < 1)) return DWG_E_ARG;
    DWG_LAUNCH("sds_gradient", k_sds_gradient, dim3(blocks_for((long long)V * (n / 4))), dim3(256), 0, (hipStream_t)stream, V, (long long)(n / 4),
               (const float4*)eps, (const float4*)noise, alphas_cumprod, n_steps, timestep, guidance_scale, weight_type, nan_to_num, (float4*)gradients,
               (float4*)noise_pred);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
