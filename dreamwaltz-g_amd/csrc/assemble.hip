// assemble.hip -- per-Gaussian activations + non-rigid composition of DreamWaltzG.animate, one launch per direction
// (include/dwg_gaussian.h; reference avatar.py:1283-1290,1464-1498).  Pure streaming: ~100 B in, ~60 B out per Gaussian; the
// point is to replace ~45 element-wise PyTorch kernels (forward + autograd) by two.  One lane per Gaussian.
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "assemble_math.h"
#include "../../include/dwg_gaussian.h"

namespace {

__device__ __forceinline__ void ld3(const float* p, size_t i, float v[3]) { v[0] = p[3 * i]; v[1] = p[3 * i + 1]; v[2] = p[3 * i + 2]; }
__device__ __forceinline__ void st3(float* p, size_t i, const float v[3]) { p[3 * i] = v[0]; p[3 * i + 1] = v[1]; p[3 * i + 2] = v[2]; }

__global__ __launch_bounds__(256) void k_assemble_fwd(int n_free, int n_total, const float* __restrict__ positions,
                                                      const float* __restrict__ offsets, float init_offset,
                                                      const float* __restrict__ log_scales, const float* __restrict__ mlp_scales,
                                                      float init_scale, const float* __restrict__ quats, const float4* __restrict__ h,
                                                      float* __restrict__ pos_out, float* __restrict__ scl_out,
                                                      float4* __restrict__ q_out, float* __restrict__ col_out, float* __restrict__ op_out,
                                                      int mlp_ld /* row stride of offsets / mlp_scales in floats (3: dense) */) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_total) return;
    const bool free_row = i < (size_t)n_free;
    const float4 hv = h[i];
    const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
    float col[3], op;
    dwg_assemble_color(hh, free_row ? 0 : 1, col, &op);
    st3(col_out, i, col); op_out[i] = op;
    if (!free_row) return;
    float p[3], off[3], ls[3], ms[3], pos[3], scl[3], qn[4];
    ld3(positions, i, p); ld3(log_scales, i, ls);
    { const float* o = offsets + i * (size_t)mlp_ld; off[0] = o[0]; off[1] = o[1]; off[2] = o[2]; }
    { const float* m = mlp_scales + i * (size_t)mlp_ld; ms[0] = m[0]; ms[1] = m[1]; ms[2] = m[2]; }
    const float4 qv = reinterpret_cast<const float4*>(quats)[i];
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    dwg_assemble_geom(p, off, init_offset, ls, ms, init_scale, q, pos, scl, qn);
    st3(pos_out, i, pos); st3(scl_out, i, scl);
    q_out[i] = make_float4(qn[0], qn[1], qn[2], qn[3]);
}

__global__ __launch_bounds__(256) void k_assemble_bwd(int n_free, int n_total, float init_offset, const float* __restrict__ log_scales,
                                                      float init_scale, const float* __restrict__ quats, const float4* __restrict__ h,
                                                      const float* __restrict__ g_pos, const float* __restrict__ g_scl,
                                                      const float4* __restrict__ g_q, const float* __restrict__ g_col,
                                                      const float* __restrict__ g_op, float* __restrict__ d_pos,
                                                      float* __restrict__ d_off, float* __restrict__ d_ls, float* __restrict__ d_ms,
                                                      float4* __restrict__ d_q, float4* __restrict__ d_h, int mlp_ld, int mlp_tail) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_total) return;
    const bool free_row = i < (size_t)n_free;
    const float4 hv = h[i];
    const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
    float gc[3] = {0.f, 0.f, 0.f}, dh[4];
    if (g_col) ld3(g_col, i, gc);
    dwg_assemble_color_bwd(hh, free_row ? 0 : 1, gc, g_op ? g_op[i] : 0.f, dh);
    d_h[i] = make_float4(dh[0], dh[1], dh[2], dh[3]);
    if (!free_row) return;
    float ls[3], gp[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    ld3(log_scales, i, ls);
    if (g_pos) ld3(g_pos, i, gp);
    if (g_scl) ld3(g_scl, i, gs);
    if (g_q) { const float4 t = g_q[i]; gq[0] = t.x; gq[1] = t.y; gq[2] = t.z; gq[3] = t.w; }
    const float4 qv = reinterpret_cast<const float4*>(quats)[i];
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    float dp[3], doff[3], dls[3], dms[3], dq[4];
    dwg_assemble_geom_bwd(ls, q, init_offset, init_scale, gp, gs, gq, dp, doff, dls, dms, dq);
    st3(d_pos, i, dp); st3(d_ls, i, dls);
    {   // packed form: d_off / d_ms are columns of ONE [n_free, mlp_ld] gradient whose remaining mlp_tail columns (behind d_ms) get zeros here
        float* o = d_off + i * (size_t)mlp_ld; o[0] = doff[0]; o[1] = doff[1]; o[2] = doff[2];
        float* m = d_ms + i * (size_t)mlp_ld; m[0] = dms[0]; m[1] = dms[1]; m[2] = dms[2];
        for (int k = 0; k < mlp_tail; k++) m[3 + k] = 0.f;
    }
    d_q[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
}

struct SegJobs { dwg_segment j[DWG_MAX_SEGMENTS]; };

// job blockIdx.y: count floats from src to dst; div != 0: out = (in + add) * (1 / div) -- the grid encoder's input normalisation (gridencoder
// grid.py `(inputs + bound) / (2 * bound)`) applied while the canonical positions are gathered, in the form torch's elementwise kernels give a
// division by a scalar (one reciprocal in fp32, then a multiply per element): the same bits as the reference's two launches
__global__ __launch_bounds__(256) void k_copy_segments(SegJobs J, float add, float div) {
    const float inv = div != 0.f ? 1.f / div : 0.f;
    const dwg_segment s = J.j[blockIdx.y];
    const float* __restrict__ src = reinterpret_cast<const float*>(s.src);
    float* __restrict__ dst = reinterpret_cast<float*>(s.dst);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.count; i += (long long)gridDim.x * 256) {
        const float v = src[i];
        dst[i] = div != 0.f ? (v + add) * inv : v;
    }
}

// job blockIdx.y: dst[i] += src[i] (the gradient of a packed tensor handed back to the pieces it was gathered from, added into their slices)
__global__ __launch_bounds__(256) void k_add_segments(SegJobs J) {
    const dwg_segment s = J.j[blockIdx.y];
    const float* __restrict__ src = reinterpret_cast<const float*>(s.src);
    float* __restrict__ dst = reinterpret_cast<float*>(s.dst);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.count; i += (long long)gridDim.x * 256) dst[i] += src[i];
}

}  // namespace

extern "C" {

int dwg_gaussian_assemble_forward_ld(int32_t n_free, int32_t n_total, const float* positions, const float* offsets, float init_offset,
                                     const float* log_scales, const float* mlp_scales, int32_t mlp_ld, float init_scale, const float* quaternions,
                                     const float* h, float* pos_out, float* scales_out, float* quats_out, float* colors_out,
                                     float* opac_out, dwg_stream_t stream) {
    if (n_free < 0 || n_total < n_free || mlp_ld < 3) return DWG_E_ARG;
    if (n_total == 0) return DWG_OK;
    if (!h || !colors_out || !opac_out) return DWG_E_ARG;
    if (n_free > 0 && (!positions || !offsets || !log_scales || !mlp_scales || !quaternions || !pos_out || !scales_out || !quats_out))
        return DWG_E_ARG;
    if (((uintptr_t)h | (uintptr_t)quaternions | (uintptr_t)quats_out) & 15) return DWG_E_ARG;
    DWG_LAUNCH("gaussian_assemble_fwd", k_assemble_fwd, dim3(dwg_cdiv(n_total, 256)), dim3(256), 0, (hipStream_t)stream, n_free, n_total,
               positions, offsets, init_offset, log_scales, mlp_scales, init_scale, quaternions, (const float4*)h, pos_out, scales_out,
               (float4*)quats_out, colors_out, opac_out, mlp_ld);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_gaussian_assemble_forward(int32_t n_free, int32_t n_total, const float* positions, const float* offsets, float init_offset,
                                  const float* log_scales, const float* mlp_scales, float init_scale, const float* quaternions,
                                  const float* h, float* pos_out, float* scales_out, float* quats_out, float* colors_out,
                                  float* opac_out, dwg_stream_t stream) {
    return dwg_gaussian_assemble_forward_ld(n_free, n_total, positions, offsets, init_offset, log_scales, mlp_scales, 3, init_scale, quaternions, h,
                                            pos_out, scales_out, quats_out, colors_out, opac_out, stream);
}

int dwg_gaussian_assemble_backward_ld(int32_t n_free, int32_t n_total, float init_offset, const float* log_scales, float init_scale,
                                      const float* quaternions, const float* h, const float* g_pos, const float* g_scales,
                                      const float* g_quats, const float* g_colors, const float* g_opac, float* d_positions,
                                      float* d_offsets, float* d_log_scales, float* d_mlp_scales, int32_t mlp_ld, int32_t mlp_tail,
                                      float* d_quaternions, float* d_h, dwg_stream_t stream) {
    if (n_free < 0 || n_total < n_free || mlp_ld < 3 || mlp_tail < 0) return DWG_E_ARG;
    if (n_total == 0) return DWG_OK;
    if (!h || !d_h) return DWG_E_ARG;
    if (n_free > 0 && (!log_scales || !quaternions || !d_positions || !d_offsets || !d_log_scales || !d_mlp_scales || !d_quaternions))
        return DWG_E_ARG;
    if (((uintptr_t)h | (uintptr_t)d_h | (uintptr_t)quaternions | (uintptr_t)d_quaternions | (uintptr_t)g_quats) & 15) return DWG_E_ARG;
    DWG_LAUNCH("gaussian_assemble_bwd", k_assemble_bwd, dim3(dwg_cdiv(n_total, 256)), dim3(256), 0, (hipStream_t)stream, n_free, n_total,
               init_offset, log_scales, init_scale, quaternions, (const float4*)h, g_pos, g_scales, (const float4*)g_quats, g_colors,
               g_opac, d_positions, d_offsets, d_log_scales, d_mlp_scales, (float4*)d_quaternions, (float4*)d_h, mlp_ld, mlp_tail);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_gaussian_assemble_backward(int32_t n_free, int32_t n_total, float init_offset, const float* log_scales, float init_scale,
                                   const float* quaternions, const float* h, const float* g_pos, const float* g_scales,
                                   const float* g_quats, const float* g_colors, const float* g_opac, float* d_positions,
                                   float* d_offsets, float* d_log_scales, float* d_mlp_scales, float* d_quaternions, float* d_h,
                                   dwg_stream_t stream) {
    return dwg_gaussian_assemble_backward_ld(n_free, n_total, init_offset, log_scales, init_scale, quaternions, h, g_pos, g_scales, g_quats, g_colors,
                                             g_opac, d_positions, d_offsets, d_log_scales, d_mlp_scales, 3, 0, d_quaternions, d_h, stream);
}

// ---- several row-block copies in ONE launch (the merges of DreamWaltzG.animate: gaussian_utils.py:56-68 without one cat per tensor) ----
int dwg_copy_segments(int32_t count, const dwg_segment* segs, float add, float div, dwg_stream_t stream) {
    if (count < 0 || count > DWG_MAX_SEGMENTS || (count > 0 && !segs)) return DWG_E_ARG;
    if (count == 0) return DWG_OK;
    SegJobs J;
    long long longest = 0;
    for (int i = 0; i < count; i++) {
        if (segs[i].count < 0 || (segs[i].count > 0 && (!segs[i].dst || !segs[i].src))) return DWG_E_ARG;
        J.j[i] = segs[i];
        longest = segs[i].count > longest ? segs[i].count : longest;
    }
    if (longest == 0) return DWG_OK;
    int blocks = (int)((longest + 1023) / 1024); if (blocks > 1024) blocks = 1024;
    DWG_LAUNCH("copy_segments", k_copy_segments, dim3(blocks, count), dim3(256), 0, (hipStream_t)stream, J, add, div);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_add_segments(int32_t count, const dwg_segment* segs, dwg_stream_t stream) {
    if (count < 0 || count > DWG_MAX_SEGMENTS || (count > 0 && !segs)) return DWG_E_ARG;
    SegJobs J;
    long long longest = 0;
    for (int i = 0; i < count; i++) {
        if (segs[i].count < 0 || (segs[i].count > 0 && (!segs[i].dst || !segs[i].src))) return DWG_E_ARG;
        J.j[i] = segs[i];
        longest = segs[i].count > longest ? segs[i].count : longest;
    }
    if (longest == 0) return DWG_OK;
    int blocks = (int)((longest + 1023) / 1024); if (blocks > 1024) blocks = 1024;
    DWG_LAUNCH("add_segments", k_add_segments, dim3(blocks, count), dim3(256), 0, (hipStream_t)stream, J);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

}  // extern "C"
