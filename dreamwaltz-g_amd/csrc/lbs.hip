// lbs.hip -- SMPL-X linear-blend-skinning stage for gfx950.
//
//   k_joint_chain      one 64-lane workgroup: Rodrigues for all joints, kinematic chain through LDS,
//                      rest-pose removal and global translation  -> A[J,4,4]
//                      (replaces smplx.lbs.batch_rodrigues / batch_rigid_transform as called from
//                       /root/reference/core/human/inverse_lbs.py:688,696 and the compose at avatar.py:1441-1444)
//   k_blend_fwd / bwd  per-Gaussian blend T_i = sum_j w_ij A_j, point transform and the row-flipped quaternion
//                      path (inverse_lbs.py:190-242, avatar.py:1426-1462).  HBM-bound on the [N,J] weight rows
//                      (220 B/Gaussian at J=55): each wave streams its 64 rows with fully coalesced dword loads
//                      into LDS (row stride J is odd -> conflict-free per-lane row walks), A lives in LDS as
//                      broadcast reads.
//   k_vertex_transform per-vertex transform_V = compose(shape offset, pose offset, rigid blend, transl) applied to
//                      the mesh-bound vertex subset (inverse_lbs.py:652-717,758-772; avatar.py:1570-1576)
#include "dwg_common.h"
#include <mutex>
#include "dwg_prof_internal.h"
#include "lbs_math.h"
#include "../../include/dwg_lbs.h"

namespace {

#define MAXJ 64

// rest joints of the shaped template: J = J_template + (J_regressor . shapedirs) . shape -- 3J dot products of length S (400),
// ONE WAVE EACH across 3J/4 workgroups (lane-strided coalesced loads + a DPP reduction).  Inside the single-workgroup chain kernel
// the 165 dot products ran 42 deep per wave on cold lines: ~100 of its 125 us.  The result is parked in the (not yet written) last
// row of each A[j] so that no scratch buffer enters the C-ABI.
__global__ __launch_bounds__(256) void k_shaped_joints(int J, const float* __restrict__ joints, const float* __restrict__ jdirs,
                                                       const float* __restrict__ shape, int S, float* __restrict__ A) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= 3 * J) return;
    const float* d = jdirs + (size_t)idx * S;
    float v = 0.f;
    for (int l = lane; l < S; l += 64) v += d[l] * shape[l];
    v = dwg_wave_sum_to_lane63(v);
    if (lane == 63) A[16 * (idx / 3) + 12 + idx % 3] = joints[idx] + v;
}

__global__ __launch_bounds__(256) void k_joint_chain(int J, const float* __restrict__ pose /*[J,3]*/,
                                                    const float* __restrict__ joints /*[J,3]*/,
                                                    const int* __restrict__ parents, const float* __restrict__ transl,
                                                    int shaped /* rest joints come from k_shaped_joints (parked in A) */,
                                                    float* __restrict__ A /*[J,16]*/, float* __restrict__ rot_mats /*[J,9]|null*/) {
    __shared__ float tm[MAXJ][16];
    __shared__ float ch[MAXJ][16];
    __shared__ float sj[MAXJ][3];
    __shared__ int par[MAXJ];
    const int t = threadIdx.x;
    if (t < 3 * J) sj[t / 3][t % 3] = shaped ? A[16 * (t / 3) + 12 + t % 3] : joints[t];
    __syncthreads();
    if (t < J) {
        float R[9];
        const float rv[3] = {pose[3 * t], pose[3 * t + 1], pose[3 * t + 2]};
        dwg_rodrigues(rv, R);                // angle = |r + 1e-8| (smplx.lbs.batch_rodrigues)
        int p = parents[t];
        par[t] = p;
        float rel[3];
        for (int k = 0; k < 3; k++) rel[k] = sj[t][k] - (t > 0 ? sj[p][k] : 0.f);
        for (int r = 0; r < 3; r++) {
            for (int cc = 0; cc < 3; cc++) tm[t][4 * r + cc] = R[3 * r + cc];
            tm[t][4 * r + 3] = rel[r];
        }
        tm[t][12] = 0.f; tm[t][13] = 0.f; tm[t][14] = 0.f; tm[t][15] = 1.f;
        if (rot_mats) for (int k = 0; k < 9; k++) rot_mats[9 * t + k] = R[k];
    }
    __syncthreads();
    if (t < 16) ch[0][t] = tm[0][t];
    __syncthreads();
    for (int i = 1; i < J; i++) {
        if (t < 16) {
            int r = t >> 2, c = t & 3, p = par[i];
            ch[i][t] = ch[p][4 * r] * tm[i][c] + ch[p][4 * r + 1] * tm[i][4 + c] + ch[p][4 * r + 2] * tm[i][8 + c] +
                       ch[p][4 * r + 3] * tm[i][12 + c];
        }
        __syncthreads();
    }
    if (t < J) {
        float jx = sj[t][0], jy = sj[t][1], jz = sj[t][2];
        float o[16];
        for (int k = 0; k < 16; k++) o[k] = ch[t][k];
        for (int r = 0; r < 4; r++) o[4 * r + 3] -= ch[t][4 * r] * jx + ch[t][4 * r + 1] * jy + ch[t][4 * r + 2] * jz;
        if (transl) { o[3] += transl[0]; o[7] += transl[1]; o[11] += transl[2]; }  // compose(J_pose_rigid, G_transl_offset)
        for (int k = 0; k < 16; k++) A[16 * t + k] = o[k];
    }
}

// 256 threads = 4 waves; each wave owns 64 consecutive Gaussians and its own LDS slab of weight rows.
template <bool HAS_Q>
__global__ __launch_bounds__(256) void k_blend_fwd(int N, int J, int normalize, const float* __restrict__ A,
                                                   const float* __restrict__ w, const float* __restrict__ p,
                                                   const float* __restrict__ q, float* __restrict__ pout,
                                                   float* __restrict__ qout, float* __restrict__ T12save) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // [J][12]
    float* sW = smem + MAXJ * 12;     // [4 waves][64][J]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < J * 12; k += 256) { int j = k / 12, e = k - j * 12; sA[k] = A[16 * j + e]; }
    const int base = blockIdx.x * 256 + wave * 64;
    const int rows = min(64, N - base);
    float* myW = sW + wave * 64 * J;
    if (rows > 0) {
        const float* src = w + (size_t)base * J;
        for (int k = lane; k < rows * J; k += 64) myW[k] = src[k];
    }
    __syncthreads();
    const int i = base + lane;
    if (lane >= rows || rows <= 0) return;
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; e++) T[e] = 0.f;
    float wsum = 0.f;
    const float* row = myW + lane * J;
    for (int j = 0; j < J; j++) {
        float wj = row[j];
        wsum += wj;
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] += wj * sA[j * 12 + e];
    }
    if (normalize) {
        float inv = 1.f / wsum;
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] *= inv;
    }
    float pi[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}, po[3], qi[4], qo[4];
    if (HAS_Q) { qi[0] = q[4 * i]; qi[1] = q[4 * i + 1]; qi[2] = q[4 * i + 2]; qi[3] = q[4 * i + 3]; }
    dwg_lbs_apply(T, pi, HAS_Q ? qi : nullptr, po, qo);
    pout[3 * i] = po[0]; pout[3 * i + 1] = po[1]; pout[3 * i + 2] = po[2];
    if (HAS_Q) { qout[4 * i] = qo[0]; qout[4 * i + 1] = qo[1]; qout[4 * i + 2] = qo[2]; qout[4 * i + 3] = qo[3]; }
    if (T12save) {
        float4* dst = reinterpret_cast<float4*>(T12save + (size_t)i * 12);
        dst[0] = make_float4(T[0], T[1], T[2], T[3]); dst[1] = make_float4(T[4], T[5], T[6], T[7]);
        dst[2] = make_float4(T[8], T[9], T[10], T[11]);
    }
}

template <bool HAS_Q>
__global__ __launch_bounds__(256) void k_blend_bwd(int N, const float* __restrict__ T12, const float* __restrict__ p,
                                                   const float* __restrict__ q, const float* __restrict__ gpout,
                                                   const float* __restrict__ gqout, float* __restrict__ gp,
                                                   float* __restrict__ gq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4* src = reinterpret_cast<const float4*>(T12 + (size_t)i * 12);
    float4 a = src[0], b = src[1], c = src[2];
    float T[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    float pi[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    float go[3] = {gpout[3 * i], gpout[3 * i + 1], gpout[3 * i + 2]};
    float qi[4], gqo[4], gpi[3], gqi[4];
    if (HAS_Q) {
        for (int k = 0; k < 4; k++) { qi[k] = q[4 * i + k]; gqo[k] = gqout[4 * i + k]; }
    }
    dwg_lbs_apply_bwd(T, pi, HAS_Q ? qi : nullptr, go, HAS_Q ? gqo : nullptr, gpi, gqi, nullptr);
    gp[3 * i] = gpi[0]; gp[3 * i + 1] = gpi[1]; gp[3 * i + 2] = gpi[2];
    if (HAS_Q) { for (int k = 0; k < 4; k++) gq[4 * i + k] = gqi[k]; }
}

// transform_V for a vertex subset: out = T_rigid(v) * (x + shape_off(v) + pose_off(v)) [transl already in A].
// One wave per vertex; the blend-shape rows of the subset are gathered ONCE on the host side into vertex-major
// [Vp, 3, S] / [Vp, 3, F] arrays, so every lane streams contiguous memory and the three dot products are wave reductions.
__global__ __launch_bounds__(256) void k_vertex_transform(int Vp, int J, int S, int F, const float* __restrict__ x,
                                                          const float* __restrict__ A /*[J,16] incl. transl*/,
                                                          const float* __restrict__ w_sub /*[Vp,J]*/,
                                                          const float* __restrict__ sdirs /*[Vp,3,S] or null*/,
                                                          const float* __restrict__ shape /*[S]*/,
                                                          const float* __restrict__ pdirs /*[Vp,3,F] or null*/,
                                                          const float* __restrict__ rot_mats /*[J,9]*/,
                                                          float* __restrict__ out) {
    __shared__ float sA[MAXJ * 12];
    __shared__ float sfeat[(MAXJ - 1) * 9];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int k = tid; k < J * 12; k += 256) { int j = k / 12, e = k - j * 12; sA[k] = A[16 * j + e]; }
    for (int k = tid; k < F; k += 256) {
        int j = k / 9 + 1, e = k % 9;
        sfeat[k] = rot_mats[9 * j + e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    __syncthreads();
    const int t = blockIdx.x * 4 + (tid >> 6);
    if (t >= Vp) return;
    float o[3] = {0.f, 0.f, 0.f};
    if (sdirs) {
        const float* sd = sdirs + (size_t)t * 3 * S;
        for (int l = lane; l < S; l += 64) { float b = shape[l]; o[0] += sd[l] * b; o[1] += sd[S + l] * b; o[2] += sd[2 * S + l] * b; }
    }
    if (pdirs) {
        const float* pd = pdirs + (size_t)t * 3 * F;
        for (int f = lane; f < F; f += 64) { float sf = sfeat[f]; o[0] += pd[f] * sf; o[1] += pd[F + f] * sf; o[2] += pd[2 * F + f] * sf; }
    }
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; e++) T[e] = 0.f;
    const float* wr = w_sub + (size_t)t * J;
    for (int j = lane; j < J; j += 64) {
        float wj = wr[j];
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] += wj * sA[j * 12 + e];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = dwg_wave_sum_all(o[c]);
#pragma unroll
    for (int e = 0; e < 12; e++) T[e] = dwg_wave_sum_all(T[e]);
    if (lane == 0) {
        float px = x[3 * t] + o[0], py = x[3 * t + 1] + o[1], pz = x[3 * t + 2] + o[2];
        out[3 * t] = T[0] * px + T[1] * py + T[2] * pz + T[3];
        out[3 * t + 1] = T[4] * px + T[5] * py + T[6] * pz + T[7];
        out[3 * t + 2] = T[8] * px + T[9] * py + T[10] * pz + T[11];
    }
}

// Backward of k_vertex_transform w.r.t. the shape coefficients (and the translation column of A): for a fixed pose the
// transformed vertex is LINEAR in the coefficients,  v' = sum_j w_vj [Rg_j (x + S_v beta + P_v) + t_j(beta)],  so
//   g_shape[l] = sum_v  S_v[:, l] . (T_v[:3,:3]^T g_v)            (this kernel, first term)
//   g_At[j]    = sum_v  w_vj g_v                                  (this kernel; chained to beta by k_joint_chain_bwd)
// No float atomics (round 6; `learn_hand_betas` is ON in sub-stage 2.1 of the shipped recipe, scripts/train_w_expr.sh:66): a wave walks its
// vertices in a fixed order and every lane keeps ITS coefficients' and ITS joints' sums in registers (lane l owns coefficients l + 64 u and
// joints l + 64 q); the workgroup adds its four waves' sums in wave order and writes ONE row of partials; k_vertex_transform_bwd_reduce adds
// the workgroups' rows in workgroup order.  The same bits on every run.
#define MAXS_PER_LANE 8          // S <= 512 shape coefficients
#define VTB_ROW (64 * MAXS_PER_LANE + MAXJ * 3)
__global__ __launch_bounds__(256) void k_vertex_transform_bwd(int Vp, int J, int S, const float* __restrict__ A,
                                                              const float* __restrict__ w_sub, const float* __restrict__ sdirs,
                                                              const float* __restrict__ g_out /*[Vp,3]*/,
                                                              float* __restrict__ part /*[gridDim.x][VTB_ROW]*/) {
    __shared__ float sA[MAXJ * 12];
    __shared__ float sW[4][VTB_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < J * 12; k += 256) { int j = k / 12, e = k - j * 12; sA[k] = A[16 * j + e]; }
    __syncthreads();
    float acc[MAXS_PER_LANE], accT[3] = {0.f, 0.f, 0.f};       // (MAXJ == 64: one joint per lane)
    static_assert(MAXJ <= 64, "a lane owns one joint");
#pragma unroll
    for (int u = 0; u < MAXS_PER_LANE; u++) acc[u] = 0.f;
    for (int t = blockIdx.x * 4 + wave; t < Vp; t += gridDim.x * 4) {
        float R[9];
#pragma unroll
        for (int e = 0; e < 9; e++) R[e] = 0.f;
        const float* wr = w_sub + (size_t)t * J;
        const float g0 = g_out[3 * t], g1 = g_out[3 * t + 1], g2 = g_out[3 * t + 2];
        if (lane < J) {
            const float wj = wr[lane];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) R[3 * r + c] = wj * sA[lane * 12 + 4 * r + c];
            accT[0] += wj * g0; accT[1] += wj * g1; accT[2] += wj * g2;
        }
#pragma unroll
        for (int e = 0; e < 9; e++) R[e] = dwg_wave_sum_all(R[e]);
        const float gx0 = R[0] * g0 + R[3] * g1 + R[6] * g2, gx1 = R[1] * g0 + R[4] * g1 + R[7] * g2, gx2 = R[2] * g0 + R[5] * g1 + R[8] * g2;
        if (sdirs) {
            const float* sd = sdirs + (size_t)t * 3 * S;
#pragma unroll
            for (int u = 0; u < MAXS_PER_LANE; u++) {
                const int l = lane + 64 * u;
                if (l < S) acc[u] += sd[l] * gx0 + sd[S + l] * gx1 + sd[2 * S + l] * gx2;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXS_PER_LANE; u++) sW[wave][lane + 64 * u] = acc[u];
#pragma unroll
    for (int c = 0; c < 3; c++) sW[wave][64 * MAXS_PER_LANE + 3 * lane + c] = accT[c];
    __syncthreads();
    for (int k = tid; k < VTB_ROW; k += 256) part[(size_t)blockIdx.x * VTB_ROW + k] = ((sW[0][k] + sW[1][k]) + sW[2][k]) + sW[3][k];
}

// g_shape[l] = sum over the workgroups' partial rows, in workgroup order; likewise g_At.  (both OVERWRITTEN)
__global__ __launch_bounds__(256) void k_vertex_transform_bwd_reduce(int nblocks, int J, int S, const float* __restrict__ part,
                                                                     float* __restrict__ g_shape, float* __restrict__ g_At) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= S + 3 * J) return;
    const int col = k < S ? k : 64 * MAXS_PER_LANE + (k - S);
    float v = 0.f;
    for (int b = 0; b < nblocks; b++) v += part[(size_t)b * VTB_ROW + col];
    if (k < S) g_shape[k] = v; else g_At[k - S] = v;
}

// d loss / d A[:, :3, 3] -> d loss / d shape through the rest joints (one workgroup; the 55-joint chain is serial by nature and
// runs on one lane, the final [3J, S]^T matvec on all of them).
__global__ __launch_bounds__(256) void k_joint_chain_bwd(int J, const float* __restrict__ pose, const int* __restrict__ parents,
                                                         const float* __restrict__ jdirs /*[J,3,S]*/, int S,
                                                         const float* __restrict__ g_At /*[J,3]*/, float* __restrict__ g_shape) {
    __shared__ float Rg[MAXJ * 9];
    __shared__ float Gp[MAXJ * 3];
    __shared__ float dJ[MAXJ * 3];
    __shared__ int par[MAXJ];
    const int tid = threadIdx.x;
    if (tid < J) par[tid] = parents[tid];
    __syncthreads();
    if (tid == 0) dwg_joint_chain_rest_joint_bwd(J, pose, par, g_At, Rg, Gp, dJ);
    __syncthreads();
    for (int l = tid; l < S; l += 256) {
        float v = 0.f;
        for (int k = 0; k < 3 * J; k++) v += jdirs[(size_t)k * S + l] * dJ[k];
        g_shape[l] += v;
    }
}

}  // namespace

extern "C" {

int dwg_lbs_joint_chain(int32_t J, const float* pose, const float* joints, const int32_t* parents, const float* transl,
                        const float* joint_shape_dirs, const float* shape_coeffs, int32_t n_shape, float* A_out,
                        float* rot_mats_out, dwg_stream_t stream) {
    if (J <= 0 || J > MAXJ || !pose || !joints || !parents || !A_out) return DWG_E_ARG;
    if (joint_shape_dirs && (!shape_coeffs || n_shape <= 0)) return DWG_E_ARG;
    if (joint_shape_dirs)
        DWG_LAUNCH("lbs_shaped_joints", k_shaped_joints, dim3(dwg_cdiv(3 * J, 4)), dim3(256), 0, (hipStream_t)stream, J, joints,
                   joint_shape_dirs, shape_coeffs, n_shape, A_out);
    DWG_LAUNCH("lbs_joint_chain", k_joint_chain, dim3(1), dim3(256), 0, (hipStream_t)stream, J, pose, joints, parents, transl,
               joint_shape_dirs ? 1 : 0, A_out, rot_mats_out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_lbs_blend_forward(int32_t N, int32_t J, int32_t normalize_weights, const float* A, const float* weights,
                          const float* points, const float* quats, float* points_out, float* quats_out, float* T12_save,
                          dwg_stream_t stream) {
    if (N < 0 || J <= 0 || J > MAXJ) return DWG_E_ARG;
    if (N == 0) return DWG_OK;
    if (!A || !weights || !points || !points_out || (quats && !quats_out)) return DWG_E_ARG;
    size_t lds = (size_t)(MAXJ * 12 + 256 * J) * sizeof(float);
    dim3 grid(dwg_cdiv(N, 256)), block(256);
    if (quats)
        DWG_LAUNCH("lbs_blend_fwd", (k_blend_fwd<true>), grid, block, lds, (hipStream_t)stream, N, J, normalize_weights, A, weights,
                           points, quats, points_out, quats_out, T12_save);
    else
        DWG_LAUNCH("lbs_blend_fwd", (k_blend_fwd<false>), grid, block, lds, (hipStream_t)stream, N, J, normalize_weights, A, weights,
                           points, quats, points_out, quats_out, T12_save);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_lbs_blend_backward(int32_t N, const float* T12, const float* points, const float* quats, const float* g_points_out,
                           const float* g_quats_out, float* g_points, float* g_quats, dwg_stream_t stream) {
    if (N < 0) return DWG_E_ARG;
    if (N == 0) return DWG_OK;
    if (!T12 || !points || !g_points_out || !g_points || (quats && (!g_quats_out || !g_quats))) return DWG_E_ARG;
    dim3 grid(dwg_cdiv(N, 256)), block(256);
    if (quats)
        DWG_LAUNCH("lbs_blend_bwd", (k_blend_bwd<true>), grid, block, 0, (hipStream_t)stream, N, T12, points, quats, g_points_out,
                           g_quats_out, g_points, g_quats);
    else
        DWG_LAUNCH("lbs_blend_bwd", (k_blend_bwd<false>), grid, block, 0, (hipStream_t)stream, N, T12, points, quats, g_points_out,
                           g_quats_out, g_points, g_quats);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_lbs_vertex_transform(int32_t Vp, int32_t J, int32_t n_shape, int32_t n_posefeat, const float* vertex_coords,
                             const float* A, const float* lbs_weights_sub, const float* shapedirs_sub, const float* shape_coeffs,
                             const float* posedirs_sub, const float* rot_mats, float* out, dwg_stream_t stream) {
    if (Vp < 0 || J <= 0 || J > MAXJ || n_posefeat > (MAXJ - 1) * 9) return DWG_E_ARG;
    if (Vp == 0) return DWG_OK;
    if (!vertex_coords || !A || !lbs_weights_sub || !out) return DWG_E_ARG;
    if (shapedirs_sub && !shape_coeffs) return DWG_E_ARG;
    if (posedirs_sub && !rot_mats) return DWG_E_ARG;
    DWG_LAUNCH("lbs_vertex_transform", k_vertex_transform, dim3(dwg_cdiv(Vp, 4)), dim3(256), 0, (hipStream_t)stream, Vp, J, n_shape,
               posedirs_sub ? n_posefeat : 0, vertex_coords, A, lbs_weights_sub, shapedirs_sub, shape_coeffs, posedirs_sub, rot_mats,
               out);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

size_t dwg_lbs_vertex_transform_backward_shape_workspace_floats(int32_t Vp) {
    int blocks = dwg_cdiv(Vp > 0 ? Vp : 1, 4); if (blocks > 64) blocks = 64;
    return (size_t)blocks * VTB_ROW;
}

int dwg_lbs_vertex_transform_backward_shape_ws(int32_t Vp, int32_t J, int32_t n_shape, const float* A, const float* lbs_weights_sub,
                                               const float* shapedirs_sub, const float* g_out, const float* pose, const int32_t* parents,
                                               const float* joint_shape_dirs, float* g_A_transl_scratch, float* g_shape, float* workspace,
                                               dwg_stream_t stream_) {
    if (Vp < 0 || J <= 0 || J > MAXJ || n_shape <= 0 || n_shape > 64 * MAXS_PER_LANE) return DWG_E_ARG;
    if (!A || !lbs_weights_sub || !shapedirs_sub || !g_out || !pose || !parents || !joint_shape_dirs || !g_A_transl_scratch || !g_shape)
        return DWG_E_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (Vp == 0) {
        if (hipMemsetAsync(g_shape, 0, sizeof(float) * (size_t)n_shape, stream) != hipSuccess) return DWG_E_LAUNCH;
        return DWG_OK;
    }
    if (!workspace) return DWG_E_ARG;          // up to 64 workgroups, one row of partial sums each
    int blocks = dwg_cdiv(Vp, 4); if (blocks > 64) blocks = 64;
    DWG_LAUNCH("lbs_vertex_transform_bwd", k_vertex_transform_bwd, dim3(blocks), dim3(256), 0, stream, Vp, J, n_shape, A, lbs_weights_sub,
               shapedirs_sub, g_out, workspace);
    DWG_LAUNCH("lbs_vertex_transform_bwd_reduce", k_vertex_transform_bwd_reduce, dim3(dwg_cdiv(n_shape + 3 * J, 256)), dim3(256), 0, stream, blocks, J,
               n_shape, (const float*)workspace, g_shape, g_A_transl_scratch);
    DWG_LAUNCH("lbs_joint_chain_bwd", k_joint_chain_bwd, dim3(1), dim3(256), 0, stream, J, pose, parents, joint_shape_dirs, n_shape,
               (const float*)g_A_transl_scratch, g_shape);
    DWG_RETURN_IF_LAUNCH_FAILED();
    return DWG_OK;
}

int dwg_lbs_vertex_transform_backward_shape(int32_t Vp, int32_t J, int32_t n_shape, const float* A, const float* lbs_weights_sub,
                                            const float* shapedirs_sub, const float* g_out, const float* pose, const int32_t* parents,
                                            const float* joint_shape_dirs, float* g_A_transl_scratch, float* g_shape,
                                            dwg_stream_t stream_) {
    // The entry point without a workspace argument keeps ONE library-owned row block per device (64 rows, 180 KB): calls on different
    // streams of one device would share it, so they are serialised against each other by an event (this form exists for C callers of
    // the round-2 signature; the Python path passes its own workspace to the _ws form).
    static std::mutex mu;
    static float* ws_of[16] = {nullptr};
    static hipEvent_t done_of[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return DWG_E_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!ws_of[dev]) {
        if (hipMalloc(&ws_of[dev], sizeof(float) * 64 * VTB_ROW) != hipSuccess) return DWG_E_LAUNCH;
        if (hipEventCreateWithFlags(&done_of[dev], hipEventDisableTiming) != hipSuccess) return DWG_E_LAUNCH;
        hipEventRecord(done_of[dev], (hipStream_t)stream_);
    }
    if (hipStreamWaitEvent((hipStream_t)stream_, done_of[dev], 0) != hipSuccess) return DWG_E_LAUNCH;
    const int rc = dwg_lbs_vertex_transform_backward_shape_ws(Vp, J, n_shape, A, lbs_weights_sub, shapedirs_sub, g_out, pose, parents,
                                                               joint_shape_dirs, g_A_transl_scratch, g_shape, ws_of[dev], stream_);
    hipEventRecord(done_of[dev], (hipStream_t)stream_);
    return rc;
}

}  // extern "C"
