// Fused circle loss over the training similarity matrix [n_sel, M+1], gfx950.
//
// Replaces, for all selected points of a batch at once, the per-item Python loop of
//   /root/reference/models/geoMatch.py:55-83 (matching_loss: 3-D pdist radius test -> bool mask [n_i, M+1])
//   /root/reference/models/loss.py:470-494, 441-459 (CircleLoss.forward, masked log-sum-exp)
// which materialise ~8 [n_i, M+1] temporaries per item (mask, ~mask, ap, an, logit_p, logit_n, offsets).
// Here the positive mask is evaluated on the fly from (match_idx, model xyz, visible_flag, radius) and the
// two masked log-sum-exps are computed online in one pass over each row of `sim`; the backward pass
// recomputes mask and logits and writes d(loss_row)/d(sim) directly.  HBM-bound: forward reads sim once,
// backward reads sim once and writes dsim once.
//
// Row r (a selected scene point of batch item `item[r]`, ground-truth vertex `match[r]`, M = "none"):
//   mask[j] = match != M && vis[item][j] && sqrt(|xyz[match]-xyz[j]|^2 + 1e-7) < radius   (j < M)
//   mask[M] = match == M
//   ap = max(1 + m - s, 0) on mask, an = max(s + m, 0) off mask           (constants w.r.t. gradients)
//   logit_p = -ap (s - (1-m)) gamma,  logit_n = an (s - m) gamma
//   loss_row = softplus(LSE_mask(logit_p) + LSE_!mask(logit_n))            (an empty set gives LSE = -inf)
#include "gdm_common.h"
#include <math.h>

namespace {

constexpr int CL_BLOCK = 256;

struct RowCtx {
    float mx, my, mz;   // xyz of the ground-truth vertex
    bool none;          // match == M
};

__device__ __forceinline__ bool in_mask(const RowCtx& c, int j, int M, const float* __restrict__ xyz,
                                        const unsigned char* __restrict__ vis_row, float radius)
{
    if (j == M) return c.none;
    if (c.none || !vis_row[j]) return false;
    const float dx = c.mx - xyz[3 * j], dy = c.my - xyz[3 * j + 1], dz = c.mz - xyz[3 * j + 2];
    float d2 = __fmul_rn(dx, dx);
    d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
    d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
    return __fsqrt_rn(__fadd_rn(d2, 1e-7f)) < radius;              // utils/basic_utils.py:88-89
}

__device__ __forceinline__ void lse_push(float& mx, float& sm, float x)
{
    if (x > mx) {
        sm = sm * expf(mx - x) + 1.f;                               // mx == -inf: sm is 0, exp(-inf) = 0
        mx = x;
    } else {
        sm += expf(x - mx);
    }
}

__device__ __forceinline__ void lse_merge(float& mx, float& sm, float omx, float osm)
{
    if (osm == 0.f) return;
    if (sm == 0.f) {
        mx = omx;
        sm = osm;
        return;
    }
    const float m = fmaxf(mx, omx);
    sm = sm * expf(mx - m) + osm * expf(omx - m);
    mx = m;
}

__device__ __forceinline__ void block_lse(float& mx, float& sm)
{
    __shared__ float smx[CL_BLOCK / 64], ssm[CL_BLOCK / 64];
    for (int o = 1; o < 64; o <<= 1) {
        const float omx = __shfl_xor(mx, o, 64), osm = __shfl_xor(sm, o, 64);
        lse_merge(mx, sm, omx, osm);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        smx[threadIdx.x >> 6] = mx;
        ssm[threadIdx.x >> 6] = sm;
    }
    __syncthreads();
    mx = smx[0];
    sm = ssm[0];
    for (int w = 1; w < CL_BLOCK / 64; ++w) lse_merge(mx, sm, smx[w], ssm[w]);
}

__global__ __launch_bounds__(CL_BLOCK) void circle_rows_fwd_kernel(const float* __restrict__ sim, int Mp,
                                                                   const int32_t* __restrict__ match, const int32_t* __restrict__ item,
                                                                   const float* __restrict__ xyz, const unsigned char* __restrict__ vis,
                                                                   float radius, float gamma, float m,
                                                                   float* __restrict__ lse_p, float* __restrict__ lse_n,
                                                                   float* __restrict__ loss)
{
    const int r = blockIdx.x;
    const int M = Mp - 1;
    const int mt = match[r];
    RowCtx c;
    c.none = mt >= M || mt < 0;
    const int mc = c.none ? 0 : mt;
    c.mx = xyz[3 * mc]; c.my = xyz[3 * mc + 1]; c.mz = xyz[3 * mc + 2];
    const unsigned char* vr = vis + (long)item[r] * M;
    const float* s = sim + (long)r * Mp;
    float pmx = -INFINITY, psm = 0.f, nmx = -INFINITY, nsm = 0.f;
    for (int j = threadIdx.x; j < Mp; j += CL_BLOCK) {
        const float v = s[j];
        if (in_mask(c, j, M, xyz, vr, radius)) {
            const float ap = fmaxf(-v + 1.f + m, 0.f);
            lse_push(pmx, psm, -ap * (v - (1.f - m)) * gamma);
        } else {
            const float an = fmaxf(v + m, 0.f);
            lse_push(nmx, nsm, an * (v - m) * gamma);
        }
    }
    block_lse(pmx, psm);
    block_lse(nmx, nsm);
    if (threadIdx.x == 0) {
        const float lp = psm > 0.f ? pmx + logf(psm) : -INFINITY;
        const float ln = nsm > 0.f ? nmx + logf(nsm) : -INFINITY;
        const float z = lp + ln;
        lse_p[r] = lp;
        lse_n[r] = ln;
        loss[r] = z > 20.f ? z : log1pf(expf(z));                   // nn.Softplus(beta=1, threshold=20)
    }
}

__global__ __launch_bounds__(CL_BLOCK) void circle_rows_bwd_kernel(const float* __restrict__ sim, int Mp,
                                                                   const int32_t* __restrict__ match, const int32_t* __restrict__ item,
                                                                   const float* __restrict__ xyz, const unsigned char* __restrict__ vis,
                                                                   float radius, float gamma, float m,
                                                                   const float* __restrict__ lse_p, const float* __restrict__ lse_n,
                                                                   const float* __restrict__ grad_rows, float* __restrict__ dsim)
{
    const int r = blockIdx.x;
    const int M = Mp - 1;
    const int mt = match[r];
    RowCtx c;
    c.none = mt >= M || mt < 0;
    const int mc = c.none ? 0 : mt;
    c.mx = xyz[3 * mc]; c.my = xyz[3 * mc + 1]; c.mz = xyz[3 * mc + 2];
    const unsigned char* vr = vis + (long)item[r] * M;
    const float* s = sim + (long)r * Mp;
    float* d = dsim + (long)r * Mp;
    const float lp = lse_p[r], ln = lse_n[r];
    const float z = lp + ln;
    const float sig = z > 20.f ? 1.f : 1.f / (1.f + expf(-z));      // d softplus
    const float g = grad_rows[r] * sig;
    for (int j = threadIdx.x; j < Mp; j += CL_BLOCK) {
        const float v = s[j];
        float out;
        if (in_mask(c, j, M, xyz, vr, radius)) {
            const float ap = fmaxf(-v + 1.f + m, 0.f);
            const float lg = -ap * (v - (1.f - m)) * gamma;
            out = g * expf(lg - lp) * (-ap * gamma);
        } else {
            const float an = fmaxf(v + m, 0.f);
            const float lg = an * (v - m) * gamma;
            out = g * expf(lg - ln) * (an * gamma);
        }
        d[j] = isfinite(out) ? out : 0.f;                            // g == 0 with an empty set: 0 * inf
    }
}

} // namespace

extern "C" int gdm_circle_rows_fwd_hip(const float* sim, int R, int Mp, const int32_t* match, const int32_t* item,
                                       const float* xyz, const uint8_t* vis, float radius, float gamma, float m,
                                       float* lse_p, float* lse_n, float* loss, void* stream)
{
    GDM_CHECK_ARG(sim && match && item && xyz && vis && lse_p && lse_n && loss, "gdm_circle_rows_fwd_hip: NULL pointer");
    GDM_CHECK_ARG(R >= 1 && Mp >= 2, "gdm_circle_rows_fwd_hip: bad shape R=%d Mp=%d", R, Mp);
    hipLaunchKernelGGL(circle_rows_fwd_kernel, dim3(R), dim3(CL_BLOCK), 0, (hipStream_t)stream, sim, Mp, match, item, xyz, vis,
                       radius, gamma, m, lse_p, lse_n, loss);
    return gdm_launch_status("circle_rows_fwd_kernel");
}

extern "C" int gdm_circle_rows_bwd_hip(const float* sim, int R, int Mp, const int32_t* match, const int32_t* item,
                                       const float* xyz, const uint8_t* vis, float radius, float gamma, float m,
                                       const float* lse_p, const float* lse_n, const float* grad_rows, float* dsim, void* stream)
{
    GDM_CHECK_ARG(sim && match && item && xyz && vis && lse_p && lse_n && grad_rows && dsim, "gdm_circle_rows_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(R >= 1 && Mp >= 2, "gdm_circle_rows_bwd_hip: bad shape R=%d Mp=%d", R, Mp);
    hipLaunchKernelGGL(circle_rows_bwd_kernel, dim3(R), dim3(CL_BLOCK), 0, (hipStream_t)stream, sim, Mp, match, item, xyz, vis,
                       radius, gamma, m, lse_p, lse_n, grad_rows, dsim);
    return gdm_launch_status("circle_rows_bwd_kernel");
}
