// Training-mode BatchNorm (+ ReLU / LeakyReLU) forward and backward for gfx950.
//
// Replaces, in the training step of the reference (train_lm.py:171-225), every  conv -> nn.BatchNorm{1,2}d -> activation  chain of the
// embedding network (models/pytorch_utils.py:70-124, models/RandLA/pytorch_utils.py:34-105, models/cnn/extractors.py:36-58): 75 layers,
// 150 MIOpen launches + the activation launches, 15 of 65 ms.  MIOpen's spatial kernels move these maps at 1.2 (forward) to
// 2.5 TB/s (backward); the passes here are plain streaming reductions / maps:
//   forward   bn_reduce<0>   per-channel sum x, sum x^2                          (1 read)
//             bn_fwd_apply   y = act(a x + b), a, b formed inline from the sums  (1 read, 1 write)  + running statistics, saved mean / rstd
//   backward  bn_reduce<1>   per-channel sum g', sum g' x,  g' = grad * act'(a x + b)            (2 reads)
//             bn_bwd_apply   gx = A g' + B x + C, coefficients inline from the sums              (2 reads, 1 write)  + grad weight / bias
// The activation is folded into both directions (its mask is recomputed from a x + b), so the separate activation forward / backward
// passes disappear as well.  Per-thread partial sums are fp32 over <= a few hundred values in four independent lanes, everything
// above that (wave, workgroup, grid) is accumulated in double, and mean / variance / coefficients are formed in double.  The grid-level
// step has no atomics and no memset: workgroup g of channel c stores its pair to sums[g][c], and the apply kernels add the G <= 32
// pairs of their channel themselves (lane-parallel, a few hundred cycles per workgroup) -- deterministic, two launches per direction.
// x f32[B, C, inner] contiguous, inner % 4 == 0.
#include "gdm_common.h"
#include <math.h>

namespace {

constexpr int BN_T = 256;

__device__ __forceinline__ float act_mask(float pre, float g, int act, float slope)
{
    if (act == 1) return pre > 0.f ? g : 0.f;
    if (act == 2) return pre > 0.f ? g : g * slope;
    return g;
}

// MODE 0: sums[g][c] = (sum x, sum x^2) of workgroup g's share;  MODE 1: (sum g', sum g' x).  sums[2CG] = number of elements per
// channel, sums[2CG+1] = G.
template <int MODE>
__global__ __launch_bounds__(BN_T) void bn_reduce_kernel(const float4* __restrict__ x, const float4* __restrict__ go, const float* __restrict__ saved,
                                                         int C, unsigned inner4, unsigned total4, int act, float slope, double* __restrict__ sums)
{
    __shared__ double part[BN_T / 64][2];
    const int c = blockIdx.y;
    float a = 1.f, b = 0.f;
    if (MODE == 1) {
        a = saved[c];
        b = saved[C + c];
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned j = blockIdx.x * BN_T + threadIdx.x; j < total4; j += gridDim.x * BN_T) {
        const unsigned bi = j / inner4, i = j - bi * inner4;
        const long off = ((long)bi * C + c) * inner4 + i;
        const float4 v = x[off];
        const float xv[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s0[k] += xv[k];
                s1[k] = fmaf(xv[k], xv[k], s1[k]);
            }
        } else {
            const float4 g = go[off];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gm = act_mask(fmaf(xv[k], a, b), gv[k], act, slope);
                s0[k] += gm;
                s1[k] = fmaf(gm, xv[k], s1[k]);
            }
        }
    }
    double d0 = ((double)s0[0] + (double)s0[1]) + ((double)s0[2] + (double)s0[3]);
    double d1 = ((double)s1[0] + (double)s1[1]) + ((double)s1[2] + (double)s1[3]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        d0 += __shfl_xor(d0, m, 64);
        d1 += __shfl_xor(d1, m, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        part[wave][0] = d0;
        part[wave][1] = d1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int w = 0; w < BN_T / 64; ++w) {
            t0 += part[w][0];
            t1 += part[w][1];
        }
        sums[((long)blockIdx.x * C + c) * 2] = t0;
        sums[((long)blockIdx.x * C + c) * 2 + 1] = t1;
        if (blockIdx.x == 0 && c == 0) {
            sums[2L * C * gridDim.x] = (double)total4 * 4.0;
            sums[2L * C * gridDim.x + 1] = (double)gridDim.x;
        }
    }
}

// channel c's totals from the G partial pairs (every wave computes them redundantly; G <= 64)
__device__ __forceinline__ void channel_sums(const double* __restrict__ sums, int C, int c, int G, double& t0, double& t1)
{
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (lane < G) {
        a = sums[((long)lane * C + c) * 2];
        b = sums[((long)lane * C + c) * 2 + 1];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
    }
    t0 = a;
    t1 = b;
}

// y = act(a x + b) with a = w rstd, b = bias - mean a from the (possibly all-reduced) sums; the b = 0 planes also store
// saved = (a | b | mean | rstd) for the backward and update the running statistics (momentum m, unbiased variance).
__global__ __launch_bounds__(BN_T) void bn_fwd_apply_kernel(const float4* __restrict__ x, const double* __restrict__ sums, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, int C, int G, long inner4, float eps, float momentum,
                                                            int act, float slope, float* __restrict__ saved, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float4* __restrict__ y)
{
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const double n = sums[2L * C * G];
    double t0, t1;
    channel_sums(sums, C, c, G, t0, t1);
    const double mean = t0 / n;
    const double var = fmax(t1 / n - mean * mean, 0.0);
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double ad = (double)weight[c] * rstd;
    const float a = (float)ad, b = (float)((double)bias[c] - mean * ad);
    if (blockIdx.x == 0 && plane < C && threadIdx.x == 0) {
        saved[c] = a;
        saved[C + c] = b;
        saved[2 * C + c] = (float)mean;
        saved[3 * C + c] = (float)rstd;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * (n > 1.0 ? var * n / (n - 1.0) : var));
        }
    }
    const float4* xp = x + plane * inner4;
    float4* yp = y + plane * inner4;
    for (long i = (long)blockIdx.x * BN_T + threadIdx.x; i < inner4; i += (long)gridDim.x * BN_T) {
        const float4 v = xp[i];
        float o[4] = {fmaf(v.x, a, b), fmaf(v.y, a, b), fmaf(v.z, a, b), fmaf(v.w, a, b)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (act == 1) o[k] = fmaxf(o[k], 0.f);
            if (act == 2) o[k] = o[k] > 0.f ? o[k] : o[k] * slope;
        }
        yp[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// gx = A g' + B x + Cc with S1 = sum g', S2 = rstd (sum g' x - mean S1):  A = w rstd,  B = -w rstd^2 S2 / n,  Cc = -B mean - w rstd S1 / n;
// grad weight = S2, grad bias = S1 (stored by the b = 0 planes).
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ go, const double* __restrict__ sums,
                                                            const float* __restrict__ weight, const float* __restrict__ saved, int C, int G, long inner4,
                                                            int act, float slope, float* __restrict__ gweight, float* __restrict__ gbias,
                                                            float4* __restrict__ gx)
{
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const float a = saved[c], b = saved[C + c];
    const double mean = (double)saved[2 * C + c], rstd = (double)saved[3 * C + c];
    const double n = sums[2L * C * G];
    double S1, S2x;
    channel_sums(sums, C, c, G, S1, S2x);
    const double S2 = rstd * (S2x - mean * S1);
    const double wr = (double)weight[c] * rstd;
    const double Bd = -wr * rstd * S2 / n;
    const float A = (float)wr, Bc = (float)Bd, Cc = (float)(-Bd * mean - wr * S1 / n);
    if (blockIdx.x == 0 && plane < C && threadIdx.x == 0) {
        gweight[c] = (float)S2;
        gbias[c] = (float)S1;
    }
    const float4* xp = x + plane * inner4;
    const float4* gp = go + plane * inner4;
    float4* op = gx + plane * inner4;
    for (long i = (long)blockIdx.x * BN_T + threadIdx.x; i < inner4; i += (long)gridDim.x * BN_T) {
        const float4 v = xp[i], g = gp[i];
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gm = act_mask(fmaf(xv[k], a, b), gv[k], act, slope);
            o[k] = fmaf(A, gm, fmaf(Bc, xv[k], Cc));
        }
        op[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

bool bn_shape_ok(const void* p0, const void* p1, const void* p2, int B, int C, long inner)
{
    return B >= 1 && C >= 1 && inner >= 4 && inner % 4 == 0 && (long)B * C <= 65535 && (long)B * (inner / 4) < 0x7fffffffL &&
           (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0;
}

int reduce_groups(int B, int C, long inner)
{
    // ~2048 workgroups over the chip, at least ~8 float4 per thread, at most 32 partial pairs per channel
    const long total4 = (long)B * (inner / 4);
    long g = (2048 + C - 1) / C;
    const long gmax = (total4 + BN_T * 8 - 1) / (BN_T * 8);
    if (g > gmax) g = gmax;
    if (g > 32) g = 32;
    if (g < 1) g = 1;
    return (int)g;
}

dim3 apply_grid(int B, int C, long inner)
{
    // ~8 float4 per thread (the per-workgroup prologue forms the channel's coefficients in double), but no fewer than ~1024 workgroups
    const long inner4 = inner / 4, planes = (long)B * C;
    long gx = inner4 / (BN_T * 8);
    if (gx * planes < 1024) gx = (1024 + planes - 1) / planes;
    const long gmax = (inner4 + BN_T - 1) / BN_T;
    if (gx > gmax) gx = gmax;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)planes);
}

} // namespace

extern "C" long gdm_bn_sums_len(int B, int C, long inner)
{
    if (B < 1 || C < 1 || inner < 4) return 0;
    return 2L * C * reduce_groups(B, C, inner) + 2;
}

extern "C" int gdm_bn_stats_hip(const float* x, int B, int C, long inner, double* sums, void* stream)
{
    GDM_CHECK_ARG(x && sums, "gdm_bn_stats_hip: NULL pointer");
    GDM_CHECK_ARG(bn_shape_ok(x, nullptr, nullptr, B, C, inner), "gdm_bn_stats_hip: B=%d C=%d inner=%ld (inner %% 4 == 0, B*C <= 65535, 16-byte aligned)", B, C, inner);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(reduce_groups(B, C, inner), C), dim3(BN_T), 0, s, (const float4*)x, (const float4*)nullptr, (const float*)nullptr, C,
                       (unsigned)(inner / 4), (unsigned)((long)B * (inner / 4)), 0, 0.f, sums);
    return gdm_launch_status("bn_reduce_kernel<0>");
}

extern "C" int gdm_bn_fwd_apply_hip(const float* x, const double* sums, int groups, const float* weight, const float* bias, int B, int C, long inner,
                                    float eps, float momentum, int act, float slope, float* saved, float* running_mean, float* running_var, float* y,
                                    void* stream)
{
    GDM_CHECK_ARG(x && sums && weight && bias && saved && y, "gdm_bn_fwd_apply_hip: NULL pointer");
    GDM_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gdm_bn_fwd_apply_hip: running_mean and running_var go together");
    GDM_CHECK_ARG(bn_shape_ok(x, y, nullptr, B, C, inner), "gdm_bn_fwd_apply_hip: B=%d C=%d inner=%ld (inner %% 4 == 0, B*C <= 65535, 16-byte aligned)", B, C, inner);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_bn_fwd_apply_hip: act=%d", act);
    GDM_CHECK_ARG(groups >= 0 && groups <= 64, "gdm_bn_fwd_apply_hip: groups=%d not in [0,64]", groups);
    hipLaunchKernelGGL(bn_fwd_apply_kernel, apply_grid(B, C, inner), dim3(BN_T), 0, (hipStream_t)stream, (const float4*)x, sums, weight, bias, C,
                       groups ? groups : reduce_groups(B, C, inner), inner / 4, eps, momentum, act, slope, saved, running_mean, running_var, (float4*)y);
    return gdm_launch_status("bn_fwd_apply_kernel");
}

extern "C" int gdm_bn_bwd_reduce_hip(const float* x, const float* grad_out, const float* saved, int B, int C, long inner, int act, float slope,
                                     double* sums, void* stream)
{
    GDM_CHECK_ARG(x && grad_out && saved && sums, "gdm_bn_bwd_reduce_hip: NULL pointer");
    GDM_CHECK_ARG(bn_shape_ok(x, grad_out, nullptr, B, C, inner), "gdm_bn_bwd_reduce_hip: B=%d C=%d inner=%ld (inner %% 4 == 0, B*C <= 65535, 16-byte aligned)", B, C, inner);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_bn_bwd_reduce_hip: act=%d", act);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(reduce_groups(B, C, inner), C), dim3(BN_T), 0, s, (const float4*)x, (const float4*)grad_out, saved, C,
                       (unsigned)(inner / 4), (unsigned)((long)B * (inner / 4)), act, slope, sums);
    return gdm_launch_status("bn_reduce_kernel<1>");
}

extern "C" int gdm_bn_bwd_apply_hip(const float* x, const float* grad_out, const double* sums, int groups, const float* weight, const float* saved,
                                    int B, int C, long inner, int act, float slope, float* grad_weight, float* grad_bias, float* grad_x, void* stream)
{
    GDM_CHECK_ARG(x && grad_out && sums && weight && saved && grad_weight && grad_bias && grad_x, "gdm_bn_bwd_apply_hip: NULL pointer");
    GDM_CHECK_ARG(bn_shape_ok(x, grad_out, grad_x, B, C, inner), "gdm_bn_bwd_apply_hip: B=%d C=%d inner=%ld (inner %% 4 == 0, B*C <= 65535, 16-byte aligned)", B, C, inner);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_bn_bwd_apply_hip: act=%d", act);
    GDM_CHECK_ARG(groups >= 0 && groups <= 64, "gdm_bn_bwd_apply_hip: groups=%d not in [0,64]", groups);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, apply_grid(B, C, inner), dim3(BN_T), 0, (hipStream_t)stream, (const float4*)x, (const float4*)grad_out, sums,
                       weight, saved, C, groups ? groups : reduce_groups(B, C, inner), inner / 4, act, slope, grad_weight, grad_bias, (float4*)grad_x);
    return gdm_launch_status("bn_bwd_apply_kernel");
}
