// lib/pointops operator surface that is not already covered by gdm_knn.hip / gdm_gather.hip:
// ball query, furthest point sampling, 3-point interpolation (forward / backward) and the label histograms, for gfx950.
//
// The reference's CUDA sources for these are absent (lib/pointops/setup.py:9-28 lists files that
// do not exist in the tree); only the Python wrapper survives, so the semantics follow the
// wrapper's contract (/root/reference/lib/pointops/functions/pointops.py:40-50 FPS: temp
// initialised to 1e10, idx int32 [b,m]; :205-219 ball query: idx int32 [b,m,nsample] zero
// initialised, "first nsample points with d2 < r2 per centre") and the PointNet++ convention the
// wrapper was written for: remaining slots repeat the first hit.
#include "gdm_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void ballquery_kernel(int n, int m, float r2, int nsample,
                                                        const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                        int32_t* __restrict__ idx)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const float* c = new_xyz + ((long)b * m + j) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float* p = xyz + (long)b * n * 3;
    int32_t* o = idx + ((long)b * m + j) * nsample;
    int cnt = 0;
    for (int k = 0; k < n && cnt < nsample; ++k) {
        const float dx = __fsub_rn(cx, p[3 * k]), dy = __fsub_rn(cy, p[3 * k + 1]), dz = __fsub_rn(cz, p[3 * k + 2]);
        float d2 = __fmul_rn(dx, dx);
        d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
        d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
        if (d2 < r2) {
            if (cnt == 0)
                for (int l = 0; l < nsample; ++l) o[l] = k;
            o[cnt] = k;
            ++cnt;
        }
    }
    if (cnt == 0)
        for (int l = 0; l < nsample; ++l) o[l] = 0;
}

constexpr int FPS_T = 1024;

__global__ __launch_bounds__(FPS_T) void fps_kernel(int n, int m, const float* __restrict__ xyz, float* __restrict__ temp,
                                                    int32_t* __restrict__ idx)
{
    __shared__ float sv[FPS_T / 64];
    __shared__ int si[FPS_T / 64];
    __shared__ int s_last;
    const int b = blockIdx.x;
    const float* p = xyz + (long)b * n * 3;
    float* t = temp + (long)b * n;
    int32_t* o = idx + (long)b * m;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += FPS_T) t[k] = 1e10f;
    if (tid == 0) {
        o[0] = 0;
        s_last = 0;
    }
    __syncthreads();
    for (int i = 1; i < m; ++i) {
        const int last = s_last;
        const float lx = p[3 * last], ly = p[3 * last + 1], lz = p[3 * last + 2];
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int k = tid; k < n; k += FPS_T) {
            const float dx = __fsub_rn(p[3 * k], lx), dy = __fsub_rn(p[3 * k + 1], ly), dz = __fsub_rn(p[3 * k + 2], lz);
            float d2 = __fmul_rn(dx, dx);
            d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
            d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
            const float v = fminf(t[k], d2);
            t[k] = v;
            if (v > bv) {          // ascending k per thread: first maximum
                bv = v;
                bi = k;
            }
        }
        for (int mk = 1; mk < 64; mk <<= 1) {
            const float ov = __shfl_xor(bv, mk, 64);
            const int oi = __shfl_xor(bi, mk, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            sv[tid >> 6] = bv;
            si[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            float v = sv[0];
            int ix = si[0];
            for (int w = 1; w < FPS_T / 64; ++w)
                if (sv[w] > v || (sv[w] == v && si[w] < ix)) {
                    v = sv[w];
                    ix = si[w];
                }
            o[i] = ix;
            s_last = ix;
        }
        __syncthreads();
    }
}

// out[b,c,j] = sum_k weight[b,j,k] * feat[b,c,idx[b,j,k]]  (pointops.py:114-131 interpolation_forward_cuda), k < 3
__global__ __launch_bounds__(256) void interp3_fwd_kernel(int c, int m, int n, const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ w, float* __restrict__ out)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int32_t* ij = idx + ((long)b * n + j) * 3;
    const float* wj = w + ((long)b * n + j) * 3;
    const float* f = feat + ((long)b * c + ch) * m;
    out[((long)b * c + ch) * n + j] = f[ij[0]] * wj[0] + f[ij[1]] * wj[1] + f[ij[2]] * wj[2];
}

// grad_feat[b,c,idx[b,j,k]] += weight[b,j,k] * grad_out[b,c,j]  (pointops.py:134-144 interpolation_backward_cuda)
__global__ __launch_bounds__(256) void interp3_bwd_kernel(int c, int m, int n, const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ w, float* __restrict__ gfeat)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int32_t* ij = idx + ((long)b * n + j) * 3;
    const float* wj = w + ((long)b * n + j) * 3;
    const float g = gout[((long)b * c + ch) * n + j];
    float* f = gfeat + ((long)b * c + ch) * m;
    atomicAdd(f + ij[0], g * wj[0]);
    atomicAdd(f + ij[1], g * wj[1]);
    atomicAdd(f + ij[2], g * wj[2]);
}

// Label histograms (pointops.py:289-366).  label_stat i32[b,n,nclass]; one thread per (b, centre j, class l).
//   MODE 0  labelstat_ballrange: sum over ALL points with d2 < r2
//   MODE 1  labelstat_idx:       sum over idx[b,j,0..nsample)
__global__ __launch_bounds__(256) void labelstat_kernel(int mode, int n, int m, float r2, int nsample, int nclass,
                                                        const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                        const int32_t* __restrict__ label_stat, const int32_t* __restrict__ idx,
                                                        int32_t* __restrict__ out)
{
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)m * nclass) return;
    const int j = (int)(e / nclass), l = (int)(e - (long)j * nclass);
    const int32_t* ls = label_stat + (long)b * n * nclass;
    int acc = 0;
    if (mode == 0) {
        const float* cxyz = new_xyz + ((long)b * m + j) * 3;
        const float cx = cxyz[0], cy = cxyz[1], cz = cxyz[2];
        const float* p = xyz + (long)b * n * 3;
        for (int k = 0; k < n; ++k) {
            const float dx = __fsub_rn(cx, p[3 * k]), dy = __fsub_rn(cy, p[3 * k + 1]), dz = __fsub_rn(cz, p[3 * k + 2]);
            float d2 = __fmul_rn(dx, dx);
            d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
            d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
            if (d2 < r2) acc += ls[(long)k * nclass + l];
        }
    } else {
        const int32_t* ij = idx + ((long)b * m + j) * nsample;
        for (int k = 0; k < nsample; ++k) acc += ls[(long)ij[k] * nclass + l];
    }
    out[((long)b * m + j) * nclass + l] = acc;
}

} // namespace

extern "C" int gdm_ballquery_hip(int B, int n, int m, float radius, int nsample,
                                 const float* new_xyz, const float* xyz, int32_t* idx, void* stream)
{
    GDM_CHECK_ARG(new_xyz && xyz && idx, "gdm_ballquery_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1 && m >= 1 && nsample >= 1, "gdm_ballquery_hip: bad shape");
    hipLaunchKernelGGL(ballquery_kernel, dim3(gdm_cdiv(m, 256), B), dim3(256), 0, (hipStream_t)stream,
                       n, m, radius * radius, nsample, new_xyz, xyz, idx);
    return gdm_launch_status("ballquery_kernel");
}

extern "C" int gdm_furthestsampling_hip(int B, int n, int m, const float* xyz, float* temp, int32_t* idx, void* stream)
{
    GDM_CHECK_ARG(xyz && temp && idx, "gdm_furthestsampling_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && n >= 1 && m >= 1 && m <= n, "gdm_furthestsampling_hip: bad shape B=%d n=%d m=%d", B, n, m);
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_T), 0, (hipStream_t)stream, n, m, xyz, temp, idx);
    return gdm_launch_status("fps_kernel");
}

extern "C" int gdm_interpolation_forward_hip(int B, int c, int m, int n, const float* feat, const int32_t* idx, const float* weight,
                                             float* out, void* stream)
{
    GDM_CHECK_ARG(feat && idx && weight && out, "gdm_interpolation_forward_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && c >= 1 && c <= 65535 && m >= 1 && n >= 1, "gdm_interpolation_forward_hip: bad shape");
    hipLaunchKernelGGL(interp3_fwd_kernel, dim3(gdm_cdiv(n, 256), c, B), dim3(256), 0, (hipStream_t)stream, c, m, n, feat, idx, weight, out);
    return gdm_launch_status("interp3_fwd_kernel");
}

// grad_feat must be zero-filled by the caller (as the reference wrapper does, pointops.py:141)
extern "C" int gdm_interpolation_backward_hip(int B, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                                              float* grad_feat, void* stream)
{
    GDM_CHECK_ARG(grad_out && idx && weight && grad_feat, "gdm_interpolation_backward_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && c >= 1 && c <= 65535 && m >= 1 && n >= 1, "gdm_interpolation_backward_hip: bad shape");
    hipLaunchKernelGGL(interp3_bwd_kernel, dim3(gdm_cdiv(n, 256), c, B), dim3(256), 0, (hipStream_t)stream, c, m, n, grad_out, idx, weight, grad_feat);
    return gdm_launch_status("interp3_bwd_kernel");
}

extern "C" int gdm_labelstat_ballrange_hip(int B, int n, int m, float radius, int nclass, const float* new_xyz, const float* xyz,
                                           const int32_t* label_stat, int32_t* new_label_stat, void* stream)
{
    GDM_CHECK_ARG(new_xyz && xyz && label_stat && new_label_stat, "gdm_labelstat_ballrange_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1 && m >= 1 && nclass >= 1, "gdm_labelstat_ballrange_hip: bad shape");
    hipLaunchKernelGGL(labelstat_kernel, dim3(gdm_cdiv((long)m * nclass, 256), B), dim3(256), 0, (hipStream_t)stream, 0, n, m, radius * radius, 0,
                       nclass, new_xyz, xyz, label_stat, (const int32_t*)nullptr, new_label_stat);
    return gdm_launch_status("labelstat_kernel");
}

extern "C" int gdm_labelstat_idx_hip(int B, int n, int m, int nsample, int nclass, const int32_t* label_stat, const int32_t* idx,
                                     int32_t* new_label_stat, void* stream)
{
    GDM_CHECK_ARG(label_stat && idx && new_label_stat, "gdm_labelstat_idx_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1 && m >= 1 && nsample >= 1 && nclass >= 1, "gdm_labelstat_idx_hip: bad shape");
    hipLaunchKernelGGL(labelstat_kernel, dim3(gdm_cdiv((long)m * nclass, 256), B), dim3(256), 0, (hipStream_t)stream, 1, n, m, 0.f, nsample,
                       nclass, (const float*)nullptr, (const float*)nullptr, label_stat, idx, new_label_stat);
    return gdm_launch_status("labelstat_kernel");
}
