// Batched, masked sufficient statistics for the least-squares pose fit (Kabsch), gfx950.
//
// Replaces the per-thread `.cpu().numpy()` + `best_fit_transform` of the reference's pose solve
//   /root/reference/evaluator.py:85-100 (selected scene points, matched model vertices)
//   /root/reference/utils/pvn3d_eval_utils_kpls.py:43-77 (centroids, H = AA^T BB, SVD)
// One workgroup per crop streams the crop's points once (HBM-bound: 12 B/point scene xyz + 4 B index +
// 1 B mask + a 12-B gathered model vertex) and emits 16 doubles: n, sum A (3), sum B (3), sum A B^T (9),
// A = model vertex matched to the point, B = scene point.  The 3x3 SVDs (a few hundred flops per crop)
// stay on torch.linalg on the device.  fp64 accumulation: H = sum A B^T - n cA cB^T cancels heavily.
#include "gdm_common.h"

namespace {

__global__ __launch_bounds__(256) void kabsch_stats_kernel(const float* __restrict__ scene_xyz, long scene_bstride, int pt_stride,
                                                           int ch_stride, const float* __restrict__ model_xyz,
                                                           const int32_t* __restrict__ best_idx, const uint8_t* __restrict__ mask,
                                                           int N, int M, double* __restrict__ out)
{
    __shared__ double red[4][16];
    const int b = blockIdx.x;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    const float* sp = scene_xyz + (long)b * scene_bstride;
    for (int i = threadIdx.x; i < N; i += 256) {
        if (!mask[(long)b * N + i]) continue;
        int j = best_idx[(long)b * N + i];
        j = min(max(j, 0), M - 1);
        const double ax = model_xyz[3 * j], ay = model_xyz[3 * j + 1], az = model_xyz[3 * j + 2];
        const double bx = sp[(long)i * pt_stride], by = sp[(long)i * pt_stride + ch_stride], bz = sp[(long)i * pt_stride + 2 * ch_stride];
        acc[0] += 1.0;
        acc[1] += ax; acc[2] += ay; acc[3] += az;
        acc[4] += bx; acc[5] += by; acc[6] += bz;
        acc[7] += ax * bx; acc[8] += ax * by; acc[9] += ax * bz;
        acc[10] += ay * bx; acc[11] += ay * by; acc[12] += ay * bz;
        acc[13] += az * bx; acc[14] += az * by; acc[15] += az * bz;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[i] = v;
    }
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 16; ++i) red[threadIdx.x >> 6][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 16) out[(long)b * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

} // namespace

extern "C" int gdm_kabsch_stats_hip(const float* scene_xyz, long scene_bstride, int pt_stride, int ch_stride, const float* model_xyz,
                                    const int32_t* best_idx, const uint8_t* mask, int B, int N, int M, double* out, void* stream)
{
    GDM_CHECK_ARG(scene_xyz && model_xyz && best_idx && mask && out, "gdm_kabsch_stats_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && N >= 1 && M >= 1, "gdm_kabsch_stats_hip: bad shape");
    hipLaunchKernelGGL(kabsch_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scene_xyz, scene_bstride, pt_stride, ch_stride,
                       model_xyz, best_idx, mask, N, M, out);
    return gdm_launch_status("kabsch_stats_kernel");
}
