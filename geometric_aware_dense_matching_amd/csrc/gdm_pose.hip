// Batched, masked sufficient statistics for the least-squares pose fit (Kabsch), gfx950.
//
// Replaces the per-thread `.cpu().numpy()` + `best_fit_transform` of the reference's pose solve
//   /root/reference/evaluator.py:85-100 (selected scene points, matched model vertices)
//   /root/reference/utils/pvn3d_eval_utils_kpls.py:43-77 (centroids, H = AA^T BB, SVD)
// One workgroup per crop streams the crop's points once (HBM-bound: 12 B/point scene xyz + 4 B index +
// 1 B mask + a 12-B gathered model vertex) and emits 16 doubles: n, sum A (3), sum B (3), sum A B^T (9),
// A = model vertex matched to the point, B = scene point.  fp64 accumulation: H = sum A B^T - n cA cB^T cancels heavily.
// kabsch_solve_kernel turns the 16 statistics into [R|t] on the device, one lane per crop (no library SVD, no host
// synchronisation, so the whole step can be captured in a hipGraph).
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

// The proper rotation maximising trace(R H), H = sum (a - cA)(b - cB)^T, is what Kabsch-with-reflection-fix returns
// (pvn3d_eval_utils_kpls.py:60-70: R = V U^T, last row of V^T negated when det < 0).  It is computed here as Horn's unit
// quaternion: the eigenvector of the largest eigenvalue of the symmetric 4x4 matrix built from H, by cyclic Jacobi in fp64.
__global__ __launch_bounds__(64) void kabsch_solve_kernel(const double* __restrict__ stats, int B, int min_points,
                                                          float* __restrict__ RT, uint8_t* __restrict__ valid)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double* st = stats + (long)b * 16;
    const double n = st[0];
    float* o = RT + (long)b * 12;
    if (!(n >= (double)min_points)) {                                   // evaluator.py:94-96 sentinel pose
        o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        o[4] = 0.f; o[5] = 1.f; o[6] = 0.f; o[7] = 0.f;
        o[8] = 0.f; o[9] = 0.f; o[10] = 1.f; o[11] = -1000.f;
        valid[b] = 0;
        return;
    }
    double cA[3], cB[3], S[3][3];
    for (int i = 0; i < 3; ++i) { cA[i] = st[1 + i] / n; cB[i] = st[4 + i] / n; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i][j] = st[7 + 3 * i + j] - n * cA[i] * cB[j];
    double A[4][4], V[4][4];
    A[0][0] = S[0][0] + S[1][1] + S[2][2];
    A[0][1] = S[1][2] - S[2][1]; A[0][2] = S[2][0] - S[0][2]; A[0][3] = S[0][1] - S[1][0];
    A[1][1] = S[0][0] - S[1][1] - S[2][2];
    A[1][2] = S[0][1] + S[1][0]; A[1][3] = S[2][0] + S[0][2];
    A[2][2] = -S[0][0] + S[1][1] - S[2][2];
    A[2][3] = S[1][2] + S[2][1];
    A[3][3] = -S[0][0] - S[1][1] + S[2][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { if (j < i) A[i][j] = A[j][i]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    double scale = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) scale = fmax(scale, fabs(A[i][j]));
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) off = fmax(off, fabs(A[p][q]));
        if (off <= 1e-18 * scale) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) <= 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {                            // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {                            // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[best][best]) best = i;
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    const double nq = sqrt(w * w + x * x + y * y + z * z);
    if (nq > 0.0) { w /= nq; x /= nq; y /= nq; z /= nq; } else { w = 1.0; x = y = z = 0.0; }
    double R[3][3];
    R[0][0] = 1.0 - 2.0 * (y * y + z * z); R[0][1] = 2.0 * (x * y - w * z); R[0][2] = 2.0 * (x * z + w * y);
    R[1][0] = 2.0 * (x * y + w * z); R[1][1] = 1.0 - 2.0 * (x * x + z * z); R[1][2] = 2.0 * (y * z - w * x);
    R[2][0] = 2.0 * (x * z - w * y); R[2][1] = 2.0 * (y * z + w * x); R[2][2] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i) {
        double t = cB[i];
        for (int j = 0; j < 3; ++j) { o[4 * i + j] = (float)R[i][j]; t -= R[i][j] * cA[j]; }
        o[4 * i + 3] = (float)t;
    }
    valid[b] = 1;
}

} // namespace

extern "C" int gdm_kabsch_solve_hip(const double* stats, int B, int min_points, float* RT, uint8_t* valid, void* stream)
{
    GDM_CHECK_ARG(stats && RT && valid, "gdm_kabsch_solve_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1, "gdm_kabsch_solve_hip: bad shape");
    hipLaunchKernelGGL(kabsch_solve_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, B, min_points, RT, valid);
    return gdm_launch_status("kabsch_solve_kernel");
}

extern "C" int gdm_kabsch_stats_hip(const float* scene_xyz, long scene_bstride, int pt_stride, int ch_stride, const float* model_xyz,
                                    const int32_t* best_idx, const uint8_t* mask, int B, int N, int M, double* out, void* stream)
{
    GDM_CHECK_ARG(scene_xyz && model_xyz && best_idx && mask && out, "gdm_kabsch_stats_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && N >= 1 && M >= 1, "gdm_kabsch_stats_hip: bad shape");
    hipLaunchKernelGGL(kabsch_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scene_xyz, scene_bstride, pt_stride, ch_stride,
                       model_xyz, best_idx, mask, N, M, out);
    return gdm_launch_status("kabsch_stats_kernel");
}
