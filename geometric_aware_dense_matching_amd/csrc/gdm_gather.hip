// Feature gather / pool / scatter kernels of the RandLA point branch and the RGB<->point fusion,
// gfx950 (MI355X).  All are HBM/L2-bound byte movers: no MFMA, no LDS tiling needed -- what
// matters is that index rows are read once per output point (not once per channel, as the
// reference's `index.repeat(1, C, 1)` + torch.gather does) and that every global access of a
// wave is contiguous along the innermost (point / neighbour) axis.
//
// Reference chains replaced (/root/reference):
//   group_gather   models/RandLA/RandLANet.py:729-738 gather_neighbour (+ permute :704-716)
//   gather_max     models/ffb6d.py:128-146 random_sample
//   gather_nn      models/ffb6d.py:148-163 nearest_interpolation, :278-281 choose gather
//   rel_pos_enc    models/RandLA/RandLANet.py:720-727, 701-702
//   att_pool       models/RandLA/RandLANet.py:749-752
// and the lib/pointops signatures grouping / gathering (functions/pointops.py:61-82,151-176).
#include "gdm_common.h"
#include <stdlib.h>
#include <stdint.h>
#include <math.h>

namespace {

constexpr int GB = 256;      // threads per block
constexpr int CCHUNK = 8;    // channels per block in the gather kernels

// out[b,c,e] = feat[b,c,idx[b,e]] for e in [0, m*K): covers group_gather (K>=1) and gather_nn (K==1).
__global__ __launch_bounds__(GB) void group_gather_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                          int C, int n, long mk, float* __restrict__ out)
{
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CCHUNK;
    const long e = (long)blockIdx.x * GB + threadIdx.x;
    if (e >= mk) return;
    int src = idx[(long)b * mk + e];
    src = min(max(src, 0), n - 1);
    const int cend = min(c0 + CCHUNK, C);
    for (int c = c0; c < cend; ++c) {
        const long row = (long)b * C + c;
        out[row * mk + e] = feat[row * n + src];
    }
}

// gbs: floats between two batch items of go (C * mk when go is dense; larger for a channel slice of a wider tensor)
__global__ __launch_bounds__(GB) void group_gather_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx,
                                                              int C, int n, long mk, float* __restrict__ gfeat, long gbs)
{
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CCHUNK;
    const long e = (long)blockIdx.x * GB + threadIdx.x;
    if (e >= mk) return;
    int src = idx[(long)b * mk + e];
    src = min(max(src, 0), n - 1);
    const int cend = min(c0 + CCHUNK, C);
    for (int c = c0; c < cend; ++c) {
        const long row = (long)b * C + c;
        atomicAdd(&gfeat[row * n + src], go[(long)b * gbs + (long)c * mk + e]);
    }
}

// out[b,c,j] = max_k feat[b,c,idx[b,j,k]], arg = first k attaining it (torch.max semantics on CPU)
template <int KMAX>
__global__ __launch_bounds__(GB) void gather_max_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                        int C, int n, int m, int K, float* __restrict__ out,
                                                        int32_t* __restrict__ arg)
{
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CCHUNK;
    const int j = blockIdx.x * GB + threadIdx.x;
    if (j >= m) return;
    int nb[KMAX];
    const int32_t* ip = idx + ((long)b * m + j) * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int v = k < K ? ip[k] : ip[0];
        nb[k] = min(max(v, 0), n - 1);
    }
    const int cend = min(c0 + CCHUNK, C);
    for (int c = c0; c < cend; ++c) {
        const long row = (long)b * C + c;
        const float* f = feat + row * n;
        float best = f[nb[0]];
        int bi = nb[0];
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
            const float v = f[nb[k]];
            if (v > best) {
                best = v;
                bi = nb[k];
            }
        }
        out[row * m + j] = best;
        if (arg) arg[row * m + j] = bi;
    }
}

// The same for LARGE source rows (pixel maps: random_sample of a 64x64 .. 128x128 feature map onto the points, ffb6d.py:128-146): the
// kernel above reads K scattered floats per channel and output -- 4 useful bytes per 32-byte sector, the map fetched ~8x.  Here a
// workgroup owns one (crop, channel) row: the row goes into LDS with one coalesced pass (<= 64 KiB), the gathers hit LDS, the
// index rows are re-read per channel from L2 (16 B vector loads).  Same values, same first-maximum rule.
template <int KMAX>
__global__ __launch_bounds__(GB) void gather_max_rowlds_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                               int C, int n, int m, int K, float* __restrict__ out,
                                                               int32_t* __restrict__ arg)
{
    extern __shared__ __attribute__((aligned(16))) float lrow[];
    const int b = blockIdx.y, c = blockIdx.x;
    const long row = (long)b * C + c;
    const float* f = feat + row * n;
    if ((n & 3) == 0 && (((uintptr_t)f) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < n; i += GB * 4) *reinterpret_cast<float4*>(lrow + i) = *reinterpret_cast<const float4*>(f + i);
    } else {
        for (int i = threadIdx.x; i < n; i += GB) lrow[i] = f[i];
    }
    __syncthreads();
    const bool vec = KMAX == 16 && K == 16 && (((uintptr_t)idx) & 15) == 0;
    for (int j = threadIdx.x; j < m; j += GB) {
        const int32_t* ip = idx + ((long)b * m + j) * K;
        int nb[KMAX];
        if (vec) {
#pragma unroll
            for (int q = 0; q < KMAX / 4; ++q) {
                const int4 v = *reinterpret_cast<const int4*>(ip + 4 * q);
                nb[4 * q] = v.x; nb[4 * q + 1] = v.y; nb[4 * q + 2] = v.z; nb[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) nb[k] = k < K ? ip[k] : ip[0];
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) nb[k] = min(max(nb[k], 0), n - 1);
        float best = lrow[nb[0]];
        int bi = nb[0];
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
            const float v = lrow[nb[k]];
            if (v > best) {
                best = v;
                bi = nb[k];
            }
        }
        out[row * m + j] = best;
        if (arg) arg[row * m + j] = bi;
    }
}

// Few output points (the deep levels: 32 .. 128 points per crop against a 1024 .. 4096-pixel map): one thread per (channel, point),
// so a crop's C * m outputs fill whole waves instead of m lanes walking 8 channels in sequence.
template <int KMAX>
__global__ __launch_bounds__(GB) void gather_max_flat_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                             int C, int n, int m, int K, float* __restrict__ out,
                                                             int32_t* __restrict__ arg)
{
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * GB + threadIdx.x;           // over [C, m]
    if (e >= (long)C * m) return;
    const int c = (int)(e / m), j = (int)(e - (long)c * m);
    const int32_t* ip = idx + ((long)b * m + j) * K;
    const float* f = feat + ((long)b * C + c) * n;
    int n0 = min(max(ip[0], 0), n - 1);
    float best = f[n0];
    int bi = n0;
    float v[KMAX];
    int nb[KMAX];
#pragma unroll
    for (int k = 1; k < KMAX; ++k) {
        nb[k] = min(max(k < K ? ip[k] : ip[0], 0), n - 1);
        v[k] = f[nb[k]];
    }
#pragma unroll
    for (int k = 1; k < KMAX; ++k) {
        if (v[k] > best) {
            best = v[k];
            bi = nb[k];
        }
    }
    out[((long)b * C + c) * m + j] = best;
    if (arg) arg[((long)b * C + c) * m + j] = bi;
}

__global__ __launch_bounds__(GB) void gather_max_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ arg,
                                                            int n, long total_rows_m, int m, float* __restrict__ gfeat)
{
    const long e = (long)blockIdx.x * GB + threadIdx.x;   // over [B*C, m]
    if (e >= total_rows_m) return;
    const long row = e / m;
    atomicAdd(&gfeat[row * n + arg[e]], go[e]);
}

// Same scatter-add with the accumulators PRIVATISED in LDS, for the contended case (few sources, many entries: the backward
// of the neighbour gather has n*K entries landing on n points, that of nearest_interpolation a whole pixel grid on a few hundred
// points).  A block owns (batch item, LCH channels, one segment of the entries): LDS atomics while scanning, then one global
// atomic per touched (source, channel).  10.6 ms -> see DESIGN.md of a 111 ms training step with the global-atomic form.
constexpr int LCH = 4;
template <bool VEC>
__global__ __launch_bounds__(GB) void group_gather_bwd_lds_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx,
                                                                  int C, int n, long mk, long seg_len, float* __restrict__ gfeat, long gbs)
{
    extern __shared__ float accs[];                     // [LCH][n]
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * LCH;
    const int nc = min(LCH, C - c0);
    for (int i = threadIdx.x; i < LCH * n; i += GB) accs[i] = 0.f;
    __syncthreads();
    const long e0 = (long)blockIdx.x * seg_len, e1 = min(mk, e0 + seg_len);
    if (VEC) {
        // four consecutive entries per thread and step (host: mk % 4 == 0, seg_len % 4 == 0, 16-byte aligned rows): the loads of a
        // step -- one int4 of indices, one float4 per channel -- are issued together, the scan is a quarter as many dependent steps
        for (long e = e0 + 4L * threadIdx.x; e < e1; e += 4L * GB) {
            const int4 s4 = *reinterpret_cast<const int4*>(idx + (long)b * mk + e);
            float4 g[LCH];
#pragma unroll
            for (int c = 0; c < LCH; ++c)
                g[c] = c < nc ? *reinterpret_cast<const float4*>(go + (long)b * gbs + (long)(c0 + c) * mk + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int s0 = min(max(s4.x, 0), n - 1), s1 = min(max(s4.y, 0), n - 1), s2 = min(max(s4.z, 0), n - 1), s3 = min(max(s4.w, 0), n - 1);
#pragma unroll
            for (int c = 0; c < LCH; ++c) {
                if (c < nc) {
                    atomicAdd(&accs[c * n + s0], g[c].x);
                    atomicAdd(&accs[c * n + s1], g[c].y);
                    atomicAdd(&accs[c * n + s2], g[c].z);
                    atomicAdd(&accs[c * n + s3], g[c].w);
                }
            }
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += GB) {
            int src = idx[(long)b * mk + e];
            src = min(max(src, 0), n - 1);
            for (int c = 0; c < nc; ++c) atomicAdd(&accs[c * n + src], go[(long)b * gbs + (long)(c0 + c) * mk + e]);
        }
    }
    __syncthreads();
    if (gridDim.x == 1) {
        // one segment: this block is the only writer of its (batch item, channels) rows -- plain stores, zeros included
        for (int i = threadIdx.x; i < nc * n; i += GB) {
            const int c = i / n, j = i - c * n;
            gfeat[((long)b * C + c0 + c) * n + j] = accs[i];
        }
        return;
    }
    for (int i = threadIdx.x; i < nc * n; i += GB) {
        const float v = accs[i];
        if (v != 0.f) {
            const int c = i / n, j = i - c * n;
            atomicAdd(&gfeat[((long)b * C + c0 + c) * n + j], v);
        }
    }
}

// xyz f32[B,n,3], idx i32[B,n,K] -> out f32[B,10,n,K]
__global__ __launch_bounds__(GB) void rel_pos_enc_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx,
                                                         int n, int K, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const long nk = (long)n * K;
    const long e = (long)blockIdx.x * GB + threadIdx.x;
    if (e >= nk) return;
    const int i = (int)(e / K);
    int jn = idx[(long)b * nk + e];
    jn = min(max(jn, 0), n - 1);
    const float* pi = xyz + ((long)b * n + i) * 3;
    const float* pj = xyz + ((long)b * n + jn) * 3;
    const float ax = pi[0], ay = pi[1], az = pi[2];
    const float bx = pj[0], by = pj[1], bz = pj[2];
    // torch: relative = tile - neighbour; dis = sqrt(sum(relative^2)) (RandLANet.py:723-725);
    // each square and sum rounded separately like the elementwise torch ops.
    const float rx = __fsub_rn(ax, bx), ry = __fsub_rn(ay, by), rz = __fsub_rn(az, bz);
    float s = __fmul_rn(rx, rx);
    s = __fadd_rn(s, __fmul_rn(ry, ry));
    s = __fadd_rn(s, __fmul_rn(rz, rz));
    const float dis = __fsqrt_rn(s);
    float* o = out + (long)b * 10 * nk + e;
    o[0 * nk] = dis;
    o[1 * nk] = rx; o[2 * nk] = ry; o[3 * nk] = rz;
    o[4 * nk] = ax; o[5 * nk] = ay; o[6 * nk] = az;
    o[7 * nk] = bx; o[8 * nk] = by; o[9 * nk] = bz;
}

// att, feat f32[rows, K] (rows = B*C*n) -> out f32[rows]: sum_k softmax_k(att) * feat
template <int KMAX>
__global__ __launch_bounds__(GB) void att_pool_kernel(const float* __restrict__ att, const float* __restrict__ feat,
                                                      long rows, int K, float* __restrict__ out)
{
    const long r = (long)blockIdx.x * GB + threadIdx.x;
    if (r >= rows) return;
    const float* a = att + r * K;
    const float* f = feat + r * K;
    float av[KMAX], fv[KMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            av[k] = a[k];
            fv[k] = f[k];
            mx = fmaxf(mx, av[k]);
        }
    }
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float ex = expf(av[k] - mx);
            den += ex;
            av[k] = ex;
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) num += fv[k] * (av[k] / den);      // feature * softmax, then summed (RandLANet.py:750-751)
    }
    out[r] = num;
}

template <int KMAX>
__global__ __launch_bounds__(GB) void att_pool_bwd_kernel(const float* __restrict__ att, const float* __restrict__ feat,
                                                          const float* __restrict__ go, long rows, int K,
                                                          float* __restrict__ gatt, float* __restrict__ gfeat)
{
    const long r = (long)blockIdx.x * GB + threadIdx.x;
    if (r >= rows) return;
    const float* a = att + r * K;
    const float* f = feat + r * K;
    float sv[KMAX], fv[KMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            sv[k] = a[k];
            fv[k] = f[k];
            mx = fmaxf(mx, sv[k]);
        }
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            sv[k] = expf(sv[k] - mx);
            den += sv[k];
        }
    }
    float outv = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            sv[k] = sv[k] / den;
            outv += sv[k] * fv[k];
        }
    }
    const float g = go[r];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            gfeat[r * K + k] = g * sv[k];
            gatt[r * K + k] = g * sv[k] * (fv[k] - outv);
        }
    }
}

// K == 16 forms: thread = (row, quarter of the row).  A wave's float4 loads and stores then cover 1 KiB of consecutive addresses
// (the one-row-per-thread form strides its lanes by 64 B: four times the memory instructions per byte, each touching 64 lines),
// and the reductions over K are three in-thread operations plus two xor-shuffles inside the 4-lane group.
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 1, 64)); return fmaxf(v, __shfl_xor(v, 2, 64)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1, 64); return v + __shfl_xor(v, 2, 64); }

__global__ __launch_bounds__(GB) void att_pool16_kernel(const float4* __restrict__ att, const float4* __restrict__ feat, long rows,
                                                        float* __restrict__ out)
{
    const long q = (long)blockIdx.x * GB + threadIdx.x;          // quarter-row index; rows * 4 is a multiple of 4, so a 4-lane group is all-in or all-out
    const bool live = q < rows * 4;
    const float4 a = live ? att[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 f = live ? feat[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float mx = quad_max(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    const float e0 = expf(a.x - mx), e1 = expf(a.y - mx), e2 = expf(a.z - mx), e3 = expf(a.w - mx);
    const float den = quad_sum((e0 + e1) + (e2 + e3));
    const float num = quad_sum((f.x * (e0 / den) + f.y * (e1 / den)) + (f.z * (e2 / den) + f.w * (e3 / den)));
    if (live && (threadIdx.x & 3) == 0) out[q >> 2] = num;
}

__global__ __launch_bounds__(GB) void att_pool16_bwd_kernel(const float4* __restrict__ att, const float4* __restrict__ feat,
                                                            const float* __restrict__ go, long rows, float4* __restrict__ gatt,
                                                            float4* __restrict__ gfeat)
{
    const long q = (long)blockIdx.x * GB + threadIdx.x;
    const bool live = q < rows * 4;
    const float4 a = live ? att[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 f = live ? feat[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float g = live ? go[q >> 2] : 0.f;
    const float mx = quad_max(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    const float e0 = expf(a.x - mx), e1 = expf(a.y - mx), e2 = expf(a.z - mx), e3 = expf(a.w - mx);
    const float den = quad_sum((e0 + e1) + (e2 + e3));
    const float s0 = e0 / den, s1 = e1 / den, s2 = e2 / den, s3 = e3 / den;
    const float outv = quad_sum((s0 * f.x + s1 * f.y) + (s2 * f.z + s3 * f.w));
    if (!live) return;
    gfeat[q] = make_float4(g * s0, g * s1, g * s2, g * s3);
    gatt[q] = make_float4(g * s0 * (f.x - outv), g * s1 * (f.y - outv), g * s2 * (f.z - outv), g * s3 * (f.w - outv));
}

// seg f32[B,2,N] -> mask u8[B,N], count i32[B]
__global__ __launch_bounds__(GB) void seg_mask_kernel(const float* __restrict__ seg, int N, uint8_t* __restrict__ mask,
                                                      int32_t* __restrict__ count)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * GB + threadIdx.x;
    int sel = 0;
    if (i < N) {
        const float s0 = seg[((long)b * 2 + 0) * N + i];
        const float s1 = seg[((long)b * 2 + 1) * N + i];
        sel = s1 > s0 ? 1 : 0;                          // torch.argmax: first maximum wins on ties -> class 0
        mask[(long)b * N + i] = (uint8_t)sel;
    }
    const unsigned long long bal = __ballot(sel);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&count[b], __popcll(bal));
}

// The same with ONE workgroup per crop (N up to a few 10^4 points): the count is a block reduction written once -- no memset node in
// front of the kernel, no atomics.
__global__ __launch_bounds__(1024) void seg_mask_crop_kernel(const float* __restrict__ seg, int N, uint8_t* __restrict__ mask,
                                                             int32_t* __restrict__ count)
{
    __shared__ int wsum[16];
    const int b = blockIdx.x;
    const float* s0p = seg + (long)b * 2 * N;
    const float* s1p = s0p + N;
    int mine = 0;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const int sel = s1p[i] > s0p[i] ? 1 : 0;        // torch.argmax: first maximum wins on ties -> class 0
        mask[(long)b * N + i] = (uint8_t)sel;
        mine += sel;
    }
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        count[b] = t;
    }
}

} // namespace

#define STREAM(s) ((hipStream_t)(s))

extern "C" int gdm_group_gather_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m, int K,
                                    float* out, void* stream)
{
    GDM_CHECK_ARG(feat && idx && out, "gdm_group_gather_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && C >= 1 && n >= 1 && m >= 1 && K >= 1, "gdm_group_gather_hip: bad shape B=%d C=%d n=%d m=%d K=%d", B, C, n, m, K);
    const long mk = (long)m * K;
    dim3 grid(gdm_cdiv(mk, GB), gdm_cdiv(C, CCHUNK), B);
    GDM_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gdm_group_gather_hip: grid too large");
    hipLaunchKernelGGL(group_gather_kernel, grid, dim3(GB), 0, STREAM(stream), feat, idx, C, n, mk, out);
    return gdm_launch_status("group_gather_kernel");
}

extern "C" int gdm_group_gather_bwd2_hip(const float* go, long go_bstride, const int32_t* idx, int B, int C, int n, int m, int K,
                                         float* gfeat, void* stream);

extern "C" int gdm_group_gather_bwd_hip(const float* go, const int32_t* idx, int B, int C, int n, int m, int K,
                                        float* gfeat, void* stream)
{
    return gdm_group_gather_bwd2_hip(go, (long)C * m * K, idx, B, C, n, m, K, gfeat, stream);
}

// go_bstride: floats between two batch items of go (rows of m * K floats, channel stride m * K): a channel slice of a concatenation's
// gradient is read in place
extern "C" int gdm_group_gather_bwd2_hip(const float* go, long go_bstride, const int32_t* idx, int B, int C, int n, int m, int K,
                                         float* gfeat, void* stream)
{
    GDM_CHECK_ARG(go && idx && gfeat, "gdm_group_gather_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && C >= 1 && n >= 1 && m >= 1 && K >= 1, "gdm_group_gather_bwd_hip: bad shape");
    GDM_CHECK_ARG(go_bstride >= (long)C * m * K, "gdm_group_gather_bwd_hip: batch stride %ld below C*m*K", go_bstride);
    const long gbs = go_bstride;
    const long mk = (long)m * K;
    if ((size_t)n * LCH * sizeof(float) <= 64 * 1024 && mk >= 4L * n && gdm_cdiv(C, LCH) <= 65535 && B <= 65535) {
        // contended: privatise in LDS; segments sized so that a block scans >= 8 entries per accumulator it later flushes
        long seg_len = 8L * n;
        if (seg_len < 4096) seg_len = 4096;
        long nseg = gdm_cdiv(mk, seg_len);
        if (nseg > 65535) { nseg = 65535; seg_len = gdm_cdiv(mk, nseg); }
        dim3 grid((unsigned)nseg, gdm_cdiv(C, LCH), B);
        const bool vec = mk % 4 == 0 && seg_len % 4 == 0 && gbs % 4 == 0 && (((uintptr_t)go | (uintptr_t)idx) & 15) == 0;
        if (vec)
            hipLaunchKernelGGL(group_gather_bwd_lds_kernel<true>, grid, dim3(GB), (size_t)n * LCH * sizeof(float), STREAM(stream), go, idx, C, n,
                               mk, seg_len, gfeat, gbs);
        else
            hipLaunchKernelGGL(group_gather_bwd_lds_kernel<false>, grid, dim3(GB), (size_t)n * LCH * sizeof(float), STREAM(stream), go, idx, C, n,
                               mk, seg_len, gfeat, gbs);
        return gdm_launch_status("group_gather_bwd_lds_kernel");
    }
    dim3 grid(gdm_cdiv(mk, GB), gdm_cdiv(C, CCHUNK), B);
    GDM_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gdm_group_gather_bwd_hip: grid too large");
    hipLaunchKernelGGL(group_gather_bwd_kernel, grid, dim3(GB), 0, STREAM(stream), go, idx, C, n, mk, gfeat, gbs);
    return gdm_launch_status("group_gather_bwd_kernel");
}

extern "C" int gdm_gather_nn_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m, float* out, void* stream)
{
    return gdm_group_gather_hip(feat, idx, B, C, n, m, 1, out, stream);
}

extern "C" int gdm_gather_nn_bwd_hip(const float* go, const int32_t* idx, int B, int C, int n, int m, float* gfeat, void* stream)
{
    return gdm_group_gather_bwd_hip(go, idx, B, C, n, m, 1, gfeat, stream);
}

extern "C" int gdm_gather_max_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m, int K,
                                  float* out, int32_t* arg, void* stream)
{
    GDM_CHECK_ARG(feat && idx && out, "gdm_gather_max_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && C >= 1 && n >= 1 && m >= 1, "gdm_gather_max_hip: bad shape");
    GDM_CHECK_ARG(K >= 1 && K <= 32, "gdm_gather_max_hip: K=%d not in [1,32]", K);
    dim3 grid(gdm_cdiv(m, GB), gdm_cdiv(C, CCHUNK), B);
    GDM_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gdm_gather_max_hip: grid too large");
    // pixel-map sources: one workgroup per (crop, channel) row staged in LDS (GDM_GATHER_MAX_ROWLDS=0: the scattered form, for A/B)
    static int rowlds = -1;
    if (rowlds < 0) {
        const char* e = getenv("GDM_GATHER_MAX_ROWLDS");
        rowlds = (e && e[0] == '0') ? 0 : 1;
    }
    if (rowlds && m < 256 && (long)C * m >= 4096) {
        dim3 g3(gdm_cdiv((long)C * m, GB), B);
        if (K <= 16)
            hipLaunchKernelGGL(gather_max_flat_kernel<16>, g3, dim3(GB), 0, STREAM(stream), feat, idx, C, n, m, K, out, arg);
        else
            hipLaunchKernelGGL(gather_max_flat_kernel<32>, g3, dim3(GB), 0, STREAM(stream), feat, idx, C, n, m, K, out, arg);
        return gdm_launch_status("gather_max_flat_kernel");
    }
    if (rowlds && n >= 1024 && n <= 16384 && (long)m * K >= 2048 && C <= 65535) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)gather_max_rowlds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            (void)hipFuncSetAttribute((const void*)gather_max_rowlds_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            attr = true;
        }
        dim3 g2(C, B);
        const size_t lds = (size_t)((n + 3) & ~3) * 4;
        if (K <= 16)
            hipLaunchKernelGGL(gather_max_rowlds_kernel<16>, g2, dim3(GB), lds, STREAM(stream), feat, idx, C, n, m, K, out, arg);
        else
            hipLaunchKernelGGL(gather_max_rowlds_kernel<32>, g2, dim3(GB), lds, STREAM(stream), feat, idx, C, n, m, K, out, arg);
        return gdm_launch_status("gather_max_rowlds_kernel");
    }
    if (K <= 16)
        hipLaunchKernelGGL(gather_max_kernel<16>, grid, dim3(GB), 0, STREAM(stream), feat, idx, C, n, m, K, out, arg);
    else
        hipLaunchKernelGGL(gather_max_kernel<32>, grid, dim3(GB), 0, STREAM(stream), feat, idx, C, n, m, K, out, arg);
    return gdm_launch_status("gather_max_kernel");
}

extern "C" int gdm_gather_max_bwd_hip(const float* go, const int32_t* arg, int B, int C, int n, int m,
                                      float* gfeat, void* stream)
{
    GDM_CHECK_ARG(go && arg && gfeat, "gdm_gather_max_bwd_hip: NULL pointer");
    const long total = (long)B * C * m;
    hipLaunchKernelGGL(gather_max_bwd_kernel, dim3(gdm_cdiv(total, GB)), dim3(GB), 0, STREAM(stream), go, arg, n, total, m, gfeat);
    return gdm_launch_status("gather_max_bwd_kernel");
}

extern "C" int gdm_rel_pos_enc_hip(const float* xyz, const int32_t* idx, int B, int n, int K, float* out, void* stream)
{
    GDM_CHECK_ARG(xyz && idx && out, "gdm_rel_pos_enc_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && n >= 1 && K >= 1 && B <= 65535, "gdm_rel_pos_enc_hip: bad shape");
    dim3 grid(gdm_cdiv((long)n * K, GB), B);
    hipLaunchKernelGGL(rel_pos_enc_kernel, grid, dim3(GB), 0, STREAM(stream), xyz, idx, n, K, out);
    return gdm_launch_status("rel_pos_enc_kernel");
}

extern "C" int gdm_att_pool_hip(const float* att, const float* feat, int B, int C, int n, int K, float* out, void* stream)
{
    GDM_CHECK_ARG(att && feat && out, "gdm_att_pool_hip: NULL pointer");
    GDM_CHECK_ARG(K >= 1 && K <= 32, "gdm_att_pool_hip: K=%d not in [1,32]", K);
    const long rows = (long)B * C * n;
    if (K == 16 && (((uintptr_t)att | (uintptr_t)feat) & 15) == 0)
        hipLaunchKernelGGL(att_pool16_kernel, dim3(gdm_cdiv(rows * 4, GB)), dim3(GB), 0, STREAM(stream), (const float4*)att, (const float4*)feat, rows, out);
    else if (K <= 16)
        hipLaunchKernelGGL(att_pool_kernel<16>, dim3(gdm_cdiv(rows, GB)), dim3(GB), 0, STREAM(stream), att, feat, rows, K, out);
    else
        hipLaunchKernelGGL(att_pool_kernel<32>, dim3(gdm_cdiv(rows, GB)), dim3(GB), 0, STREAM(stream), att, feat, rows, K, out);
    return gdm_launch_status("att_pool_kernel");
}

extern "C" int gdm_att_pool_bwd_hip(const float* att, const float* feat, const float* go, int B, int C, int n, int K,
                                    float* gatt, float* gfeat, void* stream)
{
    GDM_CHECK_ARG(att && feat && go && gatt && gfeat, "gdm_att_pool_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(K >= 1 && K <= 32, "gdm_att_pool_bwd_hip: K=%d not in [1,32]", K);
    const long rows = (long)B * C * n;
    if (K == 16 && (((uintptr_t)att | (uintptr_t)feat | (uintptr_t)gatt | (uintptr_t)gfeat) & 15) == 0)
        hipLaunchKernelGGL(att_pool16_bwd_kernel, dim3(gdm_cdiv(rows * 4, GB)), dim3(GB), 0, STREAM(stream), (const float4*)att, (const float4*)feat, go,
                           rows, (float4*)gatt, (float4*)gfeat);
    else if (K <= 16)
        hipLaunchKernelGGL(att_pool_bwd_kernel<16>, dim3(gdm_cdiv(rows, GB)), dim3(GB), 0, STREAM(stream), att, feat, go, rows, K, gatt, gfeat);
    else
        hipLaunchKernelGGL(att_pool_bwd_kernel<32>, dim3(gdm_cdiv(rows, GB)), dim3(GB), 0, STREAM(stream), att, feat, go, rows, K, gatt, gfeat);
    return gdm_launch_status("att_pool_bwd_kernel");
}

extern "C" int gdm_seg_mask_hip(const float* seg, int B, int N, uint8_t* mask, int32_t* count, void* stream)
{
    GDM_CHECK_ARG(seg && mask && count, "gdm_seg_mask_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && N >= 1 && B <= 65535, "gdm_seg_mask_hip: bad shape");
    if (N <= 65536) {
        hipLaunchKernelGGL(seg_mask_crop_kernel, dim3(B), dim3(1024), 0, STREAM(stream), seg, N, mask, count);
        return gdm_launch_status("seg_mask_crop_kernel");
    }
    GDM_HIP(hipMemsetAsync(count, 0, sizeof(int32_t) * B, STREAM(stream)));
    hipLaunchKernelGGL(seg_mask_kernel, dim3(gdm_cdiv(N, GB), B), dim3(GB), 0, STREAM(stream), seg, N, mask, count);
    return gdm_launch_status("seg_mask_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Batched strided copies: the strided xyz grids (linemod_pbr.py:517-527), the prefix sub-clouds (:538) and the pooling index
// prefixes of a whole neighbour pyramid, each a dense [B, R1, R2, E] array of 4-byte words read from a strided view -- one launch
// for the whole table instead of one `contiguous()` copy kernel per view.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct CopyTable {
    gdm_copy_job j[GDM_COPY_MAX_JOBS];
    long begin[GDM_COPY_MAX_JOBS + 1];    // first word (of the launch) of every job
    int n;
};

__global__ __launch_bounds__(256) void copy_jobs_kernel(const CopyTable t)
{
    const long w = (long)blockIdx.x * 256 + threadIdx.x;
    if (w >= t.begin[t.n]) return;
    int ji = 0;
    while (ji + 1 < t.n && w >= t.begin[ji + 1]) ++ji;
    const gdm_copy_job& j = t.j[ji];
    long r = w - t.begin[ji];
    const int e = (int)(r % j.E);
    r /= j.E;
    const int r2 = (int)(r % j.R2);
    r /= j.R2;
    const int r1 = (int)(r % j.R1);
    const long b = r / j.R1;
    static_cast<uint32_t*>(j.dst)[w - t.begin[ji]] =
        static_cast<const uint32_t*>(j.src)[b * j.sb + (long)r1 * j.s1 + (long)r2 * j.s2 + e];
}

} // namespace

extern "C" int gdm_copy_jobs_hip(const gdm_copy_job* jobs, int njobs, void* stream)
{
    GDM_CHECK_ARG(jobs && njobs >= 1 && njobs <= GDM_COPY_MAX_JOBS, "gdm_copy_jobs_hip: njobs=%d out of range", njobs);
    CopyTable t;
    t.n = njobs;
    long total = 0;
    for (int i = 0; i < njobs; ++i) {
        const gdm_copy_job& j = jobs[i];
        GDM_CHECK_ARG(j.dst && j.src && j.B >= 1 && j.R1 >= 1 && j.R2 >= 1 && j.E >= 1, "gdm_copy_jobs_hip: job %d is empty", i);
        t.j[i] = j;
        t.begin[i] = total;
        total += (long)j.B * j.R1 * j.R2 * j.E;
    }
    t.begin[njobs] = total;
    GDM_CHECK_ARG(total <= 0x7fffffffL * 256, "gdm_copy_jobs_hip: too many words");
    hipLaunchKernelGGL(copy_jobs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t);
    return gdm_launch_status("copy_jobs_kernel");
}
