// Per-point (1x1) layers of the point branch and of the RGB <-> point fusion, gfx950 (inference).
//
// Replaces, per layer, the chain  torch.cat -> 1x1 Conv1d/Conv2d (a library GEMM) -> BatchNorm -> activation (-> residual add)  of
//   /root/reference/models/pytorch_utils.py:70-124 and models/RandLA/pytorch_utils.py:34-99 (`_ConvBase`, eval mode),
//   Dilated_res_block's tail  lrelu(mlp2(f) + shortcut(x))            /root/reference/models/RandLA/RandLANet.py:685-688,
//   the fusion layers over cat(point features, pooled pixel features)   /root/reference/models/ffb6d.py:224-231,259-265,
//   the decoder layers over cat(skip, nearest_interpolation(deeper))    /root/reference/models/ffb6d.py:246-250,268-272
// by ONE launch:   out[b,:,i] = act( scale * (W . [x0 ; x1][b,:,i]) + shift  (+ rscale * (Wr . xr[b,:,i]) + rshift) )
// where every input segment is a channel-major tensor f32[B,C,n_src] read either in place (n_src == n) or through a per-point
// index idx[b,i] (the nearest-neighbour interpolation of ffb6d.py:148-163 folded into the load), or a point-major tensor f32[B*n,C].
// The concat is never formed: the K loop walks the segments in order, so the sum runs over the same channels in the same order.
//
// These layers are tiny (8..1024 channels at 128..32768 points: < 2 GFLOP for all ~40 of them per step) and the step is bound by
// their launch count, not their arithmetic: fp32 FMAs (exact fp32 products, ascending-k summation), a 64-point x 64-channel tile
// per 256-thread workgroup, operands staged through LDS in 16-deep K chunks, coalesced along the points.
#include "gdm_common.h"

namespace {

constexpr int PT = 64, CT = 64, KT = 16, XS = 68;       // XS: padded row of the x tile (keeps float4 reads aligned, spreads banks)

struct PwSegDev {
    const float* x;
    const int32_t* idx;
    int C, n_src, point_major;
};

struct PwArgs {
    PwSegDev seg[2];
    int nseg;
    const float* wt;          // [K][Cout], K = sum of the segments' channels, rows in segment order
    const float* scale;       // [Cout] or NULL (1)
    const float* shift;       // [Cout] or NULL (0)
    PwSegDev rseg;            // residual branch (x == NULL: none)
    const float* rwt;         // [Cr][Cout]
    const float* rscale;
    const float* rshift;
    float* out;
    int n, Cout, outC, out_c0, point_major, act;
    float slope;
    long total;               // B * n
};

__device__ __forceinline__ void pw_gemm(const PwSegDev& s, const float* __restrict__ wt, int Cout, int n, long g0, long total,
                                        int c0, float (*xs)[XS], float (*ws)[CT], float (&acc)[4][4])
{
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int C = s.C;
    // this thread's column of the x tile
    const float* xb = nullptr;
    long xstride = 0;
    int lp, lk;
    if (!s.point_major) {
        lp = tid & 63;
        lk = tid >> 6;                                    // 0..3, rows lk + 4 r
        const long g = g0 + lp;
        if (g < total) {
            const long b = g / n;
            const int i = (int)(g - b * n);
            int col = i;
            if (s.idx) col = min(max(s.idx[g], 0), s.n_src - 1);
            xb = s.x + b * (long)C * s.n_src + col;
            xstride = s.n_src;
        }
    } else {
        lp = tid >> 4;                                    // 0..15, points lp + 16 r
        lk = tid & 15;
    }
    const int wc = c0 + (tid & 63);
    const int wk = tid >> 6;
    for (int k0 = 0; k0 < C; k0 += KT) {
        if (!s.point_major) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + lk + 4 * r;
                xs[lk + 4 * r][lp] = (xb && k < C) ? xb[(long)k * xstride] : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long g = g0 + lp + 16 * r;
                const int k = k0 + lk;
                float v = 0.f;
                if (g < total && k < C) {
                    long row = g;
                    if (s.idx) {
                        const long b = g / n;
                        row = b * s.n_src + min(max(s.idx[g], 0), s.n_src - 1);
                    }
                    v = s.x[row * C + k];
                }
                xs[lk][lp + 16 * r] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + wk + 4 * r;
            ws[wk + 4 * r][tid & 63] = (k < C && wc < Cout) ? wt[(long)k * Cout + wc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[kk][tx * 4]);
            const float4 wv = *reinterpret_cast<const float4*>(&ws[kk][ty * 4]);
            acc[0][0] = fmaf(wv.x, xv.x, acc[0][0]); acc[0][1] = fmaf(wv.x, xv.y, acc[0][1]);
            acc[0][2] = fmaf(wv.x, xv.z, acc[0][2]); acc[0][3] = fmaf(wv.x, xv.w, acc[0][3]);
            acc[1][0] = fmaf(wv.y, xv.x, acc[1][0]); acc[1][1] = fmaf(wv.y, xv.y, acc[1][1]);
            acc[1][2] = fmaf(wv.y, xv.z, acc[1][2]); acc[1][3] = fmaf(wv.y, xv.w, acc[1][3]);
            acc[2][0] = fmaf(wv.z, xv.x, acc[2][0]); acc[2][1] = fmaf(wv.z, xv.y, acc[2][1]);
            acc[2][2] = fmaf(wv.z, xv.z, acc[2][2]); acc[2][3] = fmaf(wv.z, xv.w, acc[2][3]);
            acc[3][0] = fmaf(wv.w, xv.x, acc[3][0]); acc[3][1] = fmaf(wv.w, xv.y, acc[3][1]);
            acc[3][2] = fmaf(wv.w, xv.z, acc[3][2]); acc[3][3] = fmaf(wv.w, xv.w, acc[3][3]);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pointwise_kernel(const PwArgs a)
{
    __shared__ __attribute__((aligned(16))) float xs[KT][XS];
    __shared__ __attribute__((aligned(16))) float ws[KT][CT];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const long g0 = (long)blockIdx.x * PT;
    const int c0 = (int)blockIdx.y * CT;
    const int n = a.n, Cout = a.Cout;

    float acc[4][4], racc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = racc[j][q] = 0.f;

    const float* wt = a.wt;
    for (int s = 0; s < a.nseg; ++s) {
        pw_gemm(a.seg[s], wt, Cout, n, g0, a.total, c0, xs, ws, acc);
        wt += (long)a.seg[s].C * Cout;
    }
    const bool has_res = a.rseg.x != nullptr;
    if (has_res) pw_gemm(a.rseg, a.rwt, Cout, n, g0, a.total, c0, xs, ws, racc);

    const long p0 = g0 + tx * 4;
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty * 4 + j;
        const bool live = c < Cout;
        const float sc = (live && a.scale) ? a.scale[c] : 1.f;
        const float sh = (live && a.shift) ? a.shift[c] : 0.f;
        const float rs = (live && has_res && a.rscale) ? a.rscale[c] : 1.f;
        const float rb = (live && has_res && a.rshift) ? a.rshift[c] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = fmaf(acc[j][q], sc, sh);
            if (has_res) y += fmaf(racc[j][q], rs, rb);
            if (a.act == 1) y = fmaxf(y, 0.f);
            else if (a.act == 2) y = y > 0.f ? y : y * a.slope;
            v[j][q] = y;
        }
    }
    if (a.point_major) {
        // out[p][out_c0 + c]: the thread's four channels are contiguous
        const int c = c0 + ty * 4;
        const bool vec = (a.outC % 4 == 0) && (a.out_c0 % 4 == 0) && (c + 3 < Cout);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long p = p0 + q;
            if (p >= a.total) continue;
            float* o = a.out + p * a.outC + a.out_c0 + c;
            if (vec) *reinterpret_cast<float4*>(o) = make_float4(v[0][q], v[1][q], v[2][q], v[3][q]);
            else
                for (int j = 0; j < 4; ++j)
                    if (c + j < Cout) o[j] = v[j][q];
        }
    } else {
        const bool vec = (n % 4 == 0) && (p0 + 3 < a.total);       // then the four points lie in one crop and the row is 16-B aligned
        const long b = p0 < a.total ? p0 / n : 0;
        const int i = (int)(p0 - b * n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty * 4 + j;
            if (c >= Cout) continue;
            if (vec) {
                float* o = a.out + ((long)b * a.outC + a.out_c0 + c) * n + i;
                *reinterpret_cast<float4*>(o) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
            } else {
                for (int q = 0; q < 4; ++q) {
                    const long p = p0 + q;
                    if (p >= a.total) break;
                    const long bb = p / n;
                    a.out[((long)bb * a.outC + a.out_c0 + c) * n + (p - bb * n)] = v[j][q];
                }
            }
        }
    }
}

bool seg_ok(const gdm_pw_seg& s, int n)
{
    return s.x && s.C >= 1 && s.n_src >= 1 && (s.idx || s.n_src == n);
}

PwSegDev to_dev(const gdm_pw_seg& s) { return PwSegDev{s.x, s.idx, s.C, s.n_src, s.point_major}; }

} // namespace

extern "C" int gdm_pointwise_hip(const gdm_pw_seg* segs, int nseg, const float* wt, const float* scale, const float* shift,
                                 const gdm_pw_seg* rseg, const float* rwt, const float* rscale, const float* rshift,
                                 int B, int n, int Cout, int act, float slope, float* out, int out_C, int out_c0, int point_major,
                                 void* stream)
{
    GDM_CHECK_ARG(segs && wt && out, "gdm_pointwise_hip: NULL pointer");
    GDM_CHECK_ARG(nseg >= 1 && nseg <= 2, "gdm_pointwise_hip: nseg=%d not in [1,2]", nseg);
    GDM_CHECK_ARG(B >= 1 && n >= 1 && Cout >= 1, "gdm_pointwise_hip: bad shape B=%d n=%d Cout=%d", B, n, Cout);
    GDM_CHECK_ARG(out_c0 >= 0 && out_c0 + Cout <= out_C, "gdm_pointwise_hip: channels [%d, %d) outside the output's %d", out_c0,
                  out_c0 + Cout, out_C);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_pointwise_hip: act=%d", act);
    for (int s = 0; s < nseg; ++s)
        GDM_CHECK_ARG(seg_ok(segs[s], n), "gdm_pointwise_hip: segment %d: NULL / empty, or n_src=%d != n=%d without an index", s,
                      segs[s].n_src, n);
    GDM_CHECK_ARG(!rseg || (seg_ok(*rseg, n) && rwt), "gdm_pointwise_hip: bad residual segment");
    GDM_CHECK_ARG(((uintptr_t)out & 15) == 0, "gdm_pointwise_hip: out must be 16-byte aligned");
    PwArgs a;
    a.nseg = nseg;
    a.seg[0] = to_dev(segs[0]);
    a.seg[1] = nseg > 1 ? to_dev(segs[1]) : PwSegDev{nullptr, nullptr, 0, 0, 0};
    a.wt = wt;
    a.scale = scale;
    a.shift = shift;
    a.rseg = rseg ? to_dev(*rseg) : PwSegDev{nullptr, nullptr, 0, 0, 0};
    a.rwt = rwt;
    a.rscale = rscale;
    a.rshift = rshift;
    a.out = out;
    a.n = n;
    a.Cout = Cout;
    a.outC = out_C;
    a.out_c0 = out_c0;
    a.point_major = point_major ? 1 : 0;
    a.act = act;
    a.slope = slope;
    a.total = (long)B * n;
    const long tiles = (a.total + PT - 1) / PT;
    GDM_CHECK_ARG(tiles <= 0x7fffffffL && gdm_cdiv(Cout, CT) <= 65535, "gdm_pointwise_hip: grid too large");
    dim3 grid((unsigned)tiles, gdm_cdiv(Cout, CT));
    hipLaunchKernelGGL(pointwise_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return gdm_launch_status("pointwise_kernel");
}
