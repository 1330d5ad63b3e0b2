// Bilinear resize (align_corners=True) of NCHW fp32 feature maps, gfx950.
//
// Used by the image branch for nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
// (/root/reference/models/cnn/pspnet.py:38 PSPUpsample) and F.interpolate(..., size=(h,w),
// mode='bilinear', align_corners=True) of the pyramid-pooling priors (pspnet.py:26-29).
// PyTorch-ROCm's own kernel for this op took 2.1 ms per call at batch 16 (14.8 ms of a 36.5 ms
// step, profiles/r01_kernel_stats_eager.csv); the op is a pure HBM stream (every output element
// written once, every input element read ~once through L2), so it is written here as one:
// one thread = 4 consecutive output pixels of one row (one 16-B store), source rows hit L1/L2.
// Index / weight arithmetic is the same as ATen's (scale = (in-1)/(out-1) in fp32,
// src = scale*dst, i0 = (int)src, lambda = src - i0, i1 = i0 + (i0 < in-1)).
#include "gdm_common.h"

namespace {

__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ in, int H, int W, int OH, int OW,
                                                                float rh, float rw, float* __restrict__ out)
{
    // one thread = 4 consecutive output pixels of one row; rows of a plane are flattened so that small maps
    // (32x32 -> 64x64 has only 16 quads per row) still fill the wave
    const long plane = blockIdx.y;
    const int qpr = (OW + 3) >> 2;                       // quads per output row
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * OH) return;
    const int oy = q / qpr;
    const int ox0 = (q - oy * qpr) * 4;
    const float sy = rh * (float)oy;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const float* r0 = in + (plane * H + y0) * (long)W;
    const float* r1 = in + (plane * H + y1) * (long)W;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = min(ox0 + j, OW - 1);
        const float sx = rw * (float)ox;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float lx = sx - (float)x0, hx = 1.f - lx;
        v[j] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
    }
    float* o = out + (plane * OH + oy) * (long)OW + ox0;
    if (ox0 + 3 < OW && (((uintptr_t)o) & 15) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int j = 0; j < 4 && ox0 + j < OW; ++j) o[j] = v[j];
    }
}

// grad_in = transpose of the interpolation, as a GATHER: one thread per input pixel sums the (few) output pixels whose two
// source rows / columns include it, with the forward's own arithmetic for (y0, y1, ly).  No atomics, deterministic; the scatter
// form (4 atomics per output pixel, neighbouring lanes on the same address) took 63 ms of a 176 ms training step.
__device__ __forceinline__ void src_range(int i, int n_in, int n_out, float r, int& lo, int& hi)
{
    // outputs o with floor(r*o) in {i-1, i}; padded by one on both sides against rounding, clipped to [0, n_out-1]
    if (r <= 0.f) { lo = 0; hi = n_out - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / r) - 1);
    hi = min(n_out - 1, (int)ceilf((float)(i + 1) / r) + 1);
}

// interpolation weight with which output index o reads input index i (the forward's arithmetic)
__device__ __forceinline__ float tap_weight(int o, int i, int n_in, float r)
{
    const float sp = r * (float)o;
    const int i0 = min((int)sp, n_in - 1);
    const int i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    const float l = sp - (float)i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

// Both transposes below run on 16x16 tiles of the low-resolution plane: the workgroup first lists, for its 16 rows and 16
// columns, the (output index, weight) pairs with a non-zero weight (<= TL_MAX each; 4-5 for x2), then every thread sums
// |rows| x |cols| products -- the per-thread weight arithmetic of the plain gather (~200 instructions per element) is gone.
// Axes with more contributors than TL_MAX (large magnifications, e.g. the 1x1 pyramid-pooling prior) use the plain loops.
constexpr int TL = 16, TL_MAX = 6;

struct TapLists {
    int yo[TL][TL_MAX], xo[TL][TL_MAX];
    float yw[TL][TL_MAX], xw[TL][TL_MAX];
    int yn[TL], xn[TL];
    int overflow;
};

// thread t < 16 lists row y0+t, thread 16 <= t < 32 lists column x0+(t-16); `shift` = tap offset (output index = sample - shift)
__device__ __forceinline__ void build_tap_lists(TapLists& L, int y0, int x0, int H, int W, int OH, int OW, float rh, float rw, int dy, int dx)
{
    const int t = threadIdx.x;
    if (t == 0) L.overflow = 0;
    __syncthreads();
    if (t < 2 * TL) {
        const bool isy = t < TL;
        const int i = (isy ? y0 : x0) + (isy ? t : t - TL);
        const int n_in = isy ? H : W, n_out = isy ? OH : OW, sh = isy ? dy : dx;
        const float r = isy ? rh : rw;
        int cnt = 0;
        if (i < n_in) {
            int lo, hi;
            src_range(i, n_in, n_out, r, lo, hi);
            lo = max(lo, max(0, sh));                        // the sample and the output pixel it belongs to must both be inside
            hi = min(hi, n_out - 1 + min(0, sh));
            for (int o = lo; o <= hi; ++o) {
                const float w = tap_weight(o, i, n_in, r);
                if (w != 0.f) {
                    if (cnt < TL_MAX) {
                        if (isy) { L.yo[t][cnt] = o - sh; L.yw[t][cnt] = w; }
                        else { L.xo[t - TL][cnt] = o - sh; L.xw[t - TL][cnt] = w; }
                    }
                    ++cnt;
                }
            }
        }
        if (cnt > TL_MAX) atomicOr(&L.overflow, 1);
        if (isy) L.yn[t] = min(cnt, TL_MAX);
        else L.xn[t - TL] = min(cnt, TL_MAX);
    }
    __syncthreads();
}

__device__ __forceinline__ float gather_transpose_plain(const float* __restrict__ gp, int y, int x, int H, int W, int OH, int OW,
                                                        float rh, float rw, int dy, int dx)
{
    int Y_lo, Y_hi, X_lo, X_hi;
    src_range(y, H, OH, rh, Y_lo, Y_hi);
    src_range(x, W, OW, rw, X_lo, X_hi);
    Y_lo = max(Y_lo, max(0, dy));
    Y_hi = min(Y_hi, OH - 1 + min(0, dy));
    X_lo = max(X_lo, max(0, dx));
    X_hi = min(X_hi, OW - 1 + min(0, dx));
    float acc = 0.f;
    for (int Y = Y_lo; Y <= Y_hi; ++Y) {
        const float wy = tap_weight(Y, y, H, rh);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X_lo; X <= X_hi; ++X) row = fmaf(tap_weight(X, x, W, rw), gp[(long)(Y - dy) * OW + (X - dx)], row);
        acc = fmaf(wy, row, acc);
    }
    return acc;
}

__device__ __forceinline__ float gather_transpose_tile(const TapLists& L, const float* __restrict__ gp, int ty, int tx, int y, int x,
                                                       int H, int W, int OH, int OW, float rh, float rw, int dy, int dx)
{
    if (L.overflow) return gather_transpose_plain(gp, y, x, H, W, OH, OW, rh, rw, dy, dx);
    float acc = 0.f;
    const int ny = L.yn[ty], nx = L.xn[tx];
    for (int a = 0; a < ny; ++a) {
        const float* rowp = gp + (long)L.yo[ty][a] * OW;
        float row = 0.f;
        for (int b = 0; b < nx; ++b) row = fmaf(L.xw[tx][b], rowp[L.xo[tx][b]], row);
        acc = fmaf(L.yw[ty][a], row, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ go, int H, int W, int OH, int OW,
                                                                    float rh, float rw, float* __restrict__ gin)
{
    __shared__ TapLists L;
    const long plane = blockIdx.y;
    const int tiles_x = (W + TL - 1) / TL;
    const int y0 = (blockIdx.x / tiles_x) * TL, x0 = (blockIdx.x % tiles_x) * TL;
    build_tap_lists(L, y0, x0, H, W, OH, OW, rh, rw, 0, 0);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) return;
    gin[plane * (long)H * W + (long)y * W + x] = gather_transpose_tile(L, go + plane * (long)OH * OW, ty, tx, y, x, H, W, OH, OW, rh, rw, 0, 0);
}

// The same transpose for SMALL source maps (the pyramid-pooling priors, 1x1 .. 8x8 -> 32x32): every source pixel collects from a large
// part of the output, so the tile form above overflows its lists and falls back to per-element loops over the whole output with
// a few threads per plane (0.42 ms per call on 24 x 512 planes).  Here a workgroup owns a plane: grad_out in LDS, then the
// separable sums  t[Y][j] = sum_X wx(X, j) go[Y][X],  gin[i][j] = sum_Y wy(Y, i) t[Y][j]  (the order of gather_transpose_plain).
constexpr int USM_IN = 8, USM_OUT = 4096, USM_OW = 64;

__global__ __launch_bounds__(256) void upsample_bilinear_bwd_small_kernel(const float* __restrict__ go, int H, int W, int OH, int OW,
                                                                          float rh, float rw, float* __restrict__ gin)
{
    __shared__ float g[USM_OUT];
    __shared__ float tsum[USM_OW * USM_IN];              // [OH][W]
    const long plane = blockIdx.x;
    const int t = threadIdx.x, ohw = OH * OW;
    const float* gp = go + plane * ohw;
    for (int i = t; i < ohw; i += 256) g[i] = gp[i];
    __syncthreads();
    for (int e = t; e < OH * W; e += 256) {
        const int Y = e / W, j = e - Y * W;
        float row = 0.f;
        for (int X = 0; X < OW; ++X) row = fmaf(tap_weight(X, j, W, rw), g[Y * OW + X], row);
        tsum[e] = row;
    }
    __syncthreads();
    for (int e = t; e < H * W; e += 256) {
        const int i = e / W, j = e - i * W;
        float acc = 0.f;
        for (int Y = 0; Y < OH; ++Y) acc = fmaf(tap_weight(Y, i, H, rh), tsum[Y * W + j], acc);
        gin[plane * (long)(H * W) + e] = acc;
    }
}

// out[plane, i] = max_k act(scale[c] * x[plane, i, k] + shift[c]), c = plane % C: eval-mode BatchNorm + LeakyReLU + the max over the
// K neighbours of the DGCNN edge convolutions (dgcnn.py:104-117) in one pass (three launches and two [B,64,n,K] round trips otherwise).
template <int ACT>
__global__ __launch_bounds__(256) void affine_act_maxk_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int C, long n, int K4, float slope,
                                                              float* __restrict__ out)
{
    const long plane = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(plane % C);
    const float a = scale[c], b = shift[c];
    const float4* xp = reinterpret_cast<const float4*>(x + (plane * n + i) * (long)(4 * K4));
    float m = -INFINITY;
    for (int k = 0; k < K4; ++k) {
        const float4 v = xp[k];
        float o[4] = {v.x * a + b, v.y * a + b, v.z * a + b, v.w * a + b};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == 1) o[j] = fmaxf(o[j], 0.f);
            if (ACT == 2) o[j] = o[j] > 0.f ? o[j] : o[j] * slope;
            m = fmaxf(m, o[j]);
        }
    }
    out[plane * n + i] = m;
}

// Single-slope PReLU with its backward (the PSPUpsample activations, pspnet.py:41, on up to 10^8 elements in training):
// y = x > 0 ? x : a x;  gx = x > 0 ? go : a go;  ga = sum_{x <= 0} x go.  torch's multi-output elementwise backward runs these at
// ~0.6 TB/s (1.9 ms per call); this is one streaming pass with a block reduction and one atomic per block for ga.
__global__ __launch_bounds__(256) void prelu1_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ slope, long n4,
                                                         float4* __restrict__ y)
{
    const float a = slope[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = x[i];
        v.x = v.x > 0.f ? v.x : a * v.x;
        v.y = v.y > 0.f ? v.y : a * v.y;
        v.z = v.z > 0.f ? v.z : a * v.z;
        v.w = v.w > 0.f ? v.w : a * v.w;
        y[i] = v;
    }
}

__global__ __launch_bounds__(256) void prelu1_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ go,
                                                         const float* __restrict__ slope, long n4, float4* __restrict__ gx,
                                                         float* __restrict__ gslope)
{
    __shared__ float red[4];
    const float a = slope[0];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = x[i], g = go[i];
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x;  acc += v.x > 0.f ? 0.f : v.x * g.x;
        o.y = v.y > 0.f ? g.y : a * g.y;  acc += v.y > 0.f ? 0.f : v.y * g.y;
        o.z = v.z > 0.f ? g.z : a * g.z;  acc += v.z > 0.f ? 0.f : v.z * g.z;
        o.w = v.w > 0.f ? g.w : a * g.w;  acc += v.w > 0.f ? 0.f : v.w * g.w;
        gx[i] = o;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gslope, (red[0] + red[1]) + (red[2] + red[3]));
}

// y = act( x * sa[c] + ba[c]  (+ r * sr[c] + br[c]) ),  c = plane % C, planes x inner, fp32, may run in place.
// Inference-mode BatchNorm (scale/shift folded on the host) + activation (+ residual branch) in ONE pass:
// replaces MIOpenBatchNormFwdInferSpatialEst + clamp / leaky_relu / prelu / add launches (2.0 ms of a 17.4 ms step).
// act: 0 none, 1 relu, 2 leaky relu / prelu with one slope.
template <int ACT, bool HAS_RES, bool RES_AFFINE>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ sa, const float* __restrict__ ba,
                                                         const float* __restrict__ r, const float* __restrict__ sr, const float* __restrict__ br,
                                                         int C, long inner4, float slope, float* __restrict__ y)
{
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const float a = sa[c], b = ba[c];
    float ra = 1.f, rb = 0.f;
    if (HAS_RES && RES_AFFINE) {
        ra = sr[c];
        rb = br[c];
    }
    const float4* xp = reinterpret_cast<const float4*>(x) + plane * inner4;
    const float4* rp = HAS_RES ? reinterpret_cast<const float4*>(r) + plane * inner4 : nullptr;
    float4* yp = reinterpret_cast<float4*>(y) + plane * inner4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < inner4; i += (long)gridDim.x * 256) {
        float4 v = xp[i];
        float o[4] = {v.x * a + b, v.y * a + b, v.z * a + b, v.w * a + b};
        if (HAS_RES) {
            const float4 q = rp[i];
            o[0] += q.x * ra + rb; o[1] += q.y * ra + rb; o[2] += q.z * ra + rb; o[3] += q.w * ra + rb;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == 1) o[j] = fmaxf(o[j], 0.f);
            if (ACT == 2) o[j] = o[j] > 0.f ? o[j] : o[j] * slope;
        }
        yp[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// conv3x3(pad 1) o bilinear-upsample(align_corners) computed from LOW-resolution channel mixes.
// Both operators are linear, and the 3x3 convolution's channel mixing commutes with the (per-channel)
// interpolation, so
//     conv3x3(up(x))[co](y,x) = sum_{tap=(dy,dx)} [ (y+dy,x+dx) inside ] * up( W_tap x )[co](y+dy, x+dx)
// i.e. ONE 1x1 convolution Cin -> 9*Cout at the low resolution (a GEMM with 4x fewer FLOPs than the 3x3
// convolution at 2x resolution) followed by this gather: per output pixel 9 bilinear taps of
// z[b, tap*Cout + co] with zero fill outside the upsampled map, then the folded BatchNorm scale/shift and
// the activation (PSPUpsample = Upsample + Conv3x3 + BN + PReLU, /root/reference/models/cnn/pspnet.py:34-45).
// Exact in real arithmetic; fp32 rounding differs from conv-after-upsample only in summation order.
// Removes the 2x-resolution input tensor entirely (never written, never read).
template <int ACT>
__global__ __launch_bounds__(256) void upconv3x3_gather_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int Cout, int H, int W, int OH, int OW,
                                                               float rh, float rw, float slope, float* __restrict__ out)
{
    const int bc = blockIdx.y;                       // b * Cout + co
    const int b = bc / Cout, co = bc - b * Cout;
    const int qpr = (OW + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpr * OH) return;
    const int oy = q / qpr;
    const int ox0 = (q - oy * qpr) * 4;
    // column parameters of the 6 source columns ox0-1 .. ox0+4
    int cx0[6], cx1[6];
    float clx[6];
    bool cok[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ox = ox0 - 1 + j;
        cok[j] = ox >= 0 && ox < OW;
        const float sx = rw * (float)max(ox, 0);
        const int x0 = min((int)sx, W - 1);
        cx0[j] = x0;
        cx1[j] = x0 + (x0 < W - 1 ? 1 : 0);
        clx[j] = sx - (float)x0;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long plane_sz = (long)H * W;
    const float* zb = z + ((long)b * 9 * Cout + co) * plane_sz;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = oy + dy;
        if (yy < 0 || yy >= OH) continue;            // zero padding of the upsampled map
        const float sy = rh * (float)yy;
        const int y0 = min((int)sy, H - 1);
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float ly = sy - (float)y0, hy = 1.f - ly;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int tap = (dy + 1) * 3 + (dx + 1);
            const float* zp = zb + (long)tap * Cout * plane_sz;
            const float* r0 = zp + (long)y0 * W;
            const float* r1 = zp + (long)y1 * W;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = j + dx + 1;            // index into the 6-column table
                if (cok[c]) {
                    const float lx = clx[c], hx = 1.f - lx;
                    acc[j] += hy * (hx * r0[cx0[c]] + lx * r0[cx1[c]]) + ly * (hx * r1[cx0[c]] + lx * r1[cx1[c]]);
                }
            }
        }
    }
    const float a = scale[co], sh = shift[co];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float o = acc[j] * a + sh;
        if (ACT == 1) o = fmaxf(o, 0.f);
        if (ACT == 2) o = o > 0.f ? o : o * slope;
        v[j] = o;
    }
    float* op = out + ((long)bc * OH + oy) * OW + ox0;
    if (ox0 + 3 < OW && (((uintptr_t)op) & 15) == 0) {
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int j = 0; j < 4 && ox0 + j < OW; ++j) op[j] = v[j];
    }
}

// Transpose of the 9-tap gather, for training: gz[b, tap*Cout+co, y, x] = sum over the output pixels (oy, ox) whose tap-shifted
// sample (oy+dy, ox+dx) reads low-resolution pixel (y, x), of wy * wx * go[b, co, oy, ox].  Gather form (one thread per element
// of gz, the forward's own (y0, y1, ly) arithmetic), no atomics.
__global__ __launch_bounds__(256) void upconv3x3_gather_bwd_kernel(const float* __restrict__ go, int Cout, int H, int W, int OH, int OW,
                                                                   float rh, float rw, float* __restrict__ gz)
{
    __shared__ TapLists L;
    const long plane = blockIdx.y;                       // (b, tap, co)
    const int co = (int)(plane % Cout);
    const int tap = (int)((plane / Cout) % 9);
    const long b = plane / (9L * Cout);
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int tiles_x = (W + TL - 1) / TL;
    const int y0 = (blockIdx.x / tiles_x) * TL, x0 = (blockIdx.x % tiles_x) * TL;
    build_tap_lists(L, y0, x0, H, W, OH, OW, rh, rw, dy, dx);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) return;
    gz[plane * (long)H * W + (long)y * W + x] =
        gather_transpose_tile(L, go + (b * Cout + co) * (long)OH * OW, ty, tx, y, x, H, W, OH, OW, rh, rw, dy, dx);
}

// The same transpose for the x2 stages of PSPUpsample, where the tap lists are short and the nine taps of one (b, co) read the SAME
// window of grad_out: a workgroup stages that window (<= UB_P x UB_P for a 16x16 low-resolution tile) in LDS once, keeps the row /
// column lists of all three shifts, and writes the nine gz planes of its tile from LDS.  The one-plane kernel above re-reads the
// window from L2 nine times with ~16 scattered loads per element (texture-address bound: 1.5 ms per call at 128^2 x 64 x 24).
constexpr int UB_P = 44;

struct TapLists3 {
    int yo[3][TL][TL_MAX], xo[3][TL][TL_MAX];            // window-relative output index
    float yw[3][TL][TL_MAX], xw[3][TL][TL_MAX];
    int yn[3][TL];
    int xmax, overflow;
};

template <int NX>
__device__ __forceinline__ void upconv_bwd_taps(const TapLists3& L, const float (*patch)[UB_P + 1], int ty, int tx, bool live, int Cout, long hw,
                                                float* __restrict__ gzb)
{
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
        int xo[NX];
        float xw[NX];
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            xo[b] = L.xo[dxi][tx][b];
            xw[b] = L.xw[dxi][tx][b];
        }
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
            const int ny = L.yn[dyi][ty];
            float acc = 0.f;
            for (int a = 0; a < ny; ++a) {
                const float* row = patch[L.yo[dyi][ty][a]];
                float rs = 0.f;
#pragma unroll
                for (int b = 0; b < NX; ++b) rs = fmaf(xw[b], row[xo[b]], rs);
                acc = fmaf(L.yw[dyi][ty][a], rs, acc);
            }
            if (live) gzb[(long)((dyi * 3 + dxi) * Cout) * hw] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void upconv3x3_gather_bwd_lds_kernel(const float* __restrict__ go, int Cout, int H, int W, int OH, int OW,
                                                                       float rh, float rw, float* __restrict__ gz)
{
    __shared__ TapLists3 L;
    __shared__ float patch[UB_P][UB_P + 1];
    const int bc = blockIdx.y;                           // b * Cout + co
    const int b = bc / Cout, co = bc - b * Cout;
    const int tiles_x = (W + TL - 1) / TL;
    const int y0 = (blockIdx.x / tiles_x) * TL, x0 = (blockIdx.x % tiles_x) * TL;
    const int t = threadIdx.x;
    // window of grad_out any list entry of this tile can name: src_range of the first / last row, one more for the tap shift
    int lo, hi, tmp;
    src_range(y0, H, OH, rh, lo, tmp);
    src_range(min(y0 + TL, H) - 1, H, OH, rh, tmp, hi);
    const int py0 = max(lo - 1, 0), ph = min(min(hi + 1, OH - 1) - py0 + 1, UB_P);
    src_range(x0, W, OW, rw, lo, tmp);
    src_range(min(x0 + TL, W) - 1, W, OW, rw, tmp, hi);
    const int px0 = max(lo - 1, 0), pw = min(min(hi + 1, OW - 1) - px0 + 1, UB_P);
    if (t == 0) { L.overflow = 0; L.xmax = 0; }
    __syncthreads();
    if (t < 6 * TL) {                                    // (axis, shift, row / column of the tile)
        const bool isy = t < 3 * TL;
        const int u = isy ? t : t - 3 * TL;
        const int si = u / TL, k = u - si * TL, sh = si - 1;
        const int i = (isy ? y0 : x0) + k;
        const int n_in = isy ? H : W, n_out = isy ? OH : OW, p0 = isy ? py0 : px0, pn = isy ? ph : pw;
        const float r = isy ? rh : rw;
        int* oo = isy ? L.yo[si][k] : L.xo[si][k];
        float* ww = isy ? L.yw[si][k] : L.xw[si][k];
        int cnt = 0;
        bool bad = false;
        if (i < n_in) {
            int l2, h2;
            src_range(i, n_in, n_out, r, l2, h2);
            l2 = max(l2, max(0, sh));
            h2 = min(h2, n_out - 1 + min(0, sh));
            for (int o = l2; o <= h2; ++o) {
                const float w = tap_weight(o, i, n_in, r);
                if (w != 0.f) {
                    const int rel = o - sh - p0;
                    if (rel < 0 || rel >= pn) bad = true;
                    if (cnt < TL_MAX) { oo[cnt] = rel; ww[cnt] = w; }
                    ++cnt;
                }
            }
        }
        if (cnt > TL_MAX || bad) atomicOr(&L.overflow, 1);
        for (int c = min(cnt, TL_MAX); c < TL_MAX; ++c) { oo[c] = 0; ww[c] = 0.f; }
        if (isy) L.yn[si][k] = min(cnt, TL_MAX);
        else atomicMax(&L.xmax, cnt);
    }
    const float* gp = go + (long)bc * OH * OW;
    for (int r = t >> 6; r < ph; r += 4) {
        const int c = t & 63;
        if (c < pw) patch[r][c] = gp[(long)(py0 + r) * OW + px0 + c];
    }
    __syncthreads();
    const int ty = t >> 4, tx = t & 15;
    const int y = y0 + ty, x = x0 + tx;
    const bool live = y < H && x < W;
    const long hw = (long)H * W;
    float* gzb = gz + ((long)b * 9 * Cout + co) * hw + (long)min(y, H - 1) * W + min(x, W - 1);
    if (L.overflow) {                                    // (not reached for the x2 stages the host sends here)
        if (live)
            for (int tap = 0; tap < 9; ++tap)
                gzb[(long)(tap * Cout) * hw] = gather_transpose_plain(gp, y, x, H, W, OH, OW, rh, rw, tap / 3 - 1, tap % 3 - 1);
        return;
    }
    if (L.xmax <= 4) upconv_bwd_taps<4>(L, patch, ty, tx, live, Cout, hw, gzb);
    else upconv_bwd_taps<TL_MAX>(L, patch, ty, tx, live, Cout, hw, gzb);
}

// LDS-tiled form of upconv3x3_gather for scale factors <= ~0.5 (the x2 upsampling of PSPUpsample): a workgroup
// owns a 64x16 output tile of one (b, co); the source patch it needs from each of the 9 tap planes
// (<= 12 x 36 floats for rh, rw <= 0.51) is staged once in LDS with coalesced row reads, then every thread
// blends its 4 outputs per tap from LDS.  12x fewer global/L1 loads than the direct kernel.
constexpr int UT_W = 64, UT_H = 16, UP_PH = 12, UP_PW = 36;

template <int ACT>
__global__ __launch_bounds__(256) void upconv3x3_gather_lds_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int Cout, int H, int W, int OH, int OW,
                                                                   float rh, float rw, float slope, float* __restrict__ out)
{
    // one extra row and column (clamped duplicates) so that the "+1" bilinear neighbour always exists in the patch: its
    // weight is 0 whenever it is a duplicate, and the two reads of a row become one ds_read2_b32
    __shared__ float patch[9][UP_PH + 1][UP_PW + 2];
    const int bc = blockIdx.z;
    const int b = bc / Cout, co = bc - b * Cout;
    const int ox_t = blockIdx.x * UT_W, oy_t = blockIdx.y * UT_H;
    // source window covering output rows [oy_t-1, oy_t+UT_H] and columns [ox_t-1, ox_t+UT_W]
    const int ys0 = min((int)(rh * (float)max(oy_t - 1, 0)), H - 1);
    const int ys1 = min((int)(rh * (float)min(oy_t + UT_H, OH - 1)) + 1, H - 1);
    const int xs0 = min((int)(rw * (float)max(ox_t - 1, 0)), W - 1);
    const int xs1 = min((int)(rw * (float)min(ox_t + UT_W, OW - 1)) + 1, W - 1);
    const int ph = ys1 - ys0 + 2, pw = xs1 - xs0 + 2;           // incl. the duplicate row / column
    const long plane_sz = (long)H * W;
    const float* zb = z + ((long)b * 9 * Cout + co) * plane_sz;
    // (r, c) = (i / pw, i % pw) by a 20-bit reciprocal (exact for i < 2^12, 3 <= pw <= 38): no integer division in the fill loop
    const unsigned inv_pw = ((1u << 20) + pw - 1) / pw;
    for (int i = threadIdx.x; i < ph * pw; i += 256) {
        const int r = (int)(((unsigned)i * inv_pw) >> 20);
        const int c = i - r * pw;
        const float* src = zb + (long)min(ys0 + r, H - 1) * W + min(xs0 + c, W - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) patch[tap][r][c] = src[(long)tap * Cout * plane_sz];
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 quads x 16 rows
    const int oy = oy_t + ty, ox0 = ox_t + tx * 4;
    if (oy >= OH || ox0 >= OW) return;
    int cx0[6];
    float clx[6], chx[6];
    bool cok[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ox = ox0 - 1 + j;
        cok[j] = ox >= 0 && ox < OW;
        const float sx = rw * (float)min(max(ox, 0), OW - 1);
        const int x0 = min((int)sx, W - 1);
        cx0[j] = x0 - xs0;
        clx[j] = sx - (float)x0;
        chx[j] = 1.f - clx[j];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = oy + dy;
        if (yy < 0 || yy >= OH) continue;
        const float sy = rh * (float)yy;
        const int y0 = min((int)sy, H - 1);
        const float ly = sy - (float)y0, hy = 1.f - ly;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int tap = (dy + 1) * 3 + (dx + 1);
            const float* r0 = &patch[tap][y0 - ys0][0];
            const float* r1 = r0 + (UP_PW + 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = j + dx + 1;
                if (cok[c] && ox0 + j < OW) {
                    // explicit FMAs (the build runs with -ffp-contract=off): 6 VALU ops per tap and output instead of 10;
                    // the kernel is VALU-bound (9 taps x 4 corners per output)
                    const float lx = clx[c], hx = chx[c];
                    const int xo = cx0[c];
                    const float top = fmaf(lx, r0[xo + 1], hx * r0[xo]);
                    const float bot = fmaf(lx, r1[xo + 1], hx * r1[xo]);
                    acc[j] = fmaf(ly, bot, fmaf(hy, top, acc[j]));
                }
            }
        }
    }
    const float a = scale[co], sh = shift[co];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float o = acc[j] * a + sh;
        if (ACT == 1) o = fmaxf(o, 0.f);
        if (ACT == 2) o = o > 0.f ? o : o * slope;
        v[j] = o;
    }
    float* op = out + ((long)bc * OH + oy) * OW + ox0;
    if (ox0 + 3 < OW && (((uintptr_t)op) & 15) == 0) {
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int j = 0; j < 4 && ox0 + j < OW; ++j) op[j] = v[j];
    }
}

// EIGHT output channels per workgroup (Cout % 8 == 0): a 32 x 8 output tile, thread = one output pixel, the interpolation
// coefficients of a pixel computed once for its eight channels (the one-channel form above is VALU-bound on exactly those), and the
// result ALSO written as the packed split-bf16 operand of the next GEMM over the map (conv_pack_act_kernel's layout: the eight channels
// of a pixel = one 16-byte group in the hi plane and one in the lo plane): the pack launch that followed (36 us in front of the p2r GEMM
// of the first up stage, on the image branch's critical path) is gone.  Same taps, same blend expression, same order: same bits.
constexpr int U8_W = 32, U8_H = 8, U8_PH = 8, U8_PW = 20;

template <int ACT>
__global__ __launch_bounds__(256) void upconv3x3_gather8_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int Cout, int H, int W, int OH, int OW,
                                                                float rh, float rw, float slope, float* __restrict__ out,
                                                                unsigned char* __restrict__ ypk)
{
    // the eight channels in two passes of four: 28 KB of LDS per workgroup (five workgroups per CU) instead of 57 KB (two)
    __shared__ float patch[4][9][U8_PH + 1][U8_PW + 2];
    const int groups = Cout / 8;
    const int b = blockIdx.z / groups, c0 = (blockIdx.z - b * groups) * 8;
    const int ox_t = blockIdx.x * U8_W, oy_t = blockIdx.y * U8_H;
    const int ys0 = min((int)(rh * (float)max(oy_t - 1, 0)), H - 1);
    const int ys1 = min((int)(rh * (float)min(oy_t + U8_H, OH - 1)) + 1, H - 1);
    const int xs0 = min((int)(rw * (float)max(ox_t - 1, 0)), W - 1);
    const int xs1 = min((int)(rw * (float)min(ox_t + U8_W, OW - 1)) + 1, W - 1);
    const int ph = ys1 - ys0 + 2, pw = xs1 - xs0 + 2;
    const long plane_sz = (long)H * W;
    const unsigned inv_pw = ((1u << 20) + pw - 1) / pw;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int oy = oy_t + ty, ox = ox_t + tx;
    const bool live = oy < OH && ox < OW;
    float acc[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) acc[ch] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float* zb = z + ((long)b * 9 * Cout + c0 + 4 * pass) * plane_sz;
        if (pass) __syncthreads();                                  // the first four channels' patch has been read
        for (int i = threadIdx.x; i < ph * pw; i += 256) {
            const int r = (int)(((unsigned)i * inv_pw) >> 20);
            const int c = i - r * pw;
            const float* src = zb + (long)min(ys0 + r, H - 1) * W + min(xs0 + c, W - 1);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) patch[ch][tap][r][c] = src[((long)tap * Cout + ch) * plane_sz];
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = oy + dy;
                if (yy < 0 || yy >= OH) continue;
                const float sy = rh * (float)yy;
                const int y0 = min((int)sy, H - 1);
                const float ly = sy - (float)y0, hy = 1.f - ly;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = ox + dx;
                    if (xx < 0 || xx >= OW) continue;
                    const int tap = (dy + 1) * 3 + (dx + 1);
                    const float sx = rw * (float)xx;
                    const int x0 = min((int)sx, W - 1);
                    const float lx = sx - (float)x0, hx = 1.f - lx;
                    const int xo = x0 - xs0, yo = y0 - ys0;
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const float* r0 = &patch[ch][tap][yo][0];
                        const float* r1 = r0 + (U8_PW + 2);
                        const float top = fmaf(lx, r0[xo + 1], hx * r0[xo]);
                        const float bot = fmaf(lx, r1[xo + 1], hx * r1[xo]);
                        acc[4 * pass + ch] = fmaf(ly, bot, fmaf(hy, top, acc[4 * pass + ch]));
                    }
                }
            }
        }
    }
    if (!live) return;
    float v[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        float o = acc[ch] * scale[c0 + ch] + shift[c0 + ch];
        if (ACT == 1) o = fmaxf(o, 0.f);
        if (ACT == 2) o = o > 0.f ? o : o * slope;
        v[ch] = o;
        out[(((long)b * Cout + c0 + ch) * OH + oy) * OW + ox] = o;
    }
    if (ypk) {
        const long pplane = (long)(OH + 2) * (OW + 2);
        const int nchunk = (Cout + 127) / 128, chunk = c0 / 128, q = (c0 % 128) / 8;
        unsigned char* o = ypk + ((((long)b * nchunk + chunk) * 32 + q) * pplane + (long)(oy + 1) * (OW + 2) + ox + 1) * 16;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gdm_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 16 * pplane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// Pyramid-pooling bottleneck without the 2560-channel concat (pspnet.py:24-31):
//   relu(W . cat(up(p1), up(p2), up(p3), up(p6), f) + b) = relu(W_f f + b + sum_s up(W_s p_s))
// (1x1 convolution commutes with bilinear interpolation).  g = W_f f is a GEMM with K = 512 instead of 2560;
// the four W_s p_s live at 1x1..6x6 and this kernel adds their interpolations, the bias and the ReLU.
struct PspMaps {
    const float* y[4];      // f32[B, C, s, s]
    int s[4];
    float sy[4], sx[4];     // align_corners scale factors (s-1)/(H-1), (s-1)/(W-1)
};

__global__ __launch_bounds__(256) void psp_combine_kernel(const float* __restrict__ g, PspMaps maps, const float* __restrict__ bias,
                                                          int C, int H, int W, float* __restrict__ out)
{
    const long plane = blockIdx.y;                    // b * C + c
    const int c = (int)(plane % C);
    const int hw = H * W;
    const float bc = bias ? bias[c] : 0.f;
    // the plane's four prior maps (1 + 4 + 9 + 36 floats for the reference's bin sizes) live in LDS; the align_corners scale
    // factors come from the host: eight float divisions per thread were most of a one-element-per-thread kernel
    __shared__ float pm[4][64];
    {
        const int k = threadIdx.x >> 6, j = threadIdx.x & 63;
        if (j < maps.s[k] * maps.s[k]) pm[k][j] = maps.y[k][plane * maps.s[k] * maps.s[k] + j];
        __syncthreads();
    }
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < hw; i0 += gridDim.x * 1024) {
        int oy = i0 / W, ox = i0 - oy * W;
        float v[4];
        const bool vec = i0 + 3 < hw;
        if (vec) {
            const float4 gv = *reinterpret_cast<const float4*>(g + plane * hw + i0);     // hw % 4 == 0 (checked on the host)
            v[0] = gv.x; v[1] = gv.y; v[2] = gv.z; v[3] = gv.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i0 + e < hw) {
                float acc = (vec ? v[e] : g[plane * hw + i0 + e]) + bc;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int S = maps.s[k];
                    const float* m = &pm[k][0];
                    const float fy = maps.sy[k] * (float)oy, fx = maps.sx[k] * (float)ox;
                    const int y0 = min((int)fy, S - 1), x0 = min((int)fx, S - 1);
                    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
                    const float ly = fy - (float)y0, lx = fx - (float)x0;
                    acc += (1.f - ly) * ((1.f - lx) * m[y0 * S + x0] + lx * m[y0 * S + x1]) + ly * ((1.f - lx) * m[y1 * S + x0] + lx * m[y1 * S + x1]);
                }
                v[e] = fmaxf(acc, 0.f);
            }
            if (++ox == W) { ox = 0; ++oy; }
        }
        if (vec) *reinterpret_cast<float4*>(out + plane * hw + i0) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int e = 0; e < 4 && i0 + e < hw; ++e) out[plane * hw + i0 + e] = v[e];
    }
}

// The same for EIGHT planes per workgroup (C % 8 == 0, W % 4 == 0): the interpolation coordinates of a pixel are computed once for its
// eight channels, and the result is ALSO written as the packed split-bf16 operand of the next GEMM over the map (conv_pack_act_kernel's
// layout: the eight channels of a pixel are one 16-byte group in the hi plane and one in the lo plane) -- the pack launch that followed
// this kernel (33 us on the image branch's critical path) is gone.
__global__ __launch_bounds__(256) void psp_combine8_kernel(const float* __restrict__ g, PspMaps maps, const float* __restrict__ bias,
                                                           int C, int H, int W, float* __restrict__ out, unsigned char* __restrict__ ypk)
{
    const long plane0 = (long)blockIdx.y * 8;            // b * C + c0
    const int b = (int)(plane0 / C), c0 = (int)(plane0 - (long)b * C);
    const int hw = H * W;
    __shared__ float pm[8][4][64];
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += 256) {
        const int j = i & 63, k = (i >> 6) & 3, ch = i >> 8;
        const int ss = maps.s[k] * maps.s[k];
        if (j < ss) pm[ch][k][j] = maps.y[k][(plane0 + ch) * ss + j];
    }
    __syncthreads();
    float bc[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) bc[ch] = bias ? bias[c0 + ch] : 0.f;
    const long pplane = (long)(H + 2) * (W + 2);
    const int nchunk = (C + 127) / 128, chunk = c0 / 128, q = (c0 % 128) / 8;
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < hw; i0 += gridDim.x * 1024) {
        const int oy = i0 / W, ox0 = i0 - oy * W;                      // four pixels of one row (W % 4 == 0)
        float v[8][4];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const float4 gv = *reinterpret_cast<const float4*>(g + (plane0 + ch) * hw + i0);
            v[ch][0] = gv.x + bc[ch]; v[ch][1] = gv.y + bc[ch]; v[ch][2] = gv.z + bc[ch]; v[ch][3] = gv.w + bc[ch];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = ox0 + e;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int S = maps.s[k];
                const float fy = maps.sy[k] * (float)oy, fx = maps.sx[k] * (float)ox;
                const int y0 = min((int)fy, S - 1), x0 = min((int)fx, S - 1);
                const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
                const float ly = fy - (float)y0, lx = fx - (float)x0;
                const int o00 = y0 * S + x0, o01 = y0 * S + x1, o10 = y1 * S + x0, o11 = y1 * S + x1;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const float* m = &pm[ch][k][0];                     // the same expression, in the same order, as psp_combine_kernel
                    v[ch][e] += (1.f - ly) * ((1.f - lx) * m[o00] + lx * m[o01]) + ly * ((1.f - lx) * m[o10] + lx * m[o11]);
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[ch][e] = fmaxf(v[ch][e], 0.f);
            *reinterpret_cast<float4*>(out + (plane0 + ch) * hw + i0) = make_float4(v[ch][0], v[ch][1], v[ch][2], v[ch][3]);
        }
        if (ypk) {
            unsigned char* o = ypk + ((((long)b * nchunk + chunk) * 32 + q) * pplane + (long)(oy + 1) * (W + 2) + ox0 + 1) * 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) gdm_split2(v[2 * i][e], v[2 * i + 1][e], hi[i], lo[i]);
                *reinterpret_cast<uint4*>(o + e * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(o + e * 16 + 16 * pplane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
    }
}

// y[b,c,j] = act(scale[c] * (x[b,c,j] + t[b,c,idx[b,j]]) + shift[c]) : the point->pixel fusion layers
// conv1x1(cat(rgb, nearest_interp(p))) + BN + ReLU (ffb6d.py:216-222,252-258) with the point half of the
// convolution done at the (few) points and gathered afterwards (a 1x1 convolution commutes with a gather).
template <int ACT>
__global__ __launch_bounds__(256) void gather_add_affine_act_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                                    const int32_t* __restrict__ idx, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int C, int n, int m, float slope,
                                                                    float* __restrict__ y,
                                                                    // optional: the result ALSO as the packed split-bf16 operand of the next
                                                                    // convolution / GEMM over the [B, C, m / W, W] map (gdm_conv.hip
                                                                    // conv_pack_act_kernel's layout): a block's 8 channels are one 16-byte
                                                                    // operand group, so the pack launch that would follow is free here
                                                                    unsigned char* __restrict__ ypk, int W)
{
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 8;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    int src = idx[(long)b * m + j];
    src = min(max(src, 0), n - 1);
    const int cend = min(c0 + 8, C);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c0 + i;
        if (c >= cend) break;
        const long row = (long)b * C + c;
        float o = scale[c] * (x[row * m + j] + t[row * n + src]) + shift[c];
        if (ACT == 1) o = fmaxf(o, 0.f);
        if (ACT == 2) o = o > 0.f ? o : o * slope;
        y[row * m + j] = o;
        v[i] = o;
    }
    if (ypk) {
        const int H = m / W, yy = j / W, xx = j - yy * W;
        const long plane = (long)(H + 2) * (W + 2);
        const int nchunk = (C + 127) / 128, chunk = c0 / 128, q = (c0 % 128) / 8;
        unsigned char* o = ypk + ((((long)b * nchunk + chunk) * 32 + q) * plane + (long)(yy + 1) * (W + 2) + xx + 1) * 16;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gdm_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 16 * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// The same fusion tail with the pixel half of the convolution inside: y[b,co,j] = act(scale[co] * (sum_ci W[co,ci] x[b,ci,j]
// + t[b,co,idx[b,j]]) + shift[co]) for the 64-channel levels (ffb6d.py:216-222 at ds stage 0, :252-258 at up stages 1-2), where
// the GEMM has K = 64: hipBLASLt's f32 kernel runs it at ~0.5 TB/s (62 us per call) and the intermediate costs a write + a read.
// One pass, thread = pixel, 64 accumulators in registers, W^T rows as wave-uniform scalar loads; exact fp32 FMAs.
// PM: y is written pixel-major, f32[B, m, C] (256 contiguous bytes per pixel) -- the layout the sampled-pixel final stage reads.
template <int C, int ACT, bool PM = false>
__global__ __launch_bounds__(256) void conv1x1_gather_add_act_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                     const float* __restrict__ t, const int32_t* __restrict__ idx,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     int n, long m, float slope, float* __restrict__ y)
{
    const int b = blockIdx.y;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    int src = idx[(long)b * m + j];
    src = min(max(src, 0), n - 1);
    const float* xb = x + (long)b * C * m + j;
    const float* tb = t + (long)b * C * n + src;
    float acc[C];
#pragma unroll
    for (int co = 0; co < C; ++co) acc[co] = tb[(long)co * n];
#pragma unroll 4
    for (int ci = 0; ci < C; ++ci) {
        const float xv = xb[(long)ci * m];
#pragma unroll
        for (int co = 0; co < C; ++co) acc[co] = fmaf(wt[ci * C + co], xv, acc[co]);
    }
    float* yb = PM ? y + ((long)b * m + j) * C : y + (long)b * C * m + j;
#pragma unroll
    for (int co = 0; co < C; ++co) {
        float o = scale[co] * acc[co] + shift[co];
        if (ACT == 1) o = fmaxf(o, 0.f);
        if (ACT == 2) o = o > 0.f ? o : o * slope;
        if (PM) acc[co] = o; else yb[(long)co * m] = o;
    }
    if (PM) {
#pragma unroll
        for (int co = 0; co < C; co += 4) *reinterpret_cast<float4*>(yb + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
    }
}

// `final` = Conv2d(64,64,1) + LogSoftmax(dim=1) (pspnet.py:108-112), applied at 128x128 and 256x256 (ffb6d.py:79-80):
// one pass instead of a GEMM + bias + spatial-softmax (313 + 136 us at 256x256, batch 16).  HBM-bound (read C, write C
// floats per pixel); a thread owns one pixel, keeps the C output channels in registers, weights arrive as wave-uniform
// scalar loads, log-softmax stays in-thread.  Exact fp32 FMAs.
template <int C>
__global__ __launch_bounds__(256) void conv1x1_logsoftmax_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                 const float* __restrict__ bias, long hw, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float* xb = x + (long)b * C * hw + p;
    float y[C];
#pragma unroll
    for (int co = 0; co < C; ++co) y[co] = bias ? bias[co] : 0.f;
#pragma unroll 4
    for (int ci = 0; ci < C; ++ci) {
        const float xv = xb[(long)ci * hw];
#pragma unroll
        for (int co = 0; co < C; ++co) y[co] = fmaf(wt[ci * C + co], xv, y[co]);   // wt = W^T: 64 contiguous scalars per ci
    }
    float m = y[0];
#pragma unroll
    for (int co = 1; co < C; ++co) m = fmaxf(m, y[co]);
    float ssum = 0.f;
#pragma unroll
    for (int co = 0; co < C; ++co) ssum += __expf(y[co] - m);   // v_exp_f32 on (y - m) log2(e): arguments <= 0, relative error ~1e-6 of terms
                                                                // that sum to >= 1 (the precise expf was a third of this ALU-bound kernel)
    const float lse = m + logf(ssum);
    float* ob = out + (long)b * C * hw + p;
#pragma unroll
    for (int co = 0; co < C; ++co) ob[(long)co * hw] = y[co] - lse;
}

// The four adaptive average pools of the pyramid-pooling module (1,2,3,6 bins; pspnet.py:17-20) in one pass over the
// feature map: one workgroup per plane accumulates the 6x6-bin... every bin size separately with PyTorch's bin edges
// (start = floor(i*H/s), end = ceil((i+1)*H/s)); outputs f32[planes, s*s] each.
__global__ __launch_bounds__(256) void psp_pools_kernel(const float* __restrict__ x, int H, int W,
                                                        float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3,
                                                        float* __restrict__ o6)
{
    __shared__ float tile[64 * 64];
    const long plane = blockIdx.x;
    const int hw = H * W;
    const float* xp = x + plane * hw;
    for (int i = threadIdx.x; i < hw; i += 256) tile[i] = xp[i];
    __syncthreads();
    // 4 + 9 + 36 bins summed by runs of consecutive lanes (a power of two each: a shuffle reduction finishes the bin), sized so that no
    // lane walks more than 18 elements of a 32 x 32 map: 4x16 | 9x8 | 36x2 = 208 threads; the single 1 x 1 bin is the sum of the four
    // 2 x 2 bins (they partition the map when H and W are even; otherwise 64 lanes walk it).  Round 3 gave the 36 smallest bins one
    // lane each: 36 dependent LDS adds were the workgroup's critical path.
    __shared__ float quad[4];
    const int t = threadIdx.x;
    const bool even = (H % 2 == 0) && (W % 2 == 0);
    int s, bi, gsz, gl;
    float* o;
    if (t < 64) { s = 2; bi = t >> 4; gsz = 16; gl = t & 15; o = o2 + plane * 4; }
    else if (t < 136) { s = 3; bi = (t - 64) >> 3; gsz = 8; gl = (t - 64) & 7; o = o3 + plane * 9; }
    else if (t < 208) { s = 6; bi = (t - 136) >> 1; gsz = 2; gl = (t - 136) & 1; o = o6 + plane * 36; }
    else { s = 1; bi = 0; gsz = 1; gl = 0; o = nullptr; }
    const int by = bi / s, bx = bi - by * s;
    const int y0 = (by * H) / s, y1 = ((by + 1) * H + s - 1) / s;
    const int x0 = (bx * W) / s, x1 = ((bx + 1) * W + s - 1) / s;
    const int bw = x1 - x0, cnt = (y1 - y0) * bw;
    float acc = 0.f;
    if (o) {
        // element i = gl, gl + gsz, ... of the bin in row-major order, walked without a division per element
        int ry = gl / bw, rx = gl - ry * bw;
        for (int i = gl; i < cnt; i += gsz) {
            acc += tile[(y0 + ry) * W + x0 + rx];
            rx += gsz;
            while (rx >= bw) { rx -= bw; ++ry; }
        }
    }
    for (int m = 1; m < 64; m <<= 1) {
        const float other = __shfl_xor(acc, m, 64);
        if (m < gsz) acc += other;
    }
    if (o && gl == 0) o[bi] = acc / (float)cnt;
    if (even) {
        if (t < 64 && gl == 0) quad[bi] = acc;
        __syncthreads();
        if (t == 0) o1[plane] = ((quad[0] + quad[1]) + (quad[2] + quad[3])) / (float)hw;
    } else {
        __syncthreads();                                            // (every lane is done with its bin)
        if (t < 64) {
            float a1 = 0.f;
            for (int i = t; i < hw; i += 64) a1 += tile[i];
            for (int m = 1; m < 64; m <<= 1) a1 += __shfl_xor(a1, m, 64);
            if (t == 0) o1[plane] = a1 / (float)hw;
        }
    }
}

// Backward of the four pools in one pass: gin[y][x] = sum over the bins (of every size) that contain (y, x) of grad_bin / bin area.
// Replaces four adaptive_avg_pool2d backward launches (three of them atomic scatter kernels, 0.5 ms each on 24 x 512 planes of 32 x 32)
// and the three full-size additions that merge their results.
__global__ __launch_bounds__(256) void psp_pools_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3,
                                                            const float* __restrict__ g6, int H, int W, float* __restrict__ gin)
{
    __shared__ float gs[50];                             // 1 | 4 | 9 | 36 bins, already divided by their areas
    const long plane = blockIdx.x;
    const int t = threadIdx.x;
    if (t < 50) {
        int s, bi;
        const float* gsrc;
        if (t < 1) { s = 1; bi = 0; gsrc = g1 + plane; }
        else if (t < 5) { s = 2; bi = t - 1; gsrc = g2 + plane * 4; }
        else if (t < 14) { s = 3; bi = t - 5; gsrc = g3 + plane * 9; }
        else { s = 6; bi = t - 14; gsrc = g6 + plane * 36; }
        const int by = bi / s, bx = bi - by * s;
        const int cnt = (((by + 1) * H + s - 1) / s - (by * H) / s) * (((bx + 1) * W + s - 1) / s - (bx * W) / s);
        gs[t] = gsrc[bi] / (float)cnt;
    }
    __syncthreads();
    const int hw = H * W;
    for (int p = t; p < hw; p += 256) {
        const int y = p / W, x = p - y * W;
        float acc = 0.f;
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 3 : 6;
            for (int by = 0; by < s; ++by) {
                if (y < (by * H) / s || y >= ((by + 1) * H + s - 1) / s) continue;
                for (int bx = 0; bx < s; ++bx)
                    if (x >= (bx * W) / s && x < ((bx + 1) * W + s - 1) / s) acc += gs[base + by * s + bx];
            }
            base += s * s;
        }
        gin[plane * hw + p] = acc;
    }
}

// depth (metres) -> camera-frame xyz for a crop window of a frame, as dpt_2_pcld + the integer crop of the loader
// (/root/reference/datasets/lm/linemod_pbr.py:398-411,473): x = (u - cx) d / fx, y = (v - cy) d / fy, z = d, zeros where d <= 1e-8.
// depth f32[B,H,W]; K f32[B,3,3]; crop origin (x0,y0) i32[B,2]; out f32[B,S,S,3].
__global__ __launch_bounds__(256) void depth_to_xyz_kernel(const float* __restrict__ depth, const float* __restrict__ K,
                                                           const int32_t* __restrict__ origin, int H, int W, int S, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int v = origin[2 * b + 1] + i / S, u = origin[2 * b] + i % S;
    float d = 0.f;
    if (u >= 0 && u < W && v >= 0 && v < H) d = depth[((long)b * H + v) * W + u];
    const float* k = K + b * 9;
    const float fx = k[0], cx = k[2], fy = k[4], cy = k[5];
    // dpt_2_pcld forms these in float64 (int64 pixel maps minus a float32 intrinsic promote to double in numpy) and the loader
    // rounds to float32 once, at the end (linemod_pbr.py:398-411,573): same here, so the crop is bit-identical to the reference's.
    const double m = d > 1e-8f ? 1.0 : 0.0;
    float* o = out + ((long)b * S * S + i) * 3;
    o[0] = (float)(((double)u - (double)cx) * (double)d / (double)fx * m);
    o[1] = (float)(((double)v - (double)cy) * (double)d / (double)fy * m);
    o[2] = (float)((double)d * m);
}

inline float scale_ac(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

} // namespace

extern "C" int gdm_upsample_bilinear_hip(const float* in, long planes, int H, int W, int OH, int OW, float* out, void* stream)
{
    GDM_CHECK_ARG(in && out, "gdm_upsample_bilinear_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && planes <= 65535L * 32768 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1,
                  "gdm_upsample_bilinear_hip: bad shape planes=%ld %dx%d -> %dx%d", planes, H, W, OH, OW);
    // planes ride blockIdx.y (<= 65535): fold the excess into repeated launches
    const long zmax = 65535;
    const int quads = ((OW + 3) / 4) * OH;
    for (long p0 = 0; p0 < planes; p0 += zmax) {
        const long np = planes - p0 < zmax ? planes - p0 : zmax;
        dim3 grid(gdm_cdiv(quads, 256), (unsigned)np);
        hipLaunchKernelGGL(upsample_bilinear_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                           in + p0 * H * W, H, W, OH, OW, scale_ac(H, OH), scale_ac(W, OW), out + p0 * OH * OW);
    }
    return gdm_launch_status("upsample_bilinear_kernel");
}

extern "C" int gdm_upsample_bilinear_bwd_hip(const float* grad_out, long planes, int H, int W, int OH, int OW, float* grad_in, void* stream)
{
    GDM_CHECK_ARG(grad_out && grad_in, "gdm_upsample_bilinear_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1 && OH <= 65535, "gdm_upsample_bilinear_bwd_hip: bad shape");
    if (H <= USM_IN && W <= USM_IN && OH <= USM_OW && OW <= USM_OW && OH * OW <= USM_OUT && planes <= 0x7fffffffL) {
        hipLaunchKernelGGL(upsample_bilinear_bwd_small_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, grad_out, H, W, OH, OW,
                           scale_ac(H, OH), scale_ac(W, OW), grad_in);
        return gdm_launch_status("upsample_bilinear_bwd_small_kernel");
    }
    const long zmax = 65535;
    for (long p0 = 0; p0 < planes; p0 += zmax) {
        const long np = planes - p0 < zmax ? planes - p0 : zmax;
        dim3 grid(gdm_cdiv(H, TL) * gdm_cdiv(W, TL), (unsigned)np);
        hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                           grad_out + p0 * OH * OW, H, W, OH, OW, scale_ac(H, OH), scale_ac(W, OW), grad_in + p0 * H * W);
    }
    return gdm_launch_status("upsample_bilinear_bwd_kernel");
}

extern "C" int gdm_affine_act_maxk_hip(const float* x, const float* scale, const float* shift, long planes, int C, long n, int K, int act,
                                       float slope, float* out, void* stream)
{
    GDM_CHECK_ARG(x && scale && shift && out, "gdm_affine_act_maxk_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && planes <= 65535 && C >= 1 && n >= 1 && K >= 4 && K % 4 == 0 && act >= 0 && act <= 2,
                  "gdm_affine_act_maxk_hip: planes=%ld C=%d n=%ld K=%d (K %% 4 == 0, planes <= 65535)", planes, C, n, K);
    GDM_CHECK_ARG(((uintptr_t)x & 15) == 0, "gdm_affine_act_maxk_hip: x must be 16-byte aligned");
    dim3 grid(gdm_cdiv(n, 256), (unsigned)planes);
    hipStream_t s = (hipStream_t)stream;
    if (act == 0) hipLaunchKernelGGL(affine_act_maxk_kernel<0>, grid, dim3(256), 0, s, x, scale, shift, C, n, K / 4, slope, out);
    else if (act == 1) hipLaunchKernelGGL(affine_act_maxk_kernel<1>, grid, dim3(256), 0, s, x, scale, shift, C, n, K / 4, slope, out);
    else hipLaunchKernelGGL(affine_act_maxk_kernel<2>, grid, dim3(256), 0, s, x, scale, shift, C, n, K / 4, slope, out);
    return gdm_launch_status("affine_act_maxk_kernel");
}

extern "C" int gdm_prelu1_hip(const float* x, const float* slope, long n, float* y, void* stream)
{
    GDM_CHECK_ARG(x && slope && y, "gdm_prelu1_hip: NULL pointer");
    GDM_CHECK_ARG(n >= 4 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "gdm_prelu1_hip: n=%ld must be a multiple of 4 and the pointers 16-byte aligned", n);
    long blocks = gdm_cdiv(n / 4, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(prelu1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, slope, n / 4, (float4*)y);
    return gdm_launch_status("prelu1_fwd_kernel");
}

extern "C" int gdm_prelu1_bwd_hip(const float* x, const float* grad_out, const float* slope, long n, float* grad_x, float* grad_slope, void* stream)
{
    GDM_CHECK_ARG(x && grad_out && slope && grad_x && grad_slope, "gdm_prelu1_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(n >= 4 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)grad_out | (uintptr_t)grad_x) & 15) == 0, "gdm_prelu1_bwd_hip: n=%ld must be a multiple of 4 and the pointers 16-byte aligned", n);
    long blocks = gdm_cdiv(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prelu1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)grad_out, slope,
                       n / 4, (float4*)grad_x, grad_slope);
    return gdm_launch_status("prelu1_bwd_kernel");
}

extern "C" int gdm_affine_act_hip(const float* x, const float* scale, const float* shift, const float* res, const float* res_scale,
                                  const float* res_shift, long planes, int C, long inner, int act, float slope, float* y, void* stream)
{
    GDM_CHECK_ARG(x && scale && shift && y, "gdm_affine_act_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && planes <= 65535 && C >= 1 && inner >= 4 && inner % 4 == 0, "gdm_affine_act_hip: planes=%ld C=%d inner=%ld (inner %% 4 == 0, planes <= 65535)", planes, C, inner);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_affine_act_hip: act=%d", act);
    GDM_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "gdm_affine_act_hip: pointers must be 16-byte aligned");
    const long inner4 = inner / 4;
    int gx = gdm_cdiv(inner4, 256);
    if (gx > 64) gx = 64;
    dim3 grid(gx, (unsigned)planes);
    hipStream_t s = (hipStream_t)stream;
#define AA(A, H, R) hipLaunchKernelGGL((affine_act_kernel<A, H, R>), grid, dim3(256), 0, s, x, scale, shift, res, res_scale, res_shift, C, inner4, slope, y)
    const bool has = res != nullptr, aff = res_scale != nullptr && res_shift != nullptr;
    if (act == 0) { if (!has) AA(0, false, false); else if (aff) AA(0, true, true); else AA(0, true, false); }
    else if (act == 1) { if (!has) AA(1, false, false); else if (aff) AA(1, true, true); else AA(1, true, false); }
    else { if (!has) AA(2, false, false); else if (aff) AA(2, true, true); else AA(2, true, false); }
#undef AA
    return gdm_launch_status("affine_act_kernel");
}

extern "C" int gdm_upconv3x3_gather2_hip(const float* z, const float* scale, const float* shift, int B, int Cout, int H, int W,
                                         int OH, int OW, int act, float slope, float* out, void* outpk, void* stream)
{
    GDM_CHECK_ARG(z && scale && shift && out && outpk, "gdm_upconv3x3_gather2_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && Cout >= 8 && Cout % 8 == 0 && (Cout == 64 || Cout % 128 == 0) && (long)B * (Cout / 8) <= 65535 && H >= 1 && W >= 1
                  && OH >= 1 && OW >= 1, "gdm_upconv3x3_gather2_hip: bad shape B=%d Cout=%d %dx%d -> %dx%d (Cout = 64 or a multiple of 128)",
                  B, Cout, H, W, OH, OW);
    GDM_CHECK_ARG(act >= 0 && act <= 2 && ((uintptr_t)outpk & 15) == 0, "gdm_upconv3x3_gather2_hip: act=%d / unaligned packed output", act);
    const float rh = scale_ac(H, OH), rw = scale_ac(W, OW);
    GDM_CHECK_ARG((int)(rh * (U8_H + 1)) + 3 <= U8_PH && (int)(rw * (U8_W + 1)) + 3 <= U8_PW,
                  "gdm_upconv3x3_gather2_hip: scale factors %g x %g too large for the LDS tile (the x2 stages of PSPUpsample fit)", rh, rw);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(gdm_cdiv(OW, U8_W), gdm_cdiv(OH, U8_H), B * (Cout / 8));
    if (act == 0) hipLaunchKernelGGL(upconv3x3_gather8_kernel<0>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out, (unsigned char*)outpk);
    else if (act == 1) hipLaunchKernelGGL(upconv3x3_gather8_kernel<1>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out, (unsigned char*)outpk);
    else hipLaunchKernelGGL(upconv3x3_gather8_kernel<2>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out, (unsigned char*)outpk);
    return gdm_launch_status("upconv3x3_gather8_kernel");
}

extern "C" int gdm_upconv3x3_gather_hip(const float* z, const float* scale, const float* shift, int B, int Cout, int H, int W,
                                        int OH, int OW, int act, float slope, float* out, void* stream)
{
    GDM_CHECK_ARG(z && scale && shift && out, "gdm_upconv3x3_gather_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && Cout >= 1 && (long)B * Cout <= 65535 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1,
                  "gdm_upconv3x3_gather_hip: bad shape B=%d Cout=%d %dx%d -> %dx%d", B, Cout, H, W, OH, OW);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_upconv3x3_gather_hip: act=%d", act);
    hipStream_t s = (hipStream_t)stream;
    const float rh = scale_ac(H, OH), rw = scale_ac(W, OW);
    // worst-case source window of a tile: (UT+1) output steps of size r, plus the +1 neighbour, plus rounding
    const bool lds_ok = (int)(rh * (UT_H + 1)) + 3 <= UP_PH && (int)(rw * (UT_W + 1)) + 3 <= UP_PW && (long)B * Cout <= 65535;
    if (lds_ok) {
        dim3 grid(gdm_cdiv(OW, UT_W), gdm_cdiv(OH, UT_H), B * Cout);
        if (act == 0) hipLaunchKernelGGL(upconv3x3_gather_lds_kernel<0>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
        else if (act == 1) hipLaunchKernelGGL(upconv3x3_gather_lds_kernel<1>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
        else hipLaunchKernelGGL(upconv3x3_gather_lds_kernel<2>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
        return gdm_launch_status("upconv3x3_gather_lds_kernel");
    }
    // any other scale factor: the direct form (reads its taps from global memory).  (The three launch lines below were lost when the
    // LDS form was added: the function then returned success without having written `out` -- unreachable from the model, whose
    // stages all upsample x2, but wrong for a caller of the C entry point.  tests/test_gpu_ops.py covers a 16 -> 20 stage now.)
    const int quads = ((OW + 3) / 4) * OH;
    dim3 grid(gdm_cdiv(quads, 256), B * Cout);
    if (act == 0) hipLaunchKernelGGL(upconv3x3_gather_kernel<0>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
    else if (act == 1) hipLaunchKernelGGL(upconv3x3_gather_kernel<1>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
    else hipLaunchKernelGGL(upconv3x3_gather_kernel<2>, grid, dim3(256), 0, s, z, scale, shift, Cout, H, W, OH, OW, rh, rw, slope, out);
    return gdm_launch_status("upconv3x3_gather_kernel");
}

extern "C" int gdm_upconv3x3_gather_bwd_hip(const float* grad_out, int B, int Cout, int H, int W, int OH, int OW, float* grad_z, void* stream)
{
    GDM_CHECK_ARG(grad_out && grad_z, "gdm_upconv3x3_gather_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && Cout >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1 && (long)B * 9 * Cout <= 65535,
                  "gdm_upconv3x3_gather_bwd_hip: bad shape B=%d Cout=%d (B*9*Cout <= 65535)", B, Cout);
    const float rh = scale_ac(H, OH), rw = scale_ac(W, OW);
    if (rh > 0.f && rw > 0.f && (int)ceilf((TL + 1) / rh) + 7 <= UB_P && (int)ceilf((TL + 1) / rw) + 7 <= UB_P) {
        dim3 grid3(gdm_cdiv(H, TL) * gdm_cdiv(W, TL), (unsigned)(B * Cout));
        hipLaunchKernelGGL(upconv3x3_gather_bwd_lds_kernel, grid3, dim3(256), 0, (hipStream_t)stream, grad_out, Cout, H, W, OH, OW, rh, rw, grad_z);
        return gdm_launch_status("upconv3x3_gather_bwd_lds_kernel");
    }
    dim3 grid(gdm_cdiv(H, TL) * gdm_cdiv(W, TL), (unsigned)(B * 9 * Cout));
    hipLaunchKernelGGL(upconv3x3_gather_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, Cout, H, W, OH, OW,
                       scale_ac(H, OH), scale_ac(W, OW), grad_z);
    return gdm_launch_status("upconv3x3_gather_bwd_kernel");
}

extern "C" int gdm_psp_combine2_hip(const float* g, const float* y1, int s1, const float* y2, int s2, const float* y3, int s3,
                                    const float* y4, int s4, const float* bias, int B, int C, int H, int W, float* out, void* outpk,
                                    void* stream);

extern "C" int gdm_psp_combine_hip(const float* g, const float* y1, int s1, const float* y2, int s2, const float* y3, int s3,
                                   const float* y4, int s4, const float* bias, int B, int C, int H, int W, float* out, void* stream)
{
    return gdm_psp_combine2_hip(g, y1, s1, y2, s2, y3, s3, y4, s4, bias, B, C, H, W, out, nullptr, stream);
}

extern "C" int gdm_psp_combine2_hip(const float* g, const float* y1, int s1, const float* y2, int s2, const float* y3, int s3,
                                    const float* y4, int s4, const float* bias, int B, int C, int H, int W, float* out, void* outpk,
                                    void* stream)
{
    GDM_CHECK_ARG(g && y1 && y2 && y3 && y4 && out, "gdm_psp_combine_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && C >= 1 && (long)B * C <= 65535 && H >= 1 && W >= 1 && s1 >= 1 && s2 >= 1 && s3 >= 1 && s4 >= 1,
                  "gdm_psp_combine_hip: bad shape");
    GDM_CHECK_ARG(s1 <= 8 && s2 <= 8 && s3 <= 8 && s4 <= 8, "gdm_psp_combine_hip: prior maps up to 8x8 (the reference uses 1, 2, 3, 6)");
    PspMaps maps;
    maps.y[0] = y1; maps.y[1] = y2; maps.y[2] = y3; maps.y[3] = y4;
    maps.s[0] = s1; maps.s[1] = s2; maps.s[2] = s3; maps.s[3] = s4;
    for (int k = 0; k < 4; ++k) {
        maps.sy[k] = H > 1 ? (float)(maps.s[k] - 1) / (float)(H - 1) : 0.f;
        maps.sx[k] = W > 1 ? (float)(maps.s[k] - 1) / (float)(W - 1) : 0.f;
    }
    GDM_CHECK_ARG(((long)H * W) % 4 == 0 && (((uintptr_t)g | (uintptr_t)out) & 15) == 0, "gdm_psp_combine_hip: H*W must be a multiple of 4 and the maps 16-byte aligned");
    int gx = gdm_cdiv((long)H * W, 1024);
    if (gx > 16) gx = 16;
    if (outpk) {
        GDM_CHECK_ARG(C % 8 == 0 && W % 4 == 0 && ((uintptr_t)outpk & 15) == 0 && (C == 64 || C % 128 == 0),
                      "gdm_psp_combine2_hip: packed output needs C = 64 or a multiple of 128 and W %% 4 == 0 (C=%d W=%d)", C, W);
        hipLaunchKernelGGL(psp_combine8_kernel, dim3(gx, B * C / 8), dim3(256), 0, (hipStream_t)stream, g, maps, bias, C, H, W, out,
                           (unsigned char*)outpk);
        return gdm_launch_status("psp_combine8_kernel");
    }
    hipLaunchKernelGGL(psp_combine_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, g, maps, bias, C, H, W, out);
    return gdm_launch_status("psp_combine_kernel");
}

extern "C" int gdm_gather_add_affine_act_hip(const float* x, const float* t, const int32_t* idx, const float* scale, const float* shift,
                                             int B, int C, int n, int m, int act, float slope, float* y, void* stream)
{
    return gdm_gather_add_affine_act2_hip(x, t, idx, scale, shift, B, C, n, m, act, slope, y, nullptr, 0, stream);
}

extern "C" int gdm_gather_add_affine_act2_hip(const float* x, const float* t, const int32_t* idx, const float* scale, const float* shift,
                                              int B, int C, int n, int m, int act, float slope, float* y, void* y_packed, int W, void* stream)
{
    unsigned char* ypk = (unsigned char*)y_packed;
    GDM_CHECK_ARG(x && t && idx && scale && shift && y, "gdm_gather_add_affine_act_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && C >= 1 && n >= 1 && m >= 1 && act >= 0 && act <= 2, "gdm_gather_add_affine_act_hip: bad shape");
    GDM_CHECK_ARG(!ypk || (W >= 1 && m % W == 0 && C % 8 == 0 && (C == 64 || C % 128 == 0) && ((uintptr_t)ypk & 15) == 0),
                  "gdm_gather_add_affine_act2_hip: packed output needs m %% W == 0, C = 64 or a multiple of 128, a 16-byte aligned buffer");
    dim3 grid(gdm_cdiv(m, 256), gdm_cdiv(C, 8), B);
    hipStream_t s = (hipStream_t)stream;
    if (act == 0) hipLaunchKernelGGL(gather_add_affine_act_kernel<0>, grid, dim3(256), 0, s, x, t, idx, scale, shift, C, n, m, slope, y, ypk, W);
    else if (act == 1) hipLaunchKernelGGL(gather_add_affine_act_kernel<1>, grid, dim3(256), 0, s, x, t, idx, scale, shift, C, n, m, slope, y, ypk, W);
    else hipLaunchKernelGGL(gather_add_affine_act_kernel<2>, grid, dim3(256), 0, s, x, t, idx, scale, shift, C, n, m, slope, y, ypk, W);
    return gdm_launch_status("gather_add_affine_act_kernel");
}

extern "C" int gdm_conv1x1_gather_add_act2_hip(const float* x, const float* wt, const float* t, const int32_t* idx, const float* scale,
                                               const float* shift, int B, int C, int n, long m, int act, float slope, int pixel_major,
                                               float* y, void* stream)
{
    GDM_CHECK_ARG(x && wt && t && idx && scale && shift && y, "gdm_conv1x1_gather_add_act_hip: NULL pointer");
    GDM_CHECK_ARG(C == 64, "gdm_conv1x1_gather_add_act_hip: C=%d, only the 64-channel fusion levels are built", C);
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1 && m >= 1 && act >= 0 && act <= 2, "gdm_conv1x1_gather_add_act_hip: bad shape");
    dim3 grid(gdm_cdiv(m, 256), B);
    hipStream_t s = (hipStream_t)stream;
#define CGA(A, P) hipLaunchKernelGGL((conv1x1_gather_add_act_kernel<64, A, P>), grid, dim3(256), 0, s, x, wt, t, idx, scale, shift, n, m, slope, y)
    if (pixel_major) { if (act == 0) CGA(0, true); else if (act == 1) CGA(1, true); else CGA(2, true); }
    else { if (act == 0) CGA(0, false); else if (act == 1) CGA(1, false); else CGA(2, false); }
#undef CGA
    return gdm_launch_status("conv1x1_gather_add_act_kernel");
}

extern "C" int gdm_conv1x1_gather_add_act_hip(const float* x, const float* wt, const float* t, const int32_t* idx, const float* scale,
                                              const float* shift, int B, int C, int n, long m, int act, float slope, float* y, void* stream)
{
    return gdm_conv1x1_gather_add_act2_hip(x, wt, t, idx, scale, shift, B, C, n, m, act, slope, 0, y, stream);
}

// ResNet stem tail: BatchNorm (folded) + ReLU + MaxPool2d(3, stride 2, padding 1) in one pass over the 7x7 convolution's output
// (extractors.py:128-131): the two-kernel form writes and re-reads the 2x-resolution activated map.  Thread = output pixel.
__global__ __launch_bounds__(256) void affine_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int C, int H, int W, int OH, int OW,
                                                                  float* __restrict__ y)
{
    const int plane = blockIdx.y;                                 // b * C + c
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= OH * OW) return;
    const int oy = o / OW, ox = o - oy * OW;
    const int c = plane % C;
    const float sc = scale[c], sh = shift[c];
    const float* xp = x + (long)plane * H * W;
    float best = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = 2 * oy - 1 + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = 2 * ox - 1 + dx;
            if (xx < 0 || xx >= W) continue;
            best = fmaxf(best, fmaxf(fmaf(xp[(long)yy * W + xx], sc, sh), 0.f));
        }
    }
    y[(long)plane * OH * OW + o] = best;
}

extern "C" int gdm_affine_relu_maxpool_hip(const float* x, const float* scale, const float* shift, int B, int C, int H, int W, float* y, void* stream)
{
    GDM_CHECK_ARG(x && scale && shift && y, "gdm_affine_relu_maxpool_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && C >= 1 && H >= 1 && W >= 1 && (long)B * C <= 65535, "gdm_affine_relu_maxpool_hip: bad shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;          // floor((H + 2 - 3) / 2) + 1
    hipLaunchKernelGGL(affine_relu_maxpool_kernel, dim3(gdm_cdiv((long)OH * OW, 256), B * C), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       C, H, W, OH, OW, y);
    return gdm_launch_status("affine_relu_maxpool_kernel");
}

extern "C" int gdm_conv1x1_logsoftmax_hip(const float* x, const float* w, const float* bias, int B, int C, long hw, float* out, void* stream)
{
    GDM_CHECK_ARG(x && w && out, "gdm_conv1x1_logsoftmax_hip: NULL pointer");
    GDM_CHECK_ARG(C == 64, "gdm_conv1x1_logsoftmax_hip: C=%d, only the 64-channel `final` stage is built", C);
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && hw >= 1, "gdm_conv1x1_logsoftmax_hip: bad shape");
    hipLaunchKernelGGL(conv1x1_logsoftmax_kernel<64>, dim3(gdm_cdiv(hw, 256), B), dim3(256), 0, (hipStream_t)stream, x, w, bias, hw, out);
    return gdm_launch_status("conv1x1_logsoftmax_kernel");
}

extern "C" int gdm_psp_pools_hip(const float* x, long planes, int H, int W, float* o1, float* o2, float* o3, float* o6, void* stream)
{
    GDM_CHECK_ARG(x && o1 && o2 && o3 && o6, "gdm_psp_pools_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && H >= 6 && W >= 6 && H * W <= 64 * 64, "gdm_psp_pools_hip: map %dx%d must be between 6x6 and 64x64 pixels", H, W);
    hipLaunchKernelGGL(psp_pools_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, H, W, o1, o2, o3, o6);
    return gdm_launch_status("psp_pools_kernel");
}

extern "C" int gdm_psp_pools_bwd_hip(const float* g1, const float* g2, const float* g3, const float* g6, long planes, int H, int W,
                                     float* grad_x, void* stream)
{
    GDM_CHECK_ARG(g1 && g2 && g3 && g6 && grad_x, "gdm_psp_pools_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(planes >= 1 && H >= 6 && W >= 6 && H * W <= 64 * 64, "gdm_psp_pools_bwd_hip: map %dx%d must be between 6x6 and 64x64 pixels", H, W);
    hipLaunchKernelGGL(psp_pools_bwd_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, g1, g2, g3, g6, H, W, grad_x);
    return gdm_launch_status("psp_pools_bwd_kernel");
}

extern "C" int gdm_depth_to_xyz_hip(const float* depth, const float* K, const int32_t* origin, int B, int H, int W, int S,
                                    float* out, void* stream)
{
    GDM_CHECK_ARG(depth && K && origin && out, "gdm_depth_to_xyz_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && S >= 1, "gdm_depth_to_xyz_hip: bad shape");
    hipLaunchKernelGGL(depth_to_xyz_kernel, dim3(gdm_cdiv((long)S * S, 256), B), dim3(256), 0, (hipStream_t)stream, depth, K, origin, H, W, S, out);
    return gdm_launch_status("depth_to_xyz_kernel");
}
