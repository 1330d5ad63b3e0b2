// Weight gradient of the trunk's 3x3 / stride 1 / pad 1 convolutions on the matrix cores (training), gfx950.
//
//   dW[co, ci, ky, kx] = sum_{b, y, x} go[b, co, y, x] * in[b, ci, y + ky - 1, x + kx - 1]
// (backward of /root/reference/models/cnn/extractors.py:36-58 under /root/reference/train_lm.py:285 `loss.backward()`; torch sends it
// to MIOpen's fp32 implicit-GEMM wgrad kernels: 1.02 ms for 512 -> 512 at 32 x 32, batch 24 = 113 TFLOP/s, after NHWC transposes).
// It is a GEMM whose contraction runs over the PIXELS of all images: dW_tap = GO (Cout x P) . X_tap^T (P x Cin).  The two kernels here
// re-lay the operands so that the split-bf16 MFMA GEMM of gdm_conv.hip (conv_mfma16_kernel, 1x1 form) computes it unchanged:
//   * its "input channels" (K) are the pixels kappa = b*H*W + p, in chunks of 128;
//   * its "pixels" (n) are (tap, ci): a 9 x Cin grid, so one launch covers the nine taps;
//   * its "weights" are the rows of GO: row (chunk, co) = bf16 hi | lo of go[b, co, 128 consecutive pixels].
// A tap shift cannot be a pointer offset here (the pixels sit INSIDE the 8-element operand granules), so the X operand is written
// nine times, once per tap, with the shift applied while the fp32 values pass through LDS (zero outside the map = the padding).
// K is split over a few launches (chunk ranges = pointer offsets); the partial [Cout, 9, Cin] products are added by the caller.
#include "gdm_common.h"

namespace {

// x f32[B,Cin,H,W] -> packed "activations" of the GEMM: K chunk c (global: b * (HW/128) + cl), plane q < 16 = hi of K rows
// [8q, 8q+8) of the chunk, plane 16 + q = lo; element (tap + 1, ci + 1) of the (9 + 2) x (Cin + 2) grid, 16 B.
// One block: one chunk x 64 input channels.
template <int W>
__global__ __launch_bounds__(256) void wgrad_pack_x_kernel(const float* __restrict__ x, int Cin, int H, unsigned char* __restrict__ out)
{
    constexpr int R = 128 / W;                    // image rows per chunk
    constexpr int TS = (R + 2) * W + 1;           // floats per channel in LDS (+1: lanes = channels read conflict-free)
    __shared__ float t[64 * TS];
    const int tid = threadIdx.x;
    const int hw = H * W, cpi = hw / 128;
    const int c = blockIdx.x, b = c / cpi, cl = c - b * cpi;
    const int y0 = (cl * 128) / W;
    const int ci0 = blockIdx.y * 64;
    for (int e = tid; e < 64 * (R + 2) * W; e += 256) {
        const int xx = e % W, r = (e / W) % (R + 2), ci = e / ((R + 2) * W);
        const int y = y0 - 1 + r;
        float v = 0.f;
        if (y >= 0 && y < H && ci0 + ci < Cin) v = x[(((long)b * Cin + ci0 + ci) * H + y) * W + xx];
        t[ci * TS + r * W + xx] = v;
    }
    __syncthreads();
    const long plane = (long)11 * (Cin + 2);
    for (int item = tid; item < 9 * 16 * 64; item += 256) {
        const int ci = item & 63, q = (item >> 6) & 15, tap = item >> 10;
        if (ci0 + ci >= Cin) continue;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pg = 8 * q, rr = pg / W, x0 = pg - rr * W;
        const float* row = t + ci * TS + (rr + ky) * W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xs = x0 + kx - 1 + j;
            v[j] = (xs >= 0 && xs < W) ? row[xs] : 0.f;
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
        unsigned char* o = out + ((((long)c * 32 + q) * plane) + (long)(tap + 1) * (Cin + 2) + ci0 + ci + 1) * 16;
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 16 * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// go f32[B,Cout,H,W] -> packed "weights" of the GEMM: row (chunk * CoutP + co) = 512 B: bf16 hi of go[b, co, 128 cl .. + 128] | lo;
// rows co >= Cout are zero.  Thread = (row, 8-pixel group).
__global__ __launch_bounds__(256) void wgrad_pack_go_kernel(const float* __restrict__ go, int Cout, int CoutP, int hw, long rows,
                                                            unsigned char* __restrict__ out)
{
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= rows * 16) return;
    const int g = (int)(item & 15);
    const long row = item >> 4;
    const int co = (int)(row % CoutP);
    const long c = row / CoutP;
    const int cpi = hw / 128;
    const long b = c / cpi;
    const int cl = (int)(c - b * cpi);
    float v[8];
    if (co < Cout) {
        const float4* src = reinterpret_cast<const float4*>(go + (b * Cout + co) * hw + cl * 128 + 8 * g);
        const float4 a = src[0], d = src[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
    unsigned char* r = out + row * 512;
    *reinterpret_cast<uint4*>(r + g * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(r + 256 + g * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// The 1x1 case (a plain product over pixels, dW[co,ci] = sum_{b,p} go[b,co,p] * x[b,ci,p]): no tap, no halo.  x f32[B,Cin,P], P % 128 == 0
// -> the GEMM's "activations" on a 1 x Cin grid: chunk c, plane q (hi) / 16 + q (lo), element (1, ci + 1) of a 3 x (Cin + 2) grid.
// One block: one chunk x 64 input channels; lanes = channels on the store side, = pixels on the load side (through LDS).
__global__ __launch_bounds__(256) void wgrad_pack_x1_kernel(const float* __restrict__ x, int Cin, int P, unsigned char* __restrict__ out)
{
    __shared__ float t[64][129];
    const int tid = threadIdx.x;
    const int cpi = P / 128;
    const int c = blockIdx.x, b = c / cpi, cl = c - b * cpi;
    const int ci0 = blockIdx.y * 64;
    for (int e = tid; e < 64 * 128; e += 256) {
        const int p = e & 127, ci = e >> 7;
        t[ci][p] = (ci0 + ci < Cin) ? x[((long)b * Cin + ci0 + ci) * P + cl * 128 + p] : 0.f;
    }
    __syncthreads();
    const long plane = (long)3 * (Cin + 2);
    for (int item = tid; item < 16 * 64; item += 256) {
        const int ci = item & 63, q = item >> 6;
        if (ci0 + ci >= Cin) continue;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gdm_split2(t[ci][8 * q + 2 * j], t[ci][8 * q + 2 * j + 1], hi[j], lo[j]);
        unsigned char* o = out + ((((long)c * 32 + q) * plane) + (Cin + 2) + ci0 + ci + 1) * 16;
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 16 * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// Small-channel products (Cout, Cin <= a few hundred, millions of pixels: the 1x1 layers of the 64 / 32-channel full-resolution stages and
// of the point branch): HBM-bound, so no re-layout pass -- a wave reads its MFMA fragments straight from the fp32 rows (lane = row l & 15,
// 8 consecutive pixels 8 (l >> 4) .. of a 32-pixel k-step = two 16-byte loads), splits them to bf16 hi / lo in registers and accumulates
// TM x TN tiles of 16 x 16.  A workgroup = 4 waves = 4 interleaved pixel slices of one (K split, tile block); the waves' tiles are
// added through LDS in a fixed order and written as one partial product [split][Cout][Cin] (the caller adds the splits).
// bias (optional) = row sums of go, from the same loads: partial [split][Cout], written by the workgroups of tile column 0.
typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned wg_u32x4;

// TM x TN = 16 x 16 tiles per wave (<= 2 x 2: 110 registers, four waves per SIMD -- the kernel lives on bytes in flight), WM x WN = waves
// side by side on the workgroup's (16 TM WM) x (16 TN WN) block of dW, the remaining 4 / (WM WN) waves = interleaved pixel slices.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const float* __restrict__ go, long go_bs, const float* __restrict__ x, long x_bs,
                                                           int Cout, int Cin, int P, long nsteps, int nsplit, float* __restrict__ part,
                                                           float* __restrict__ bias_part)
{
    constexpr int KW = 4 / (WM * WN);                               // pixel slices inside the workgroup
    __shared__ float red[KW > 1 ? (KW - 1) * WM * WN : 1][TM * TN * 256];
    __shared__ float bred[KW > 1 ? (KW - 1) * WM * WN : 1][TM * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int wpos = wave % (WM * WN), kw = wave / (WM * WN);       // position in the block, pixel slice
    const int wm = wpos / WN, wn = wpos - wm * WN;
    const int split = blockIdx.x;
    const int nbn = (Cin + 16 * TN * WN - 1) / (16 * TN * WN);
    const int bm = blockIdx.y / nbn, bn = blockIdx.y - bm * nbn;
    const int tm0 = (bm * WM + wm) * TM, tn0 = (bn * WN + wn) * TN;  // first tile row / column of this wave
    const int spp = P / 32;                                          // k-steps per image
    // this wave's k-steps: the split's range [s0, s1), slice kw takes s0 + kw, s0 + kw + KW, ...
    const long s0 = nsteps * split / nsplit, s1 = nsteps * (split + 1) / nsplit;
    int rowa[TM], rowb[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) rowa[m] = min((tm0 + m) * 16 + l16, Cout - 1);
#pragma unroll
    for (int n = 0; n < TN; ++n) rowb[n] = min((tn0 + n) * 16 + l16, Cin - 1);
    wg_f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.f;
    float bsum[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) bsum[m] = 0.f;

    struct Frags { float4 a[TM][2], b[TN][2]; };
    Frags f0, f1, f2;                                               // named buffers (an index would send them to scratch): two steps in flight
    auto load = [&](Frags& f, long ks) {
        const long b = ks / spp;
        const int p0 = (int)(ks - b * spp) * 32 + 8 * kg;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const float4* r = reinterpret_cast<const float4*>(go + b * go_bs + (long)rowa[m] * P + p0);
            f.a[m][0] = r[0]; f.a[m][1] = r[1];
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const float4* r = reinterpret_cast<const float4*>(x + b * x_bs + (long)rowb[n] * P + p0);
            f.b[n][0] = r[0]; f.b[n][1] = r[1];
        }
    };
    auto split8 = [](const float4& a, const float4& b, wg_u32x4& hi, wg_u32x4& lo) {
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        gdm_split2(a.x, a.y, h0, l0); gdm_split2(a.z, a.w, h1, l1); gdm_split2(b.x, b.y, h2, l2); gdm_split2(b.z, b.w, h3, l3);
        hi = wg_u32x4{h0, h1, h2, h3}; lo = wg_u32x4{l0, l1, l2, l3};
    };
    auto compute = [&](const Frags& f) {
        wg_u32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            split8(f.a[m][0], f.a[m][1], ah[m], al[m]);
            const float4 u = f.a[m][0], v = f.a[m][1];
            bsum[m] += ((u.x + u.y) + (u.z + u.w)) + ((v.x + v.y) + (v.z + v.w));
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) split8(f.b[n][0], f.b[n][1], bh[n], bl[n]);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                wg_f32x4 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, ah[m]), __builtin_bit_cast(wg_bf16x8, bl[n]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, al[m]), __builtin_bit_cast(wg_bf16x8, bh[n]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, ah[m]), __builtin_bit_cast(wg_bf16x8, bh[n]), c, 0, 0, 0);
                acc[m][n] = c;
            }
    };
    long ks = s0 + kw;
    if (ks < s1) load(f0, ks);
    if (ks + KW < s1) load(f1, ks + KW);
    for (; ks < s1; ks += 3 * KW) {
        if (ks + 2 * KW < s1) load(f2, ks + 2 * KW);
        compute(f0);
        if (ks + KW >= s1) break;
        if (ks + 3 * KW < s1) load(f0, ks + 3 * KW);
        compute(f1);
        if (ks + 2 * KW >= s1) break;
        if (ks + 4 * KW < s1) load(f1, ks + 4 * KW);
        compute(f2);
    }
    // row sums: lanes l16 + 16 kg hold the four pixel groups of row l16 -> every lane
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        bsum[m] += __shfl_xor(bsum[m], 16);
        bsum[m] += __shfl_xor(bsum[m], 32);
    }
    // slices 1.. -> LDS, slice 0 adds them in order and stores; accumulator (m, n): lane -> column 16 n + l16 (ci), rows 16 m + 4 kg + r (co)
    if (KW > 1) {
        if (kw > 0) {
            const int slot = (kw - 1) * WM * WN + wpos;
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[slot][((m * TN + n) * 4 + r) * 64 + lane] = acc[m][n][r];
            if (kg == 0) {
#pragma unroll
                for (int m = 0; m < TM; ++m) bred[slot][m * 16 + l16] = bsum[m];
            }
        }
        __syncthreads();
        if (kw != 0) return;
    }
    float* o = part + (long)split * Cout * Cin;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int ci = (tn0 + n) * 16 + l16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (tm0 + m) * 16 + 4 * kg + r;
                float v = acc[m][n][r];
                if (KW > 1) {
#pragma unroll
                    for (int k = 1; k < KW; ++k) v += red[(k - 1) * WM * WN + wpos][((m * TN + n) * 4 + r) * 64 + lane];
                }
                if (co < Cout && ci < Cin) o[(long)co * Cin + ci] = v;
            }
        }
    if (bias_part && bn == 0 && wn == 0 && kg == 0) {
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int co = (tm0 + m) * 16 + l16;
            float v = bsum[m];
            if (KW > 1) {
#pragma unroll
                for (int k = 1; k < KW; ++k) v += bred[(k - 1) * WM * WN + wpos][m * 16 + l16];
            }
            if (co < Cout) bias_part[(long)split * Cout + co] = v;
        }
    }
}

bool shape_ok(int B, int C, int H, int W)
{
    return B >= 1 && C >= 1 && H >= 1 && (W == 32 || W == 64) && (H * W) % 128 == 0;
}

} // namespace

extern "C" size_t gdm_wgrad_x_bytes(int B, int Cin, int H, int W)
{
    if (!shape_ok(B, Cin, H, W) || Cin % 32 != 0) return 0;
    return (size_t)B * (H * W / 128) * 32 * 11 * (Cin + 2) * 16;
}

extern "C" size_t gdm_wgrad_go_bytes(int B, int Cout, int H, int W)
{
    if (!shape_ok(B, Cout, H, W)) return 0;
    return (size_t)B * (H * W / 128) * ((Cout + 127) & ~127) * 512;
}

extern "C" int gdm_wgrad_pack_x_hip(const float* x, int B, int Cin, int H, int W, void* out, void* stream)
{
    GDM_CHECK_ARG(x && out, "gdm_wgrad_pack_x_hip: NULL pointer");
    GDM_CHECK_ARG(shape_ok(B, Cin, H, W) && Cin % 32 == 0, "gdm_wgrad_pack_x_hip: unsupported shape B=%d Cin=%d H=%d W=%d (W in {32, 64}, H*W %% 128 == 0, "
                  "Cin %% 16 == 0)", B, Cin, H, W);
    GDM_CHECK_ARG(((uintptr_t)out & 15) == 0, "gdm_wgrad_pack_x_hip: out must be 16-byte aligned");
    dim3 grid(B * (H * W / 128), gdm_cdiv(Cin, 64));
    if (W == 32) hipLaunchKernelGGL(wgrad_pack_x_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, Cin, H, (unsigned char*)out);
    else hipLaunchKernelGGL(wgrad_pack_x_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, Cin, H, (unsigned char*)out);
    return gdm_launch_status("wgrad_pack_x_kernel");
}

extern "C" int gdm_wgrad_pack_go_hip(const float* go, int B, int Cout, int H, int W, void* out, void* stream)
{
    GDM_CHECK_ARG(go && out, "gdm_wgrad_pack_go_hip: NULL pointer");
    GDM_CHECK_ARG(shape_ok(B, Cout, H, W), "gdm_wgrad_pack_go_hip: unsupported shape B=%d Cout=%d H=%d W=%d", B, Cout, H, W);
    GDM_CHECK_ARG(((uintptr_t)out & 15) == 0 && ((uintptr_t)go & 15) == 0, "gdm_wgrad_pack_go_hip: buffers must be 16-byte aligned");
    const int CoutP = (Cout + 127) & ~127;
    const long rows = (long)B * (H * W / 128) * CoutP;
    hipLaunchKernelGGL(wgrad_pack_go_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, go, Cout, CoutP,
                       H * W, rows, (unsigned char*)out);
    return gdm_launch_status("wgrad_pack_go_kernel");
}

extern "C" size_t gdm_wgrad_x1_bytes(int B, int Cin, int P)
{
    if (B < 1 || Cin < 1 || P < 128 || P % 128 != 0 || Cin % 32 != 0) return 0;
    return (size_t)B * (P / 128) * 32 * 3 * (Cin + 2) * 16;
}

extern "C" int gdm_wgrad_pack_x1_hip(const float* x, int B, int Cin, int P, void* out, void* stream)
{
    GDM_CHECK_ARG(x && out, "gdm_wgrad_pack_x1_hip: NULL pointer");
    GDM_CHECK_ARG(gdm_wgrad_x1_bytes(B, Cin, P) != 0, "gdm_wgrad_pack_x1_hip: unsupported shape B=%d Cin=%d P=%d (P %% 128 == 0, Cin %% 32 == 0)", B, Cin, P);
    GDM_CHECK_ARG(((uintptr_t)out & 15) == 0, "gdm_wgrad_pack_x1_hip: out must be 16-byte aligned");
    dim3 grid(B * (P / 128), gdm_cdiv(Cin, 64));
    hipLaunchKernelGGL(wgrad_pack_x1_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, Cin, P, (unsigned char*)out);
    return gdm_launch_status("wgrad_pack_x1_kernel");
}

extern "C" int gdm_wgrad_direct_hip(const float* go, long go_bstride, const float* x, long x_bstride, int B, int Cout, int Cin, int P,
                                    int nsplit, float* partial, float* bias_partial, void* stream)
{
    GDM_CHECK_ARG(go && x && partial, "gdm_wgrad_direct_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && Cout >= 1 && Cin >= 1 && P >= 32 && P % 32 == 0 && nsplit >= 1 && nsplit <= 65535,
                  "gdm_wgrad_direct_hip: unsupported shape B=%d Cout=%d Cin=%d P=%d nsplit=%d (P %% 32 == 0)", B, Cout, Cin, P, nsplit);
    GDM_CHECK_ARG(((uintptr_t)go & 15) == 0 && ((uintptr_t)x & 15) == 0 && go_bstride % 4 == 0 && x_bstride % 4 == 0,
                  "gdm_wgrad_direct_hip: rows must be 16-byte aligned");
    const long nsteps = (long)B * (P / 32);
    GDM_CHECK_ARG(nsplit <= nsteps, "gdm_wgrad_direct_hip: more splits (%d) than 32-pixel steps (%ld)", nsplit, nsteps);
    // tiles per wave (<= 2 x 2), waves side by side (<= 2 x 2); see gdm_wgrad_direct_block()
    const int tm = Cout > 16 ? 2 : 1, tn = Cin > 16 ? 2 : 1;
    const int wm = Cout > 32 ? 2 : 1, wn = Cin > 32 ? 2 : 1;
    const int nbm = gdm_cdiv(Cout, 16 * tm * wm), nbn = gdm_cdiv(Cin, 16 * tn * wn);
    GDM_CHECK_ARG((long)nbm * nbn <= 65535, "gdm_wgrad_direct_hip: too many tile blocks");
    dim3 grid(nsplit, nbm * nbn);
#define GDM_WD(TM, TN, WM, WN) hipLaunchKernelGGL((wgrad_direct_kernel<TM, TN, WM, WN>), grid, dim3(256), 0, (hipStream_t)stream, go, go_bstride, x, \
                                                  x_bstride, Cout, Cin, P, nsteps, nsplit, partial, bias_partial)
    if (wm == 2 && wn == 2) GDM_WD(2, 2, 2, 2);
    else if (wm == 2 && tn == 2) GDM_WD(2, 2, 2, 1);
    else if (wm == 2) GDM_WD(2, 1, 2, 1);
    else if (wn == 2 && tm == 2) GDM_WD(2, 2, 1, 2);
    else if (wn == 2) GDM_WD(1, 2, 1, 2);
    else if (tm == 2 && tn == 2) GDM_WD(2, 2, 1, 1);
    else if (tm == 2) GDM_WD(2, 1, 1, 1);
    else if (tn == 2) GDM_WD(1, 2, 1, 1);
    else GDM_WD(1, 1, 1, 1);
#undef GDM_WD
    return gdm_launch_status("wgrad_direct_kernel");
}
