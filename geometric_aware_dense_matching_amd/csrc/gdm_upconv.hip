// PSPUpsample(64 -> 64) = Upsample(x2, bilinear, align_corners) + Conv3x3 + BN + PReLU (models/cnn/pspnet.py:34-45) in ONE kernel, gfx950:
// the dense form (`upconv_tile64_kernel`: the upsampled tile built in LDS, nine tap products on the matrix cores), the sampled-pixel form
// of the last image stage (`upconv_final_points_kernel`) and the 64-channel fusion tail (`conv64_gather_add_act_mfma_kernel`).  All use
// split-bf16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate: the scheme of the matching / convolution kernels).
#include "gdm_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int UF_C = 64;                 // channels in and out
constexpr int UF_ROWB = 256;             // packed row: 64 bf16 hi | 64 bf16 lo

__device__ __forceinline__ unsigned short bf16_rne(float v) { return gdm_bf16_1(v); }
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// 16-byte chunk `ch` (0-7 hi, 8-15 lo) of packed row r, XOR-swizzled so that 16 rows' same chunk hit 16 different bank groups
__device__ __forceinline__ int uf_off(int r, int ch) { return r * UF_ROWB + ((ch ^ (r & 15)) << 4); }

// w f32[64,64,3,3] -> rows (tap*64 + co) of 256 B: 64 bf16 hi | 64 bf16 lo over the input channels
__global__ __launch_bounds__(256) void upconv_pack_w_kernel(const float* __restrict__ w, unsigned char* __restrict__ out)
{
    const int item = blockIdx.x * 256 + threadIdx.x;           // (row, input channel)
    if (item >= 9 * UF_C * UF_C) return;
    const int ci = item & 63, row = item >> 6;
    const int co = row & 63, tap = row >> 6;
    const float v = w[((long)co * UF_C + ci) * 9 + tap];
    const unsigned short hi = bf16_rne(v), lo = bf16_rne(v - bf16_f32(hi));
    unsigned short* r = reinterpret_cast<unsigned short*>(out + (long)row * UF_ROWB);
    r[ci] = hi;
    r[64 + ci] = lo;
}

// (the z-gather form of rounds 1-2, `upconv_fused64_kernel`, and its GDM_UPCONV_FUSED64=gather switch were removed in round 4: the
// direct form below replaced it and no test or caller reached it)

// ---------------------------------------------------------------------------------------------------------------------------------
// Direct form: the workgroup builds the UPSAMPLED tile (8 x 16 output pixels + the one-pixel ring the 3x3 taps reach, 64 channels,
// split bf16 hi / lo) in LDS and runs the 3x3 convolution on it: nine taps x K = 64 on the matrix cores with the tap's weights
// staged through LDS (double-buffered), accumulators in registers, one store per output.  Against the z-gather form above: twice
// the MFMA work (the product runs at output resolution), but no [9*64, patch] intermediate in LDS and no 36-read gather per
// output value, which is what bounded that kernel (LDS reads: 2.4 MB per 256 outputs x 64 channels).  The upsampled value is
// formed exactly as torch's upsample_bilinear2d (align_corners) forms it, then split -- so the result is conv3x3(up(x)) in the
// reference's own order of operations (models/cnn/pspnet.py:34-45), to split-bf16 accuracy.
constexpr int UT_TX = 16, UT_TY = 8;                  // output tile
constexpr int UT_NT = 4;                              // tiles (stacked in y) per workgroup: the next tile's source patch is in flight
                                                      // during this tile's matrix work, and stores never wait for a workgroup's end
constexpr int UT_HW = UT_TX + 2;                      // tile row stride with the ring (18)
constexpr int UT_ROWS = (UT_TY + 2) * UT_HW;          // 180 packed rows
constexpr int UT_TILE = UT_ROWS * UF_ROWB;            // 46 080 B
constexpr int UT_WBUF = UF_C * UF_ROWB;               // one tap's weights: 16 KiB
constexpr int UT_PH = 8, UT_PWV = 11;                 // source patch rows / columns: checked on the host for the scale factors
constexpr int UT_PW = 16;                             // patch row stride (a thread's patch position is then shifts of its index)
constexpr int UT_PS = UT_PH * UT_PW;                  // floats per channel of the patch
constexpr int UT_NE = UF_C * UT_PS / 256;             // patch elements per thread (32: channels t/128 + 2 i at position t % 128)
constexpr int UT_LDS = UT_TILE + 2 * UT_WBUF + 2 * UF_C * 4;   // 79 360 B: two workgroups per CU
static_assert(UF_C * UT_PS * 4 <= 2 * UT_WBUF, "the fp32 patch aliases the weight buffers");
static_assert(UF_C * UT_PS % 256 == 0, "whole patch elements per thread");

template <int ACT>
__global__ __launch_bounds__(256, 2) void upconv_tile64_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wpk,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int H, int W, int OH, int OW, float rh, float rw, float slope,
                                                             float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    unsigned char* ws = smem + UT_TILE;
    float* patch = reinterpret_cast<float*>(ws);                 // between two tiles' tap loops only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int ox_t = blockIdx.x * UT_TX;
    const float* xb = x + (long)b * UF_C * H * W;
    const int xs0 = min((int)(rw * (float)max(ox_t - 1, 0)), W - 1);
    const int xs1 = min((int)(rw * (float)min(ox_t + UT_TX, OW - 1)) + 1, W - 1);
    const int pw = xs1 - xs0 + 1;                                  // <= UT_PW

    u32x4 stage[4];
    auto stage_load = [&](int tap) {
        const unsigned char* src = wpk + (long)tap * UT_WBUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[i] = *reinterpret_cast<const u32x4*>(src + (long)(i * 256 + tid) * 16);
    };
    auto stage_store = [&](int buf) {
        unsigned char* dst = ws + buf * UT_WBUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = i * 256 + tid;
            *reinterpret_cast<u32x4*>(dst + uf_off(g >> 4, g & 15)) = stage[i];
        }
    };
    // source window of a tile (ring and the +1 bilinear neighbour included): first row and row count
    auto window = [&](int oy_t, int& ys0, int& ph) {
        ys0 = min((int)(rh * (float)max(oy_t - 1, 0)), H - 1);
        const int ys1 = min((int)(rh * (float)min(oy_t + UT_TY, OH - 1)) + 1, H - 1);
        ph = ys1 - ys0 + 1;                                        // <= UT_PH
    };
    float pv[UT_NE];
    const int pc = tid & 15, pr = (tid >> 4) & 7;                  // this thread's patch position, channels (tid >> 7) + 2 i
    auto patch_load = [&](int oy_t) {
        int ys0, ph;
        window(oy_t, ys0, ph);
        const bool ok = pr < ph && pc < pw;
        const float* src = xb + ((long)(tid >> 7) * H + min(ys0 + pr, H - 1)) * W + min(xs0 + pc, W - 1);
#pragma unroll
        for (int i = 0; i < UT_NE; ++i) pv[i] = ok ? src[(long)(2 * i) * H * W] : 0.f;
    };

    const int prow = 2 * wave + (lr >> 4), pcol = lr & 15;         // the wave's 32 output pixels: two tile rows of 16
    float* ssc = reinterpret_cast<float*>(smem + UT_TILE + 2 * UT_WBUF);    // scale[64] | shift[64]
    if (tid < 2 * UF_C) ssc[tid] = tid < UF_C ? scale[tid] : shift[tid - UF_C];

    const int oy_first = blockIdx.y * (UT_NT * UT_TY);
    patch_load(oy_first);
    stage_load(0);
#pragma unroll 1
    for (int it = 0; it < UT_NT; ++it) {
        const int oy_t = oy_first + it * UT_TY;
        if (oy_t >= OH) break;                                     // uniform
        // ---- A: this tile's source patch registers -> LDS (fp32); then the next tile's loads go out ----
#pragma unroll
        for (int i = 0; i < UT_NE; ++i) patch[tid + 256 * i] = pv[i];
        __syncthreads();
        int ys0, ph;
        window(oy_t, ys0, ph);
        if (it + 1 < UT_NT && oy_t + UT_TY < OH) patch_load(oy_t + UT_TY);

        // ---- B: up(x) on the tile, 8 channels of one tile pixel per item, as torch's upsample_bilinear2d (align_corners) ----
#pragma unroll 1
        for (int item = tid; item < 8 * UT_ROWS; item += 256) {
            const int grp = item / UT_ROWS, px = item - grp * UT_ROWS;
            const int r = px / UT_HW, c = px - r * UT_HW;
            const int yy = oy_t - 1 + r, xx = ox_t - 1 + c;
            unsigned hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
            if (yy >= 0 && yy < OH && xx >= 0 && xx < OW) {        // outside: the convolution's zero padding
                const float sy = rh * (float)yy, sx = rw * (float)xx;
                const int y0 = (int)sy, x0 = (int)sx;
                const int yp = y0 < H - 1 ? UT_PW : 0, xp = x0 < W - 1 ? 1 : 0;
                const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const float* p0 = patch + (grp * 8) * UT_PS + (y0 - ys0) * UT_PW + (x0 - xs0);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* p = p0 + j * UT_PS;
                    v[j] = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[yp] + lx1 * p[yp + xp]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
            }
            *reinterpret_cast<u32x4*>(tile + uf_off(px, grp)) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(tile + uf_off(px, 8 + grp)) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
        __syncthreads();                                           // tile complete, patch dead
        stage_store(0);                                            // tap 0's weights (loaded before the loop / during the last tap)
        __syncthreads();

        // ---- C: nine taps x K = 64 for 32 pixels x 64 output channels per wave ----
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            stage_load(tap + 1 < 9 ? tap + 1 : 0);                 // the last tap fetches tap 0 for the next tile
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int R = (prow + ky) * UT_HW + pcol + kx;
            const unsigned char* wb = ws + (tap & 1) * UT_WBUF;
            u32x4 bh[4], bl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bh[s] = *reinterpret_cast<const u32x4*>(tile + uf_off(R, 2 * s + h));
                bl[s] = *reinterpret_cast<const u32x4*>(tile + uf_off(R, 8 + 2 * s + h));
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int n = cb * 32 + lr;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + uf_off(n, 2 * s + h)));
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + uf_off(n, 8 + 2 * s + h)));
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, bh[s]);
                    const bf16x8 xl = __builtin_bit_cast(bf16x8, bl[s]);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[cb], 0, 0, 0);
                }
            }
            if (tap + 1 < 9) stage_store((tap + 1) & 1);           // last read in iteration tap - 1, which ended with a barrier
            __syncthreads();
        }

        // ---- D: lane = output pixel, registers = output channels ----
        const int oy = oy_t + prow, ox = ox_t + pcol;
        if (oy < OH && ox < OW) {
            float* ob = out + (((long)b * UF_C) * OH + oy) * OW + ox;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    float o = acc[cb][reg] * ssc[co] + ssc[UF_C + co];
                    if (ACT == 1) o = fmaxf(o, 0.f);
                    if (ACT == 2) o = o > 0.f ? o : o * slope;
                    ob[(long)co * OH * OW] = o;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The last image stage at the SAMPLED pixels only (inference).  FFB6DEmb ends with up_3 = PSPUpsample(64 -> 64) to full resolution,
// `final` = Conv1x1(64 -> 64) + LogSoftmax, and then keeps the N `choose` pixels of each crop (ffb6d.py:266-285 of the reference:
// cnn_up_stages[3], torch.gather with choose_emb).  Both modules are per-pixel functions of the 3x3 upsampled neighbourhood, so in
// eval mode (BatchNorm on running statistics) the values at the chosen pixels are all that is ever read: 2 048 of 65 536 pixels per
// crop.  One workgroup takes 32 chosen pixels of one crop: for every tap it forms up(x) at the tap position exactly as
// upsample_bilinear2d (align_corners) does -- from the pixel-major source map, 256 contiguous bytes per source pixel -- splits it to
// bf16 hi/lo in LDS, accumulates W_tap . up(x) on the matrix cores (weights straight from L2 into the A fragments), applies BN + PReLU,
// then runs the 64 -> 64 `final` convolution the same way on the result and finishes the log-softmax over the 64 channels
// (two lane shuffles + one exchange through LDS).  Same products and summation structure as the dense kernels, 1/32 of the work.
constexpr int FP_P = 32;                              // chosen pixels per workgroup
constexpr int FP_ABUF = FP_P * UF_ROWB;               // one tap's operand rows: 8 KiB
typedef __attribute__((ext_vector_type(4))) float f32x4v;

__device__ __forceinline__ int fp_chunk(int row, int ch) { return uf_off(row, ch); }

template <int ACT>
__global__ __launch_bounds__(256, 4) void upconv_final_points_kernel(const float* __restrict__ xpm, const int32_t* __restrict__ choose,
                                                                   const unsigned char* __restrict__ wpk, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const unsigned char* __restrict__ wfpk,
                                                                   const float* __restrict__ fbias, int H, int W, int OH, int OW, int N,
                                                                   float rh, float rw, float slope, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char abuf[2 * FP_ABUF];
    __shared__ float red[2][4][FP_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y, n0 = blockIdx.x * FP_P;

    // ---- builder role: thread = (chosen pixel tid >> 3, 8-channel group tid & 7) ----
    const int bp = tid >> 3, grp = tid & 7;
    int oy, ox;
    {
        int ch = choose[(long)b * N + min(n0 + bp, N - 1)];
        ch = min(max(ch, 0), OH * OW - 1);
        oy = ch / OW;
        ox = ch - oy * OW;
    }
    const float* xb = xpm + (long)b * H * W * UF_C + grp * 8;
    f32x4v raw[8];                                              // 4 corners x 8 channels of the tap being built
    float ly1, lx1;
    bool inside;
    auto corners_load = [&](int tap) {
        const int yy = oy + tap / 3 - 1, xx = ox + tap % 3 - 1;
        inside = yy >= 0 && yy < OH && xx >= 0 && xx < OW;       // outside: the convolution's zero padding
        const float sy = rh * (float)min(max(yy, 0), OH - 1), sx = rw * (float)min(max(xx, 0), OW - 1);
        const int y0 = (int)sy, x0 = (int)sx;
        const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
        ly1 = sy - (float)y0;
        lx1 = sx - (float)x0;
        const float* p00 = xb + ((long)y0 * W + x0) * UF_C;
        const float* p01 = p00 + xp * UF_C;
        const float* p10 = p00 + (long)yp * W * UF_C;
        const float* p11 = p10 + xp * UF_C;
        raw[0] = *reinterpret_cast<const f32x4v*>(p00);
        raw[1] = *reinterpret_cast<const f32x4v*>(p00 + 4);
        raw[2] = *reinterpret_cast<const f32x4v*>(p01);
        raw[3] = *reinterpret_cast<const f32x4v*>(p01 + 4);
        raw[4] = *reinterpret_cast<const f32x4v*>(p10);
        raw[5] = *reinterpret_cast<const f32x4v*>(p10 + 4);
        raw[6] = *reinterpret_cast<const f32x4v*>(p11);
        raw[7] = *reinterpret_cast<const f32x4v*>(p11 + 4);
    };
    auto rows_store = [&](int buf) {                              // up(x) as upsample_bilinear2d forms it, split, one 16-B chunk hi + lo
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * j + e;
                const float a = raw[c >> 2][c & 3], bq = raw[2 + (c >> 2)][c & 3], cq = raw[4 + (c >> 2)][c & 3], d = raw[6 + (c >> 2)][c & 3];
                v[e] = inside ? ly0 * (lx0 * a + lx1 * bq) + ly1 * (lx0 * cq + lx1 * d) : 0.f;
            }
            gdm_split2(v[0], v[1], hi[j], lo[j]);
        }
        unsigned char* dst = abuf + buf * FP_ABUF;
        *reinterpret_cast<u32x4*>(dst + fp_chunk(bp, grp)) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(dst + fp_chunk(bp, 8 + grp)) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    };

    // ---- matrix role: wave = 16 output channels x the 32 pixels (two 16-column blocks), 16x16x32 products ----
    // A fragment (weights): lane (row l16 = channel 16 wave + l16, k-group kg): 8 input channels 32 S + 8 kg .. of k-step S
    u32x4 wh[2], wl[2];
    auto weights_load = [&](const unsigned char* rows) {           // rows: 64 packed rows of 256 B (64 bf16 hi | 64 bf16 lo)
        const unsigned char* r = rows + (long)(16 * wave + l16) * UF_ROWB;
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            wh[S] = *reinterpret_cast<const u32x4*>(r + (4 * S + kg) * 16);
            wl[S] = *reinterpret_cast<const u32x4*>(r + (8 + 4 * S + kg) * 16);
        }
    };
    f32x4v acc[2];
    auto mma = [&](int buf) {
        const unsigned char* src = abuf + buf * FP_ABUF;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + fp_chunk(16 * pb + l16, 4 * S + kg)));
                const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + fp_chunk(16 * pb + l16, 8 + 4 * S + kg)));
                const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[S]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, wl[S]);
                acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc[pb], 0, 0, 0);
                acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc[pb], 0, 0, 0);
                acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc[pb], 0, 0, 0);
            }
    };
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[pb] = f32x4v{0.f, 0.f, 0.f, 0.f};

    corners_load(0);
    rows_store(0);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        weights_load(wpk + (long)tap * UF_C * UF_ROWB);
        if (tap + 1 < 9) corners_load(tap + 1);                    // in flight during this tap's products
        __syncthreads();                                           // buffer tap & 1 is complete; the other one's readers are done
        mma(tap & 1);
        if (tap + 1 < 9) rows_store((tap + 1) & 1);
    }
    weights_load(wfpk);                                            // `final`'s weights, in flight during the epilogue
    __syncthreads();                                               // all products of tap 8 are done with buffer 0

    // ---- BN (+ conv bias) + PReLU; the result becomes the operand rows of `final` (buffer 0) ----
    // accumulator: lane column l16 = pixel 16 pb + l16, registers r = output channels 16 wave + 4 kg + r
    {
        const int c0 = 16 * wave + 4 * kg;
        const f32x4v sc = f32x4v{scale[c0], scale[c0 + 1], scale[c0 + 2], scale[c0 + 3]};
        const f32x4v sh = f32x4v{shift[c0], shift[c0 + 1], shift[c0 + 2], shift[c0 + 3]};
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = acc[pb][r] * sc[r] + sh[r];
                if (ACT == 1) o = fmaxf(o, 0.f);
                if (ACT == 2) o = o > 0.f ? o : o * slope;
                v[r] = o;
            }
            unsigned h0, l0, h1, l1;
            gdm_split2(v[0], v[1], h0, l0);
            gdm_split2(v[2], v[3], h1, l1);
            // channels c0 .. c0+3: chunk c0 / 8 = 2 wave + (kg >> 1), bytes 8 (kg & 1) .. of the chunk
            unsigned char* row = abuf + fp_chunk(16 * pb + l16, 2 * wave + (kg >> 1)) + 8 * (kg & 1);
            *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
            unsigned char* rowl = abuf + fp_chunk(16 * pb + l16, 8 + 2 * wave + (kg >> 1)) + 8 * (kg & 1);
            *reinterpret_cast<uint2*>(rowl) = make_uint2(l0, l1);
            acc[pb] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    mma(0);                                                        // `final`: 64 -> 64 on the same pixels

    // ---- + bias, log-softmax over the 64 channels of a pixel: 4 registers x 4 k-groups (lanes l16 + 16 kg) x 4 waves ----
    {
        const int c0 = 16 * wave + 4 * kg;
        f32x4v bq = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (fbias) bq = f32x4v{fbias[c0], fbias[c0 + 1], fbias[c0 + 2], fbias[c0 + 3]};
        float y[2][4], m[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[pb][r] = acc[pb][r] + bq[r];
            m[pb] = fmaxf(fmaxf(y[pb][0], y[pb][1]), fmaxf(y[pb][2], y[pb][3]));
            m[pb] = fmaxf(m[pb], __shfl_xor(m[pb], 16));
            m[pb] = fmaxf(m[pb], __shfl_xor(m[pb], 32));
            if (kg == 0) red[0][wave][16 * pb + l16] = m[pb];
        }
        __syncthreads();
        float ssum[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int p = 16 * pb + l16;
            m[pb] = fmaxf(fmaxf(red[0][0][p], red[0][1][p]), fmaxf(red[0][2][p], red[0][3][p]));
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) q += expf(y[pb][r] - m[pb]);
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            if (kg == 0) red[1][wave][p] = q;
            ssum[pb] = q;
        }
        __syncthreads();
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int p = 16 * pb + l16;
            const float tot = (red[1][0][p] + red[1][1][p]) + (red[1][2][p] + red[1][3][p]);
            const float lse = m[pb] + logf(tot);
            if (n0 + p < N) {
                float* ob = out + ((long)b * UF_C + c0) * N + n0 + p;
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long)r * N] = y[pb][r] - lse;
            }
        }
        (void)ssum;
    }
}

// The 64-channel point->pixel fusion (ffb6d.py:216-222,252-258) on the matrix cores: y[b,co,j] = act(scale[co] * (sum_ci W[co,ci] x[b,ci,j]
// + t[b,co,idx[b,j]]) + shift[co]).  conv1x1_gather_add_act_kernel (gdm_image.hip) does the K = 64 mix with 4096 fp32 FMAs per pixel
// and is bound by them (58 us at 128^2 x 16); here a workgroup splits 64 pixels x 64 channels into operand rows in LDS, a wave forms
// 16 output channels x 64 pixels with 24 MFMAs (split-bf16 x3) and adds the gathered point term on the accumulators, which leaves the
// read and the write of the map.  Output NCHW, or pixel-major [B, m, 64] for the sampled-pixel final stage.
constexpr int CG_P = 64;
// TPM: the point term arrives point-major, t f32[B, n, 64] (one 16-byte load per pixel and lane instead of four scattered floats)
template <int ACT, bool PM, bool TPM>
__global__ __launch_bounds__(256, 4) void conv64_gather_add_act_mfma_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wpk,
                                                                          const float* __restrict__ t, const int32_t* __restrict__ idx,
                                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                                          int n, int m, float slope, float* __restrict__ y,
                                                                          // optional (NCHW form): the result also as the packed split-bf16
                                                                          // operand of the next convolution over the [B, 64, m / W, W] map
                                                                          unsigned char* __restrict__ ypk = nullptr, int W = 0)
{
    __shared__ __attribute__((aligned(16))) unsigned char rows[CG_P * UF_ROWB];      // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y, j0 = blockIdx.x * CG_P;
    const float* xb = x + (long)b * UF_C * m;
    // A fragments (weights): rows 16 wave + l16, k-step S: channels 32 S + 8 kg ..
    u32x4 wh[2], wl[2];
    {
        const unsigned char* r = wpk + (long)(16 * wave + l16) * UF_ROWB;
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            wh[S] = *reinterpret_cast<const u32x4*>(r + (4 * S + kg) * 16);
            wl[S] = *reinterpret_cast<const u32x4*>(r + (8 + 4 * S + kg) * 16);
        }
    }
    // operand rows: thread = (pixel, 8-channel group), lanes = consecutive pixels; loads without control flow
    {
        float raw[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = i * 256 + tid;
            const int p = item & (CG_P - 1), grp = item >> 6;
            const int pp = min(j0 + p, m - 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) raw[i][c] = xb[(long)(grp * 8 + c) * m + pp];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = i * 256 + tid;
            const int p = item & (CG_P - 1), grp = item >> 6;
            unsigned hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) gdm_split2(raw[i][2 * c], raw[i][2 * c + 1], hi[c], lo[c]);
            *reinterpret_cast<u32x4*>(rows + uf_off(p, grp)) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(rows + uf_off(p, 8 + grp)) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    }
    // the gathered point term of this lane's pixels and channels, in flight during the products
    const int c0 = 16 * wave + 4 * kg;
    float tv[4][4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int j = min(j0 + 16 * pb + l16, m - 1);
        int src = idx[(long)b * m + j];
        src = min(max(src, 0), n - 1);
        if (TPM) {
            const float4 q = *reinterpret_cast<const float4*>(t + ((long)b * n + src) * UF_C + c0);
            tv[pb][0] = q.x; tv[pb][1] = q.y; tv[pb][2] = q.z; tv[pb][3] = q.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[pb][r] = t[((long)b * UF_C + c0 + r) * n + src];
        }
    }
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc[r] = scale[c0 + r];
        sh[r] = shift[c0 + r];
    }
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) float f32x4w;
    f32x4w acc[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        acc[pb] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rows + uf_off(16 * pb + l16, 4 * S + kg)));
            const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rows + uf_off(16 * pb + l16, 8 + 4 * S + kg)));
            const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[S]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, wl[S]);
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc[pb], 0, 0, 0);
        }
    }
    // accumulator: lane column l16 = pixel 16 pb + l16, registers r = channels c0 + r
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int j = j0 + 16 * pb + l16;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = sc[r] * (acc[pb][r] + tv[pb][r]) + sh[r];
            if (ACT == 1) v = fmaxf(v, 0.f);
            if (ACT == 2) v = v > 0.f ? v : v * slope;
            o[r] = v;
        }
        if (!PM && ypk) {
            // lanes kg and kg ^ 1 hold the two halves of an 8-channel group of this pixel: the even one writes the 16-byte granules
            float p4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = __shfl_xor(o[r], 16);
            if (j < m && !(kg & 1)) {
                const int H = m / W, yy = j / W, xx = j - yy * W;
                const long plane = (long)(H + 2) * (W + 2);
                const int q = 2 * wave + (kg >> 1);
                unsigned char* op = ypk + ((((long)b * 32) + q) * plane + (long)(yy + 1) * (W + 2) + xx + 1) * 16;
                unsigned hi[4], lo[4];
                gdm_split2(o[0], o[1], hi[0], lo[0]); gdm_split2(o[2], o[3], hi[1], lo[1]);
                gdm_split2(p4[0], p4[1], hi[2], lo[2]); gdm_split2(p4[2], p4[3], hi[3], lo[3]);
                *reinterpret_cast<uint4*>(op) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(op + 16 * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
        if (j >= m) continue;
        if (PM) {
            *reinterpret_cast<float4*>(y + ((long)b * m + j) * UF_C + c0) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[((long)b * UF_C + c0 + r) * m + j] = o[r];
        }
    }
}

// w f32[R, 64] -> R rows of 256 B: 64 bf16 hi | 64 bf16 lo
__global__ __launch_bounds__(256) void pack_rows64_kernel(const float* __restrict__ w, int R, unsigned char* __restrict__ out)
{
    const int item = blockIdx.x * 256 + threadIdx.x;           // (row, pair of channels)
    if (item >= R * 32) return;
    const int row = item >> 5, c = (item & 31) * 2;
    unsigned hi, lo;
    gdm_split2(w[(long)row * UF_C + c], w[(long)row * UF_C + c + 1], hi, lo);
    unsigned* r = reinterpret_cast<unsigned*>(out + (long)row * UF_ROWB);
    r[c >> 1] = hi;
    r[32 + (c >> 1)] = lo;
}

inline float uf_scale_ac(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

} // namespace

extern "C" size_t gdm_upconv_fused64_weight_bytes(void) { return (size_t)9 * UF_C * UF_ROWB; }

extern "C" int gdm_upconv_fused64_pack_weight_hip(const float* w, void* wpk, void* stream)
{
    GDM_CHECK_ARG(w && wpk, "gdm_upconv_fused64_pack_weight_hip: NULL pointer");
    hipLaunchKernelGGL(upconv_pack_w_kernel, dim3(gdm_cdiv(9 * UF_C * UF_C, 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)wpk);
    return gdm_launch_status("upconv_pack_w_kernel");
}

extern "C" int gdm_upconv_fused64_hip(const float* x, const void* wpk, const float* scale, const float* shift, int B, int C, int H, int W,
                                      int OH, int OW, int act, float slope, float* out, void* stream)
{
    GDM_CHECK_ARG(x && wpk && scale && shift && out, "gdm_upconv_fused64_hip: NULL pointer");
    GDM_CHECK_ARG(C == UF_C, "gdm_upconv_fused64_hip: C=%d, built for 64 -> 64 channels", C);
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && OH >= 1 && OW >= 1 && act >= 0 && act <= 2, "gdm_upconv_fused64_hip: bad shape");
    const float rh = uf_scale_ac(H, OH), rw = uf_scale_ac(W, OW);
    hipStream_t s = (hipStream_t)stream;
    {
        // worst-case source window of a tile edge: (edge + 1) output steps of size r, + the +1 neighbour, + rounding
        GDM_CHECK_ARG((int)(rh * (UT_TY + 1)) + 3 <= UT_PH && (int)(rw * (UT_TX + 1)) + 3 <= UT_PWV,
                      "gdm_upconv_fused64_hip: scale factors %g x %g need a source patch larger than %dx%d", rh, rw, UT_PH, UT_PWV);
        static bool attr_t = false;
        if (!attr_t) {
            (void)hipFuncSetAttribute((const void*)upconv_tile64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, UT_LDS);
            (void)hipFuncSetAttribute((const void*)upconv_tile64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, UT_LDS);
            (void)hipFuncSetAttribute((const void*)upconv_tile64_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, UT_LDS);
            attr_t = true;
        }
        dim3 gt(gdm_cdiv(OW, UT_TX), gdm_cdiv(OH, UT_TY * UT_NT), B);
        GDM_CHECK_ARG(gt.y <= 65535, "gdm_upconv_fused64_hip: OH=%d too large", OH);
        if (act == 0) hipLaunchKernelGGL(upconv_tile64_kernel<0>, gt, dim3(256), UT_LDS, s, x, (const unsigned char*)wpk, scale, shift, H, W, OH, OW, rh, rw, slope, out);
        else if (act == 1) hipLaunchKernelGGL(upconv_tile64_kernel<1>, gt, dim3(256), UT_LDS, s, x, (const unsigned char*)wpk, scale, shift, H, W, OH, OW, rh, rw, slope, out);
        else hipLaunchKernelGGL(upconv_tile64_kernel<2>, gt, dim3(256), UT_LDS, s, x, (const unsigned char*)wpk, scale, shift, H, W, OH, OW, rh, rw, slope, out);
        return gdm_launch_status("upconv_tile64_kernel");
    }
}

extern "C" int gdm_pack_rows64_hip(const float* w, int R, void* out, void* stream)
{
    GDM_CHECK_ARG(w && out && R >= 1, "gdm_pack_rows64_hip: bad arguments");
    hipLaunchKernelGGL(pack_rows64_kernel, dim3(gdm_cdiv((long)R * 32, 256)), dim3(256), 0, (hipStream_t)stream, w, R, (unsigned char*)out);
    return gdm_launch_status("pack_rows64_kernel");
}

extern "C" int gdm_upconv_final_points_hip(const float* xpm, const int32_t* choose, const void* wpk, const float* scale, const float* shift,
                                           int act, float slope, const void* wfpk, const float* fbias, int B, int H, int W, int OH, int OW,
                                           int N, float* out, void* stream)
{
    GDM_CHECK_ARG(xpm && choose && wpk && scale && shift && wfpk && out, "gdm_upconv_final_points_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && OH >= 1 && OW >= 1 && N >= 1 && act >= 0 && act <= 2 &&
                  (long)OH * OW <= 0x7fffffffL, "gdm_upconv_final_points_hip: bad shape");
    const float rh = uf_scale_ac(H, OH), rw = uf_scale_ac(W, OW);
    dim3 grid(gdm_cdiv(N, FP_P), B);
    hipStream_t s = (hipStream_t)stream;
#define FPK(A) hipLaunchKernelGGL(upconv_final_points_kernel<A>, grid, dim3(256), 0, s, xpm, choose, (const unsigned char*)wpk, scale, shift, (const unsigned char*)wfpk, fbias, H, W, OH, OW, N, rh, rw, slope, out)
    if (act == 0) FPK(0); else if (act == 1) FPK(1); else FPK(2);
#undef FPK
    return gdm_launch_status("upconv_final_points_kernel");
}

extern "C" int gdm_conv64_gather_add_act_mfma2_hip(const float* x, const void* wpk, const float* t, const int32_t* idx, const float* scale,
                                                   const float* shift, int B, int n, long m, int act, float slope, int pixel_major,
                                                   int t_point_major, float* y, void* ypk, int W, void* stream);

extern "C" int gdm_conv64_gather_add_act_mfma_hip(const float* x, const void* wpk, const float* t, const int32_t* idx, const float* scale,
                                                  const float* shift, int B, int n, long m, int act, float slope, int pixel_major,
                                                  int t_point_major, float* y, void* stream)
{
    return gdm_conv64_gather_add_act_mfma2_hip(x, wpk, t, idx, scale, shift, B, n, m, act, slope, pixel_major, t_point_major, y, nullptr, 0, stream);
}

extern "C" int gdm_conv64_gather_add_act_mfma2_hip(const float* x, const void* wpk, const float* t, const int32_t* idx, const float* scale,
                                                   const float* shift, int B, int n, long m, int act, float slope, int pixel_major,
                                                   int t_point_major, float* y, void* ypk, int W, void* stream)
{
    GDM_CHECK_ARG(x && wpk && t && idx && scale && shift && y, "gdm_conv64_gather_add_act_mfma_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1 && m >= 1 && m <= 0x7fffffffL && act >= 0 && act <= 2, "gdm_conv64_gather_add_act_mfma_hip: bad shape");
    GDM_CHECK_ARG(!ypk || (!pixel_major && W >= 1 && m % W == 0 && ((uintptr_t)ypk & 15) == 0),
                  "gdm_conv64_gather_add_act_mfma2_hip: packed output goes with the NCHW form of an [m / W, W] map (m=%ld W=%d)", m, W);
    unsigned char* ypk8 = (unsigned char*)ypk;
    dim3 grid(gdm_cdiv(m, CG_P), B);
    hipStream_t s = (hipStream_t)stream;
#define CGM(A, P, T) hipLaunchKernelGGL((conv64_gather_add_act_mfma_kernel<A, P, T>), grid, dim3(256), 0, s, x, (const unsigned char*)wpk, t, idx, scale, shift, n, (int)m, slope, y, ypk8, W)
#define CGA(A) do { if (pixel_major) { if (t_point_major) CGM(A, true, true); else CGM(A, true, false); } \
                    else { if (t_point_major) CGM(A, false, true); else CGM(A, false, false); } } while (0)
    if (act == 0) CGA(0); else if (act == 1) CGA(1); else CGA(2);
#undef CGA
#undef CGM
    return gdm_launch_status("conv64_gather_add_act_mfma_kernel");
}
