// The ResNet stem in one kernel, gfx950 (inference): conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BatchNorm + ReLU + max-pool
// 3x3 / stride 2 / pad 1 of /root/reference/models/cnn/extractors.py:112-116,181-185 (conv1, bn1, relu, maxpool), which the
// reference runs as four cuDNN / elementwise launches and this package used to run as an MIOpen convolution (an ASM Winograd kernel
// picked by find mode, ~113 us at batch 16) + one fused BN/ReLU/pool launch.  The full-resolution 64-channel map (67 MB at batch
// 16) never reaches HBM: a workgroup owns a 4 x 16 tile of POOLED pixels, computes the 9 x 33 convolution pixels under it as an
// implicit GEMM on split-bf16 MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulate: the trunk's arithmetic), applies BN + ReLU, pools
// through LDS and writes the pooled map -- as fp32 NCHW (the residual branch of layer1 reads it) and, optionally, as the packed
// split-bf16 operand planes of layer1's first convolution (gdm_conv.hip), which saves that layer's pack launch.
//
// GEMM shape per workgroup: M = 297 convolution pixels (19 fragments of 16), N = 64 channels (4 fragments), K = (ky, c, kx) with kx
// padded 7 -> 8, so that the 8 consecutive k of one MFMA operand lane are 8 consecutive input columns of one row of one colour
// plane: K = 21 groups of 8 = 168, six k-steps of 32 (the last holds one live group).  The input patch (3 x 23 x 72 floats) sits in
// LDS; a lane builds its A fragment with four 8-byte LDS reads + the hi / lo split; the B fragments (weights, pre-packed in fragment
// order by stem_pack_w_kernel: 48 KB) come straight from global memory / L2 into registers, one k-step ahead.
#include "gdm_common.h"

#ifndef GDM_STEM_ABL
#define GDM_STEM_ABL 0
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int SP_H = 4, SP_W = 16;                 // pooled tile
constexpr int SC_H = 2 * SP_H + 1, SC_W = 2 * SP_W + 1;   // convolution pixels under it: 9 x 33
constexpr int SC_PIX = SC_H * SC_W;                // 297
constexpr int SM_FRAGS = (SC_PIX + 15) / 16;       // 19
constexpr int SI_H = 2 * SC_H + 5, SI_W = 72;      // input patch rows (23) and padded row length (2 * 32 + 8 columns used)
constexpr int S_STEPS = 6, S_GROUPS = 21;
constexpr int ST_STRIDE = 305;                     // floats per channel row of the convolution tile in LDS
constexpr int S_WBYTES = S_STEPS * 4 * 2 * 64 * 16;

// w f32[64,3,7,7] -> fragment-ordered split-bf16 weights: (((step*4 + nf)*2 + part)*64 + lane) * 16 B, lane l = channel 16 nf +
// (l & 15), k-group G = 4 step + (l >> 4) = ky*3 + c, the 8 values = kx 0..6 and a zero
__global__ __launch_bounds__(256) void stem_pack_w_kernel(const float* __restrict__ w, unsigned char* __restrict__ out)
{
    const int item = blockIdx.x * 256 + threadIdx.x;          // (step, nf, lane)
    if (item >= S_STEPS * 4 * 64) return;
    const int lane = item & 63, nf = (item >> 6) & 3, step = item >> 8;
    const int G = 4 * step + (lane >> 4), co = 16 * nf + (lane & 15);
    float v[8];
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) v[kx] = 0.f;
    if (G < S_GROUPS) {
        const int ky = G / 3, c = G - ky * 3;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) v[kx] = w[((co * 3 + c) * 7 + ky) * 7 + kx];
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
    unsigned char* o = out + ((long)((step * 4 + nf) * 2) * 64 + lane) * 16;
    *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(o + 64 * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wpk,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int H, int W,
                                                   int OH, int OW, int PH, int PW, float* __restrict__ out, unsigned char* __restrict__ outpk)
{
    __shared__ __attribute__((aligned(16))) float patch[3][SI_H][SI_W];       // 19.9 KB
    __shared__ __attribute__((aligned(16))) float ct[32][ST_STRIDE];          // 39 KB: relu(bn(conv)) of the 9 x 33 pixels, 32 channels at a
                                                                              // time (two passes): 59 KB in all, two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z;
    const int py0 = blockIdx.y * SP_H, px0 = blockIdx.x * SP_W;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;             // convolution pixel of tile position (0, 0)
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;             // input pixel of patch position (0, 0)

    // ---- input patch -> LDS (zero outside the image = the convolution's padding) ----
    // every load of the thread is issued before any is used (20 per thread: as a loop of load -> store the fill was twenty dependent
    // memory round trips, most of the workgroup's life: 64 -> 51 us per launch); out-of-image elements read a clamped address and are zeroed
    {
        constexpr int NE = (3 * SI_H * SI_W + 255) / 256;
        float v[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = min(tid + 256 * i, 3 * SI_H * SI_W - 1);
            const int col = e % SI_W, r = (e / SI_W) % SI_H, c = e / (SI_W * SI_H);
            const int iy = iy0 + r, ix = ix0 + col;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float t = x[(((long)b * 3 + c) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
            v[i] = in ? t : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            if (e < 3 * SI_H * SI_W) (&patch[0][0][0])[e] = v[i];
        }
    }
#if GDM_STEM_ABL & 8
    return;
#endif

    // ---- B fragments of k-step 0 (registers), the wave's A-fragment coordinates ----
    u32x4 bh[4], bl[4], nbh[4], nbl[4];
    auto load_b = [&](int step, u32x4 (&h)[4], u32x4 (&l)[4]) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const unsigned char* p = wpk + ((long)((step * 4 + nf) * 2) * 64 + lane) * 16;
            h[nf] = *reinterpret_cast<const u32x4*>(p);
            l[nf] = *reinterpret_cast<const u32x4*>(p + 64 * 16);
        }
    };
    load_b(0, bh, bl);
    constexpr int MPW = (SM_FRAGS + 3) / 4;                      // M fragments per wave (5; the last wave owns 4)
    int aoff[MPW];                                               // float offset of (row 2 cy, column 2 cx) inside a colour plane
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
        const int p = min((wave + 4 * j) * 16 + l16, SC_PIX - 1);
        const int cy = p / SC_W, cx = p - cy * SC_W;
        aoff[j] = (2 * cy) * SI_W + 2 * cx;
    }
    f32x4 acc[MPW][4];
#pragma unroll
    for (int j = 0; j < MPW; ++j)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][nf][i] = 0.f;
    __syncthreads();

    // ---- implicit GEMM: six k-steps of 32 ----
#pragma unroll
    for (int step = 0; step < S_STEPS; ++step) {
        if (step + 1 < S_STEPS) load_b(step + 1, nbh, nbl);
        const int G = 4 * step + kg;                             // this lane's k-group: (ky, c), columns 2 cx .. 2 cx + 7
        const bool live = G < S_GROUPS;
        const int ky = live ? G / 3 : 0, c = live ? G - ky * 3 : 0;
        const float* prow = &patch[c][ky][0];
#pragma unroll
        for (int j = 0; j < MPW; ++j) {
            if (wave + 4 * j >= SM_FRAGS) continue;              // uniform per wave
            float v[8];
            const float2* src = reinterpret_cast<const float2*>(prow + aoff[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 t = src[q];
                v[2 * q] = live ? t.x : 0.f;
                v[2 * q + 1] = live ? t.y : 0.f;
            }
            unsigned hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gdm_split2(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
            const u32x4 ahv = {hi[0], hi[1], hi[2], hi[3]}, alv = {lo[0], lo[1], lo[2], lo[3]};
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ahv), al = __builtin_bit_cast(bf16x8, alv);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                f32x4 cacc = acc[j][nf];
#if !(GDM_STEM_ABL & 1)
                cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bl[nf]), cacc, 0, 0, 0);
                cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, bh[nf]), cacc, 0, 0, 0);
#endif
                cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bh[nf]), cacc, 0, 0, 0);
                acc[j][nf] = cacc;
            }
        }
        if (step + 1 < S_STEPS) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                bh[nf] = nbh[nf];
                bl[nf] = nbl[nf];
            }
        }
    }

    // ---- BN + ReLU, convolution pixels outside the map -> 0 (below every ReLU output a pool window also holds), into LDS; then the
    // max-pool 3x3 / 2 / pad 1 with thread = pooled pixel (tid & 63) x 8 channels (wave); two passes of 32 channels ----
#if GDM_STEM_ABL & 4
    if (acc[0][0][0] != 12345.678f) return;
#endif
    const int pxl = tid & 15, pyl = (tid >> 4) & 3;
    const int py = py0 + pyl, px = px0 + pxl;
    const bool store = py < PH && px < PW;
    const long plane = (long)(PH + 2) * (PW + 2);
    // which of this lane's 4 MPW convolution pixels lie inside the tile and inside the map (the same for every channel and pass)
    unsigned inmap = 0, intile = 0;
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
        const int p0 = (wave + 4 * j) * 16 + 4 * kg;
        int cy = p0 / SC_W, cx = p0 - cy * SC_W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = oy0 + cy, ox = ox0 + cx;
            if (p0 + r < SC_PIX && wave + 4 * j < SM_FRAGS) intile |= 1u << (4 * j + r);
            if (oy >= 0 && oy < OH && ox >= 0 && ox < OW) inmap |= 1u << (4 * j + r);
            if (++cx == SC_W) { cx = 0; ++cy; }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();                               // the first pass's readers are done with ct
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const int nf = 2 * pass + nn;
            const int co = 16 * nf + l16;
            const float sc = scale[co], sh = shift[co];
            float* crow = &ct[16 * nn + l16][4 * kg];
#pragma unroll
            for (int j = 0; j < MPW; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!(intile >> (4 * j + r) & 1)) continue;
                    const float v = fmaxf(fmaf(acc[j][nf][r], sc, sh), 0.f);
                    crow[(wave + 4 * j) * 16 + r] = (inmap >> (4 * j + r) & 1) ? v : 0.f;
                }
            }
        }
        __syncthreads();
        float pooled[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* row = &ct[8 * wave + i][(2 * pyl) * SC_W + 2 * pxl];
            float m = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, row[dy * SC_W + dx]);
            pooled[i] = m;
        }
        if (!store) continue;
        const int c0 = 32 * pass + 8 * wave;                     // this thread's eight channels
        if (out) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out[(((long)b * 64 + c0 + i) * PH + py) * PW + px] = pooled[i];
        }
        if (outpk) {
            // packed operand of the next 3x3 convolution (gdm_conv.hip conv_pack_act_kernel's layout, one 64-channel chunk): plane
            // q < 8 = bf16 hi of channels [8q, 8q + 8), plane 16 + q = their lo; element (py + 1, px + 1) of the zero-bordered grid
            const int q = c0 / 8;
            unsigned char* ob = outpk + (((long)b * 32) * plane + (long)(py + 1) * (PW + 2) + px + 1) * 16;
            unsigned hi[4], lo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) gdm_split2(pooled[2 * jj], pooled[2 * jj + 1], hi[jj], lo[jj]);
            *reinterpret_cast<uint4*>(ob + (long)q * plane * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(ob + (long)(16 + q) * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
}

} // namespace

extern "C" size_t gdm_stem_weight_bytes(void) { return S_WBYTES; }

extern "C" int gdm_stem_pack_weight_hip(const float* w, void* wpk, void* stream)
{
    GDM_CHECK_ARG(w && wpk, "gdm_stem_pack_weight_hip: NULL pointer");
    GDM_CHECK_ARG(((uintptr_t)wpk & 15) == 0, "gdm_stem_pack_weight_hip: wpk must be 16-byte aligned");
    hipLaunchKernelGGL(stem_pack_w_kernel, dim3(gdm_cdiv(S_STEPS * 4 * 64, 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)wpk);
    return gdm_launch_status("stem_pack_w_kernel");
}

extern "C" int gdm_stem_hip(const float* x, const void* wpk, const float* scale, const float* shift, int B, int H, int W, float* out,
                            void* out_packed, void* stream)
{
    GDM_CHECK_ARG(x && wpk && scale && shift && (out || out_packed), "gdm_stem_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && H >= 7 && W >= 7, "gdm_stem_hip: bad shape B=%d H=%d W=%d", B, H, W);
    GDM_CHECK_ARG(((uintptr_t)wpk & 15) == 0 && ((uintptr_t)out_packed & 15) == 0, "gdm_stem_hip: packed buffers must be 16-byte aligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;      // conv 7x7 / 2 / pad 3
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;    // max-pool 3x3 / 2 / pad 1
    dim3 grid(gdm_cdiv(PW, SP_W), gdm_cdiv(PH, SP_H), B);
    GDM_CHECK_ARG(grid.y <= 65535, "gdm_stem_hip: map too tall");
    hipLaunchKernelGGL(stem_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (const unsigned char*)wpk, scale, shift, H, W, OH, OW,
                       PH, PW, out, (unsigned char*)out_packed);
    return gdm_launch_status("stem_kernel");
}
