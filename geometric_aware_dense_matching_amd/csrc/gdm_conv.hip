// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on split-bf16 MFMA, gfx950.
//
// For the 32x32-resolution half of the ResNet-18 trunk (layer3 / layer4 of
// /root/reference/models/cnn/extractors.py:36-58,151-177: 128..512 -> 256..512 channels), where MIOpen's
// fp32 path is an ASM Winograd F(2,3) kernel on the vector ALUs (~118 TF/s direct-equivalent).  fp32 products are
// replaced by three bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate; |err| <= 3*2^-18 per product, the same
// scheme as the descriptor matching kernel), which is ~5x the f32-MFMA rate.
//
//   out[b,co,y,x] = act( scale[co] * sum_{tap,ci} W[co,ci,tap] * in[b,ci,y+ky-1,x+kx-1] + shift[co] (+ res) )
//
// Data layout: activations are re-packed once per layer (pack kernel below) into bf16 hi / lo planes of 8 channels with a
// one-pixel zero border (fragment-planar, see conv_pack_act_kernel), so a tap is just a pixel offset, the zero padding is
// real zeros and every operand load of a wave is two contiguous 512-B runs.  Weights are packed once per weight change into
// rows (tap, chunk, co) of the same 512-B format.  Then the kernel is the matching kernel's shape: a wave keeps
// 32 pixels x 128 channels of one (tap, chunk) in 64 VGPRs as the MFMA A operand, 128 output channels of that
// (tap, chunk) sit in a double-buffered, XOR-swizzled LDS panel as the B operand (global -> registers before the
// MFMAs, registers -> LDS after them, ONE barrier per panel), 9 * Cin/128 panels accumulate into four 32x32 tiles.
#include "gdm_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int ROWB = 512;                 // bytes per packed row (128 channels, hi | lo)
constexpr int CV_PIX = 256;               // pixels per workgroup (32 per wave)
constexpr int CV_CO = 128;                // output channels per workgroup
constexpr int CV_PANEL = CV_CO * ROWB;    // 64 KiB
#ifndef GDM_CONV_TAP_INNER
#define GDM_CONV_TAP_INNER 1             // 1: panels chunk-major (nine taps of a chunk back to back); 0: tap-major (rounds 1-3)
#endif
#ifndef GDM_CONV16_PF
#define GDM_CONV16_PF 2
#endif
#ifndef GDM_CONV_GLDS
#define GDM_CONV_GLDS 1                  // 1: weight panels of the 16x16x32 kernel by LDS-DMA (eight-wave workgroups); 0: through registers
#endif

__device__ __forceinline__ void split8(const float* v, unsigned (&hi)[4], unsigned (&lo)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
}

// x f32[B,C,H,W] -> packed activations, PLANAR by 16-byte MFMA fragment: for every (b, 128-channel chunk) 32 planes of the
// zero-bordered pixel grid, plane q < 16 = bf16 "hi" of channels [8q, 8q+8), plane 16+q = their "lo"; element (yy, xx) of a
// plane is 16 B at ((b*nchunk + chunk)*32 + plane) * (H+2)(W+2) + yy*(W+2) + xx.  A wave's operand load (lane = pixel, one
// fragment) is then two contiguous 512-B runs for ANY tap shift, instead of 32 scattered 32-B pieces of pixel-major rows.
// one block: 64 consecutive pixels of one image x one 128-channel chunk
__global__ __launch_bounds__(256) void conv_pack_act_kernel(const float* __restrict__ x, int C, int H, int W,
                                                            unsigned char* __restrict__ out)
{
    __shared__ float t[128][65];
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int nchunk = (C + 127) / 128;
    const int cvalid = min(128, C - chunk * 128);          // a trailing partial chunk: the missing channels' planes stay zero
    const int hw = H * W;
    const long plane = (long)(H + 2) * (W + 2);
    const int p0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int p = min(p0 + lane, hw - 1);
    const float* xb = x + ((long)b * C + chunk * 128) * hw;
    for (int c = w; c < cvalid; c += 4) t[c][lane] = xb[(long)c * hw + p];
    __syncthreads();
    const int pp = p0 + lane;
    if (pp >= hw) return;
    const int y = pp / W, xx = pp - y * W;
    unsigned char* o = out + (((long)(b * nchunk + chunk) * 32) * plane + (long)(y + 1) * (W + 2) + xx + 1) * 16;
    for (int q = w; q * 8 < cvalid; q += 4) {            // consecutive lanes = consecutive pixels: 1-KiB contiguous stores
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t[q * 8 + j][lane];
        unsigned hi[4], lo[4];
        split8(v, hi, lo);
        *reinterpret_cast<uint4*>(o + (long)q * plane * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + (long)(16 + q) * plane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// w f32[Cout,Cin,3,3] -> rows ((tap*nchunk + chunk)*CoutP + co): channels [128 chunk, +128) of tap (ky,kx); CoutP = Cout rounded
// up to 128, rows co >= Cout are zero (a workgroup always stages 128 rows)
// dgrad: w is the FORWARD weight f32[Cin, Cout, taps] of the layer whose input gradient is wanted; the rows written are those of the
// flipped, transposed filter w'[co][ci][tap] = w[ci][co][taps - 1 - tap] (the input gradient of a 3x3/s1/p1 or 1x1 convolution is the same
// convolution of grad_out with w') -- no flip / transpose / contiguous copies in front of the pack.
__global__ __launch_bounds__(256) void conv_pack_w_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, unsigned char* __restrict__ out,
                                                          int dgrad = 0)
{
    const int nchunk = (Cin + 127) / 128;
    const int CoutP = (Cout + 127) & ~127;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;       // (row, 16 groups of 8 channels)
    const long rows = (long)taps * nchunk * CoutP;
    if (item >= rows * 16) return;
    const int ch = (int)(item & 15);
    const long row = item >> 4;
    const int co = (int)(row % CoutP);
    const int tc = (int)(row / CoutP);
    const int chunk = tc % nchunk, tap = tc / nchunk;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = chunk * 128 + ch * 8 + j;
        const bool in = co < Cout && ci < Cin;
        v[j] = !in ? 0.f : dgrad ? w[((long)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((long)co * Cin + ci) * taps + tap];
    }
    unsigned hi[4], lo[4];
    split8(v, hi, lo);
    unsigned char* r = out + row * ROWB;
    *reinterpret_cast<uint4*>(r + ch * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(r + 256 + ch * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__device__ __forceinline__ int swz(int col, int ch) { return col * ROWB + (((ch & 16) | ((ch ^ col) & 15)) << 4); }

// ---------------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16.  Same workgroup tile (256 pixels x 128 output channels), same panels, same LDS
// image, same operand bytes per MFMA cycle (a wave's 32 x 128 tile is 2 x 8 tiles of 16 x 16; a k-step is 32 channels: two
// A fragments per (hi, lo), eight B fragments); what changes is the lane <-> (row, k-group) map of the fragments and of the
// accumulators.  MI355X holds a visibly higher clock in 16x16x32 loops than in 32x32x16 loops at equal cycles per FLOP
// (MI355X_MICROARCH.md, DVFS item 7: 1.12-1.15x on LDS-fed loops), which is the whole reason for this variant.
//   A fragment (S, ph): lane l -> pixel (l & 15) + 16 ph, channels 8 (4 S + (l >> 4)) .. + 8     = plane 4 S + (l >> 4) of the chunk
//   B fragment (S, cb): lane l -> output channel 16 cb + (l & 15), same channel group            = 16-B chunk 4 S + (l >> 4) of its row
//   accumulator (ph, cb): lane l -> channel 16 cb + (l & 15), pixels 16 ph + 4 (l >> 4) + 0..3   = one 16-byte store
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;

// pixel fragments (16 pixels each) per wave: 2 = eight waves of 32 pixels (two waves per SIMD), 4 = four waves of 64 pixels (one
// wave per SIMD, 512 registers): every B fragment read from LDS then feeds twice the MFMAs and no partner wave competes for the
// matrix pipe; the per-panel barrier joins four waves instead of eight.
#ifndef GDM_CONV_NPH
#define GDM_CONV_NPH 2
#endif
// 1: the MFMAs of a unit are issued product-major (no MFMA directly behind the one whose accumulator it reads); 0: accumulator-major.
// Measured equal (512 -> 512: 191 vs 188 us on two boxes; the register-only probe gdm_mfma_probe_hip sustains 2.25 PF/s with independent and
// 2.40 PF/s with back-to-back dependent MFMAs at two waves per SIMD): dependent issue is not what idles the pipe.  0 ships.
// k-steps (of the 4 of a 128-channel panel) whose operand reload is deferred to the top of the next panel
#ifndef GDM_CONV16_LATE
#define GDM_CONV16_LATE 1
#endif
#ifndef GDM_CONV_MFMA_ORDER
#define GDM_CONV_MFMA_ORDER 0
#endif
constexpr int MF_NPH = GDM_CONV_NPH;
constexpr int MF_WPIX = 16 * MF_NPH;                  // pixels per wave
constexpr int MF_WAVES = CV_PIX / MF_WPIX;
constexpr int MF_THREADS = MF_WAVES * 64;
constexpr int MF_TSTRIDE = 132;                       // floats per pixel row of a wave's output tile in LDS (128 + 4: conflict-free reads)
constexpr int MF_SMEM = 2 * CV_PANEL > CV_PIX * MF_TSTRIDE * 4 ? 2 * CV_PANEL : CV_PIX * MF_TSTRIDE * 4;

// NCB = 16-channel output blocks per workgroup: 8 (128 channels) or 4 (64 channels: twice the workgroups for the layers whose 128-channel
// tiling leaves half the chip idle -- layer1-3 of the trunk at batch 16 -- and no zero-padded weight rows for 64-channel layers)
// WV = waves per workgroup: 8 (256 pixels), or 4 (128 pixels: the 128 -> 128 layers at 32 x 32 have 128 workgroups of the 256-pixel, 64-channel
// tiling on 256 CUs -- half tiles put one on every CU)
template <int ACT, bool HAS_RES, int TAPS = 9, bool PIXMAJOR = false, int NKS = 8, int NCB = 8, int WV = MF_WAVES>
__global__ __launch_bounds__(MF_THREADS) void conv_mfma16_kernel(const unsigned char* __restrict__ xpk, const unsigned char* __restrict__ wpk,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ res, int B, int Cin, int Cout, int H, int W,
                                                                 float* __restrict__ out, const int32_t* __restrict__ rowidx = nullptr,
                                                                 const int32_t* __restrict__ tile_co0 = nullptr,
                                                                 unsigned char* __restrict__ outpk = nullptr, int stride = 1,
                                                                 // per-image weights (the weight-gradient GEMM: image b of a workgroup's 256 pixels --
                                                                 // H*W % 256 == 0 -- reads its rows at wpk + b * wbstride); 0 = shared weights
                                                                 long wbstride = 0)
{
    // H, W: OUTPUT map; the packed input is the (H stride) x (W stride) map (stride 2: layer2's first block, extractors.py:151-177)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 x CV_PANEL
    constexpr int NS = NKS / 2;                                     // k-steps of 32 channels per chunk
    constexpr int UG = (NCB % 2 == 0) ? 2 : 3;                      // 16-channel output blocks per unit (NCB = 9: three)
    constexpr int NPR = NCB / UG;                                   // units per k-step
    constexpr int PANEL = NCB * 16 * ROWB;                          // bytes of a weight panel in LDS
    constexpr int TST = NCB * 16 + 4;                               // floats per pixel row of a wave's output tile in LDS
    static_assert(NCB == 8 || (NCB == 4 && !PIXMAJOR) || (NCB == 9 && !PIXMAJOR && WV == 8),
                  "64-channel tiles: NCHW / packed output only; 144-channel tiles: NCHW fp32 output only (no packed output: a tile straddles 128-channel chunks)");
    constexpr int NPH = MF_NPH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform FOR THE COMPILER too: everything derived from it stays scalar
    const int l16 = lane & 15, kg = lane >> 4;                      // row inside a 16-row fragment, 8-channel group inside a k-step
    const int nchunk = (Cin + 127) / 128;
    const int npanel = TAPS * nchunk;
    const int hw = H * W;
    const long ptot = rowidx ? (long)B : (long)B * hw;              // host: ptot % MF_WPIX == 0 (and hw % MF_WPIX == 0 for maps)
    const unsigned bx = blockIdx.x, by = blockIdx.y;
    const long pix0 = (long)bx * (WV * MF_WPIX) + wave * MF_WPIX;           // this wave's first pixel
    const int co0 = tile_co0 ? __builtin_amdgcn_readfirstlane(tile_co0[bx]) : by * (NCB * 16);
    const long pc = min(pix0, ptot - MF_WPIX);
    const int b = (int)(pc / hw);                                   // one image per wave (hw % MF_WPIX == 0)
    const int prem = (int)(pc - (long)b * hw);
    const int Wi = W * stride;
    const long plane = (long)(H * stride + 2) * (Wi + 2);           // input planes (operand loads)
    const long oplane = (long)(H + 2) * (W + 2);                    // output planes (packed epilogue)
    // Operand loads are BUFFER loads: descriptor over the packed activations, per-lane byte offset fixed for the whole kernel
    // (plane kg of a k-step + the lane's pixel), everything that changes per panel / tap / k-step in the SCALAR offset.  The flat
    // form recomputed a 64-bit per-lane address for every load (v_mad_u64 / v_lshl_add_u64 / v_mul_lo clumps of ~45 vector
    // instructions at the head of each half panel, issued by both waves of a SIMD at once right behind the barrier: MFMA pipe idle).
    // The host checks that both packed buffers are below 2 GiB.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xpk), (short)0, 0x7ffffff0, 0x00020000);
    int avoff[NPH];                                                 // fragment ph: 16 consecutive pixels of one image row (W % 16 == 0)
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int pr = prem + 16 * ph;
        const int yy = pr / W, xx = pr - yy * W;
        const long pixbase = rowidx ? (long)max(rowidx[pc + l16 + 16 * ph], 0) : (long)(yy * stride) * (Wi + 2) + (long)(xx + l16) * stride;
        avoff[ph] = (int)(((long)kg * plane + pixbase) * 16);
    }

    u32x4 ahi[NS][NPH], alo[NS][NPH];                               // fragments of k-step S, pixel fragment ph
    // Panel bookkeeping in scalar registers, advanced by counters (no division in the loop): panel = (chunk, ky, kx), chunk-major
    // (GDM_CONV_TAP_INNER) or tap-major.  pan_a / pan_w: byte offsets of a panel's operand rows / weight rows.
    struct Pan { int chunk, ky, kx; };
    auto pan_next = [&](Pan p) -> Pan {
        if (TAPS == 1) { ++p.chunk; return p; }
        if (GDM_CONV_TAP_INNER) {
            if (++p.kx == 3) { p.kx = 0; if (++p.ky == 3) { p.ky = 0; ++p.chunk; } }
        } else {
            if (++p.chunk == nchunk) { p.chunk = 0; if (++p.kx == 3) { p.kx = 0; ++p.ky; } }
        }
        return p;
    };
    const int a_base = (int)(((long)(rowidx ? 0 : b) * nchunk * 32) * plane * 16);
    const int a_chunk = (int)(32 * plane * 16), a_rowb = (Wi + 2) * 16;
    auto pan_a = [&](Pan p) -> int { return a_base + p.chunk * a_chunk + p.ky * a_rowb + p.kx * 16; };
    const int sstride = (int)(4 * plane * 16);                      // plane 4 S + kg -> 4 (S + 1) + kg
    const int lostride = (int)(16 * plane * 16);                    // hi plane q -> lo plane 16 + q
    auto load_a = [&](int r, int S) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            ahi[S][ph] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, avoff[ph], r + S * sstride, 0));
            alo[S][ph] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, avoff[ph], r + S * sstride + lostride, 0));
        }
    };
    // a weight panel is staged in two halves (global -> registers -> LDS), each half in flight for half a panel: 16 registers
    constexpr int THREADS = WV * 64, WGPIX = WV * MF_WPIX;
    constexpr int HS = (NCB * 512 / THREADS) / 2;
    u32x4 stage[HS];
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wpk), (short)0, 0x7ffffff0, 0x00020000);
    const int w_base = (wbstride ? (int)(((long)bx * WGPIX / hw) * wbstride) : 0) + co0 * ROWB;
    const int w_panel = ((Cout + 127) & ~127) * ROWB;               // the packed weights are (tap, chunk, co) rows
    auto pan_w = [&](Pan p) -> int { return w_base + ((TAPS == 1) ? p.chunk : (p.ky * 3 + p.kx) * nchunk + p.chunk) * w_panel; };
    auto stage_load = [&](int src, int half) {
#pragma unroll
        for (int i = 0; i < HS; ++i)
            stage[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, tid * 16, src + (half * HS + i) * THREADS * 16, 0));
    };
    auto stage_store = [&](int buf, int half) {
        unsigned char* base = smem + buf * PANEL;
#pragma unroll
        for (int i = 0; i < HS; ++i) {
            const int g = (half * HS + i) * THREADS + tid;
            *reinterpret_cast<u32x4*>(base + swz(g >> 5, g & 31)) = stage[i];
        }
    };
    // GLDS (eight-wave workgroups): the weight panel goes global -> LDS by LDS-DMA (buffer_load ... lds), no staging registers and no
    // ds_write pass.  A wave instruction fills 1 KiB = two 512-byte rows LINEARLY (LDS address = M0 base + lane * 16), so the XOR
    // swizzle of the LDS image sits on the per-lane SOURCE offset: lane p of piece j = 8 i + wave writes slot p & 31 of row
    // col = 2 j + (p >> 5), which must hold chunk ch = (s & 16) | ((s ^ col) & 15) of that row (swz is an involution); 2 j = 16 i + 2 wave,
    // so col & 15 -- and with it the lane's source offset -- is the same for every piece of a wave.
    constexpr bool GLDS = GDM_CONV_GLDS && WV == 8;
    constexpr int NPIECE = PANEL / 1024 / 8;                        // LDS-DMA instructions per wave and panel
    int glds_voff = 0;
    if (GLDS) {
        const int hrow = lane >> 5, sl = lane & 31;
        const int col15 = (2 * wave + hrow) & 15;
        glds_voff = wave * 1024 + hrow * 512 + (((sl & 16) | ((sl ^ col15) & 15)) << 4);
    }
    typedef __attribute__((address_space(3))) void* lds_void_ptr;
    auto stage_glds = [&](int src, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(smem + buf * PANEL + (i * 8 + wave) * 1024), 16, glds_voff, src + i * 8192, 0, 0);
    };

    f32x4 acc[NPH][NCB];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ph][cb][i] = 0.f;

    constexpr int LATE = GDM_CONV16_LATE < NS ? GDM_CONV16_LATE : NS;                           // k-steps whose reload is deferred to the next iteration's top
    Pan pcur = {0, (TAPS == 1) ? 1 : 0, (TAPS == 1) ? 1 : 0};
    if (GLDS) {
        stage_glds(pan_w(pcur), 0);
    } else {
        stage_load(pan_w(pcur), 0);
        stage_store(0, 0);
        stage_load(pan_w(pcur), 1);
        stage_store(0, 1);
    }
    int rcur = pan_a(pcur);
#pragma unroll
    for (int S = 0; S < NS - LATE; ++S) load_a(rcur, S);
    Pan pnxt = npanel > 1 ? pan_next(pcur) : pcur;
    int rnext = pan_a(pnxt), wnext = pan_w(pnxt);                   // (a harmless re-read of the same rows behind the last panel)
    for (int it = 0; it < npanel; ++it) {
        __syncthreads();                                            // panel `it` is in LDS; panel it-1's readers are done
        const bool more = it + 1 < npanel;
        const unsigned char* base = smem + (it & 1) * PANEL;
        // units of (k-step S, UG 16-channel output blocks): 3 UG NPH MFMAs of 16 cycles on 2 UG B fragments; PF units' reads in flight
        constexpr int PF = GDM_CONV16_PF;
        constexpr int NU = NPR * NS;
        u32x4 fh[PF + 1][UG], fl[PF + 1][UG];
        auto frag_load = [&](int un) {
            const int S = un / NPR, pr = un % NPR, slot = un % (PF + 1);
#pragma unroll
            for (int j = 0; j < UG; ++j) {
                const int col = (UG * pr + j) * 16 + l16;
                fh[slot][j] = *reinterpret_cast<const u32x4*>(base + swz(col, 4 * S + kg));
                fl[slot][j] = *reinterpret_cast<const u32x4*>(base + swz(col, 16 + 4 * S + kg));
            }
        };
#pragma unroll
        for (int un = 0; un < PF; ++un) frag_load(un);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int S = u / NPR, pr = u % NPR, slot = u % (PF + 1);
            if (u + PF < NU) frag_load(u + PF);
#if GDM_CONV_MFMA_ORDER == 0
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi[S][ph]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, alo[S][ph]);
#pragma unroll
                for (int j = 0; j < UG; ++j) {
                    f32x4 c = acc[ph][UG * pr + j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, fl[slot][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, fh[slot][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, fh[slot][j]), c, 0, 0, 0);
                    acc[ph][UG * pr + j] = c;
                }
            }
#else
            // product-major: the unit's 2 NPH accumulators take hi.lo, then lo.hi, then hi.hi -- the same three terms in the same order per
            // accumulator (bit-identical sums), but an MFMA never waits for its predecessor's result (2 NPH - 1 others in between)
#pragma unroll
            for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
                    for (int j = 0; j < UG; ++j) {
                        const bf16x8 av = __builtin_bit_cast(bf16x8, prod == 1 ? alo[S][ph] : ahi[S][ph]);
                        const bf16x8 bv = __builtin_bit_cast(bf16x8, prod == 0 ? fl[slot][j] : fh[slot][j]);
                        acc[ph][UG * pr + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[ph][UG * pr + j], 0, 0, 0);
                    }
#endif
            if (pr == NPR - 1 && S < NS - LATE) load_a(rnext, S);         // this k-step's registers are dead: next panel's data
            if (!GLDS && u == NU / 2 && more) {                     // the other buffer's readers finished at this panel's barrier
                stage_store((it + 1) & 1, 0);
                stage_load(wnext, 1);
            }
            if (u == 0) {
                if (more) { if (GLDS) stage_glds(wnext, (it + 1) & 1); else stage_load(wnext, 0); }
#pragma unroll
                for (int S2 = NS - LATE; S2 < NS; ++S2) load_a(rcur, S2);
            }
#pragma unroll
            for (int i = 0; i < 3 * UG * NPH; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (u + PF < NU && (i % NPH) == 0 && i < 2 * UG * NPH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!GLDS && more) stage_store((it + 1) & 1, 1);
        // the next panels' scalar offsets, formed here -- behind the panel's last MFMA, in front of the barrier -- instead of at the
        // head of the next panel, where both waves of a SIMD would run the same ~40 scalar instructions with the matrix pipe idle
        rcur = rnext;
        if (it + 2 < npanel) pnxt = pan_next(pnxt);
        rnext = pan_a(pnxt);
        wnext = pan_w(pnxt);
    }

    // ---- epilogue: lane = output channel 16 cb + l16, registers = 4 consecutive pixels 16 ph + 4 kg + r ----
    if (pix0 >= ptot) return;
    if (PIXMAJOR) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int co = co0 + cb * 16 + l16;
            if (co >= Cout) continue;
            const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
            const int ocol = tile_co0 ? cb * 16 + l16 : co;
            const int ostride = tile_co0 ? CV_CO : Cout;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ph][cb][r] * sc + sh;
                    if (ACT == 1) v = fmaxf(v, 0.f);
                    out[(pix0 + 16 * ph + 4 * kg + r) * ostride + ocol] = v;
                }
        }
        return;
    }
    float* tl = reinterpret_cast<float*>(smem) + wave * MF_WPIX * TST;
    if (outpk) __syncthreads();                                      // every wave has finished reading the last weight panel
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int co = co0 + cb * 16 + l16;
        const bool live = co < Cout;
        const float sc = (live && scale) ? scale[co] : 1.f, sh = (live && shift) ? shift[co] : 0.f;
        float* op = out ? out + ((long)b * Cout + (live ? co : 0)) * hw + prem : nullptr;
        const float* rp = (HAS_RES && live) ? res + ((long)b * Cout + co) * hw + prem : nullptr;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int poff = 16 * ph + 4 * kg;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[ph][cb][j] * sc + sh;
            if (HAS_RES && live) {
                const float4 r4 = *reinterpret_cast<const float4*>(rp + poff);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (ACT == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (op && live) *reinterpret_cast<float4*>(op + poff) = make_float4(v[0], v[1], v[2], v[3]);
            if (outpk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) tl[(poff + j) * TST + cb * 16 + l16] = live ? v[j] : 0.f;
            }
        }
    }
    if (outpk) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int lr = lane & 31, h = lane >> 5;
        const int ochunks = (Cout + 127) / 128;
        const int ochunk = co0 / 128;
#pragma unroll
        for (int half = 0; half < MF_WPIX / 32; ++half) {
            const int pp = prem + 32 * half + lr;                    // this lane's pixel of the wave's tile
            const int yy = pp / W, xx = pp - yy * W;
            unsigned char* ob = outpk + (((long)(b * ochunks + ochunk) * 32) * oplane + (long)(yy + 1) * (W + 2) + xx + 1) * 16;
            const float* row = tl + (32 * half + lr) * TST;
#pragma unroll
            for (int it = 0; it < NCB; ++it) {
                const int ql = 2 * it + h;                           // 8-channel group inside this workgroup's channels
                if (co0 + ql * 8 >= Cout) continue;
                const int q = (co0 % 128) / 8 + ql;                  // ... and inside the 128-channel chunk of the packed output
                const float4 a0 = *reinterpret_cast<const float4*>(row + ql * 8);
                const float4 a1 = *reinterpret_cast<const float4*>(row + ql * 8 + 4);
                const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                unsigned hi[4], lo[4];
                split8(v, hi, lo);
                *reinterpret_cast<uint4*>(ob + (long)q * oplane * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(ob + (long)(16 + q) * oplane * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
    }
}

// (the v_mfma_f32_32x32x16_bf16 form of rounds 1-3, `conv3x3_bf16x3_kernel`, was removed in round 4: compiled, never launched)
#define CONV_KERNEL conv_mfma16_kernel
constexpr int CONV_THREADS = MF_THREADS, CONV_WPIX = MF_WPIX, CONV_SMEM = MF_SMEM;

} // namespace

// Measurement aid (bench.py): what the matrix pipe sustains on THIS chip with nothing else in the way -- every wave issues `iters` x 8
// independent v_mfma_f32_16x16x32_bf16 on registers (no LDS, no memory), eight waves per CU like the convolution kernel.  The chip lowers
// its clock under such a load (MI355X_MICROARCH.md, DVFS), so this -- not the 2.5 PFLOP/s of the data sheet clock -- is the ceiling an
// MFMA-bound kernel can be compared with on the box it runs on.
namespace {
template <int CHAIN>
__global__ __launch_bounds__(512) void mfma_probe_kernel(int iters, float* __restrict__ sink)
{
    typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float pb_f32x4;
    const int lane = threadIdx.x & 63;
    pb_f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned seed = 0x3f803f80u + (unsigned)lane;             // bf16 pairs near 1.0
    u32x4 av = {seed, seed ^ 1u, seed ^ 2u, seed ^ 3u}, bv = {seed ^ 4u, seed ^ 5u, seed ^ 6u, seed ^ 7u};
    const pb_bf16x8 a = __builtin_bit_cast(pb_bf16x8, av), b = __builtin_bit_cast(pb_bf16x8, bv);
    // CHAIN consecutive MFMAs into the same accumulator before moving to the next one (1: every MFMA independent of its predecessor;
    // 3: the hi.lo / lo.hi / hi.hi triple of a split-bf16 product issued back to back)
    for (int it = 0; it < iters; it += CHAIN) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) sink[0] = t;                               // keeps the loop alive; never true
}
// The same with the B operands re-read from LDS at the convolution kernel's ratio: per unit of 12 MFMAs (four accumulators x the three
// split products) four ds_read_b128 of 64 consecutive 16-byte chunks (conflict-free); RPU = reads per unit (4 = the kernel's, 2, 1).
template <int RPU>
__global__ __launch_bounds__(512) void mfma_probe_lds_kernel(int iters, float* __restrict__ sink)
{
    typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float pb_f32x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char pl[];          // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) reinterpret_cast<u32x4*>(pl)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (unsigned)i};
    __syncthreads();
    pb_f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned seed = 0x3f803f80u + (unsigned)lane;
    u32x4 a0 = {seed, seed ^ 1u, seed ^ 2u, seed ^ 3u}, a1 = {seed ^ 4u, seed ^ 5u, seed ^ 6u, seed ^ 7u};
    u32x4 f[4] = {a0, a1, a0, a1};
    int off = lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < RPU; ++r) f[r] = *reinterpret_cast<const u32x4*>(pl + ((off + r * 1024) & 0xffff));
        off += 4096;
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, prod == 1 ? a1 : a0),
                                                                 __builtin_bit_cast(pb_bf16x8, f[(i + prod) & 3]), acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) sink[0] = t;
}
} // namespace

// LDS-fed probe: flops of one launch = blocks * 8 waves * iters * 12 MFMAs * 16384; rpu = ds_read_b128 per 12 MFMAs (1, 2 or 4)
extern "C" int gdm_mfma_probe_lds_hip(int blocks, int iters, int rpu, float* sink, void* stream)
{
    GDM_CHECK_ARG(blocks >= 1 && blocks <= 65535 && iters >= 1 && (rpu == 1 || rpu == 2 || rpu == 4) && sink, "gdm_mfma_probe_lds_hip: bad arguments");
    if (rpu == 4) hipLaunchKernelGGL(mfma_probe_lds_kernel<4>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, iters, sink);
    else if (rpu == 2) hipLaunchKernelGGL(mfma_probe_lds_kernel<2>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, iters, sink);
    else hipLaunchKernelGGL(mfma_probe_lds_kernel<1>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, iters, sink);
    return gdm_launch_status("mfma_probe_lds_kernel");
}

// flops of one launch = blocks * 8 waves * iters * 8 MFMAs * 16*16*32*2
extern "C" int gdm_mfma_probe_hip(int blocks, int iters, int chain, float* sink, void* stream)
{
    GDM_CHECK_ARG(blocks >= 1 && blocks <= 65535 && iters >= 3 && iters % 3 == 0 && (chain == 1 || chain == 3) && sink,
                  "gdm_mfma_probe_hip: bad arguments (iters a multiple of 3, chain 1 or 3)");
    if (chain == 3) hipLaunchKernelGGL(mfma_probe_kernel<3>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink);
    else hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, sink);
    return gdm_launch_status("mfma_probe_kernel");
}

static bool cin_ok(int Cin) { return Cin == 64 || (Cin >= 128 && Cin % 128 == 0); }

// 64-channel output tiles instead of 128-channel ones: when the 128-channel tiling gives fewer workgroups than the chip has CUs, or
// the layer has only 64 output channels
static bool narrow_tiles(unsigned pixel_tiles, int Cout)
{
    return Cout <= 64 || (long)pixel_tiles * ((Cout + 127) / 128) < 256;
}

extern "C" size_t gdm_conv3x3_act_bytes(int B, int Cin, int H, int W)
{
    if (B < 1 || !cin_ok(Cin) || H < 1 || W < 1) return 0;
    return (size_t)B * (H + 2) * (W + 2) * ((Cin + 127) / 128) * ROWB;
}

extern "C" size_t gdm_conv3x3_weight_bytes(int Cout, int Cin)
{
    if (Cout < 1 || !cin_ok(Cin)) return 0;
    return (size_t)9 * ((Cin + 127) / 128) * ((Cout + 127) & ~127) * ROWB;
}

extern "C" size_t gdm_conv1x1_weight_bytes(int Cout, int Cin)
{
    if (Cout < 1 || !cin_ok(Cin)) return 0;
    return (size_t)((Cin + 127) / 128) * ((Cout + 127) & ~127) * ROWB;
}

static int pack_weight(const float* w, int Cout, int Cin, int taps, void* wpk, void* stream, const char* who, int dgrad = 0)
{
    GDM_CHECK_ARG(w && wpk, "%s: NULL pointer", who);
    GDM_CHECK_ARG(Cout >= 1 && cin_ok(Cin), "%s: Cout=%d Cin=%d (Cin a multiple of 128, or 64)", who, Cout, Cin);
    const long items = (long)taps * ((Cin + 127) / 128) * ((Cout + 127) & ~127) * 16;
    hipLaunchKernelGGL(conv_pack_w_kernel, dim3(gdm_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, (unsigned char*)wpk, dgrad);
    return gdm_launch_status("conv_pack_w_kernel");
}

// The packed weights of the INPUT-GRADIENT convolution straight from the forward weight w f32[Cout, Cin, taps] (taps 9 or 1): the packed
// layer has Cin output channels and Cout input channels (Cout a multiple of 128, or 64); wpk: gdm_conv3x3_weight_bytes(Cin, Cout) /
// gdm_conv1x1_weight_bytes(Cin, Cout) bytes.
extern "C" int gdm_conv_pack_weight_dgrad_hip(const float* w, int Cout, int Cin, int taps, void* wpk, void* stream)
{
    GDM_CHECK_ARG(taps == 9 || taps == 1, "gdm_conv_pack_weight_dgrad_hip: taps=%d (9 or 1)", taps);
    return pack_weight(w, Cin, Cout, taps, wpk, stream, "gdm_conv_pack_weight_dgrad_hip", 1);
}

extern "C" int gdm_conv3x3_pack_weight_hip(const float* w, int Cout, int Cin, void* wpk, void* stream)
{
    return pack_weight(w, Cout, Cin, 9, wpk, stream, "gdm_conv3x3_pack_weight_hip");
}

extern "C" int gdm_conv1x1_pack_weight_hip(const float* w, int Cout, int Cin, void* wpk, void* stream)
{
    return pack_weight(w, Cout, Cin, 1, wpk, stream, "gdm_conv1x1_pack_weight_hip");
}

// xpk must be zero-filled by the caller (the border rows are never written).
extern "C" int gdm_conv3x3_pack_act_hip(const float* x, int B, int Cin, int H, int W, void* xpk, void* stream)
{
    GDM_CHECK_ARG(x && xpk, "gdm_conv3x3_pack_act_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && cin_ok(Cin) && H >= 1 && W >= 1, "gdm_conv3x3_pack_act_hip: bad shape (Cin=%d: a multiple of 128, or 64)", Cin);
    dim3 grid(gdm_cdiv((long)H * W, 64), (Cin + 127) / 128, B);
    hipLaunchKernelGGL(conv_pack_act_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, Cin, H, W, (unsigned char*)xpk);
    return gdm_launch_status("conv_pack_act_kernel");
}

// out (fp32 NCHW) and / or outpk (the packed operand of the next convolution over [B, Cout, H, W]; zero-filled once by the caller, only
// interior pixels are written) -- at least one of them.
static int conv3x3_launch(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                          int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* outpk, void* stream);

extern "C" int gdm_conv3x3_packed2_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                                       int B, int Cin, int Cout, int H, int W, int act, float* out, void* outpk, void* stream)
{
    return conv3x3_launch(xpk, wpk, scale, shift, res, B, Cin, Cout, H, W, 1, act, out, outpk, stream);
}

// H, W = OUTPUT size; xpk = the (H stride) x (W stride) input packed by gdm_conv3x3_pack_act_hip; stride 1 or 2
extern "C" int gdm_conv3x3_strided_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                                       int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* outpk, void* stream)
{
    return conv3x3_launch(xpk, wpk, scale, shift, res, B, Cin, Cout, H, W, stride, act, out, outpk, stream);
}

static int conv3x3_launch(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                          int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* outpk, void* stream)
{
    GDM_CHECK_ARG(stride == 1 || stride == 2, "gdm_conv3x3: stride=%d (1 or 2)", stride);
    GDM_CHECK_ARG(xpk && wpk && (out || outpk), "gdm_conv3x3_packed_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && cin_ok(Cin) && Cout >= 1, "gdm_conv3x3_packed_hip: Cin=%d Cout=%d (Cin a multiple of 128, or 64)", Cin, Cout);
    GDM_CHECK_ARG(W % 32 == 0 && H >= 1 && (H * W) % CONV_WPIX == 0,
                  "gdm_conv3x3_packed_hip: W=%d must be a multiple of 32 and H*W=%d of %d", W, H * W, CONV_WPIX);
    GDM_CHECK_ARG(act == 0 || act == 1, "gdm_conv3x3_packed_hip: act=%d", act);
    GDM_CHECK_ARG(!outpk || (Cout % 8 == 0 && ((long)B * H * W) % CV_PIX == 0),
                  "gdm_conv3x3_packed_hip: packed output needs Cout %% 8 == 0 and B*H*W %% 256 == 0 (got Cout=%d, B*H*W=%ld)", Cout, (long)B * H * W);
    const long ptot = (long)B * H * W;
    dim3 grid(gdm_cdiv(ptot, CV_PIX), gdm_cdiv(Cout, CV_CO));
    hipStream_t s = (hipStream_t)stream;
    constexpr int SMEM = CONV_SMEM;                                 // weight panels, or the waves' output tiles
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 9, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 9, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, true, 9, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, true, 9, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr = true;
    }
#define CV(A, R, NK) hipLaunchKernelGGL((CONV_KERNEL<A, R, 9, false, NK>), grid, dim3(CONV_THREADS), SMEM, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, res, B, Cin, Cout, H, W, out, (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)outpk, stride)
    // 64-channel tiles where 128-channel ones leave the chip half empty (or pad a 64-channel layer with zero rows): twice the workgroups
    if (narrow_tiles(grid.x, Cout)) {
        constexpr int SMEM4 = 2 * 64 * ROWB > CV_PIX * 68 * 4 ? 2 * 64 * ROWB : CV_PIX * 68 * 4;
        const dim3 grid4(grid.x, gdm_cdiv(Cout, 64));
        static bool attr4 = false;
#define CV4(A, R, NK) hipLaunchKernelGGL((CONV_KERNEL<A, R, 9, false, NK, 4>), grid4, dim3(CONV_THREADS), SMEM4, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, res, B, Cin, Cout, H, W, out, (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)outpk, stride)
#define AT4(A, R, NK) (void)hipFuncSetAttribute((const void*)CONV_KERNEL<A, R, 9, false, NK, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4)
        if (!attr4) {
            AT4(0, false, 4); AT4(1, false, 4); AT4(0, true, 4); AT4(1, true, 4);
            AT4(0, false, 8); AT4(1, false, 8); AT4(0, true, 8); AT4(1, true, 8);
            attr4 = true;
        }
        // still fewer workgroups than CUs (128 -> 128 at 32 x 32, batch 16: 64 x 2): 128-pixel workgroups of four waves
        if (Cin != 64 && (long)grid4.x * grid4.y < 256 && ptot % (CV_PIX / 2) == 0) {
            const dim3 grid4h(gdm_cdiv(ptot, CV_PIX / 2), grid4.y);
            static bool attr4h = false;
#define CV4H(A, R) hipLaunchKernelGGL((CONV_KERNEL<A, R, 9, false, 8, 4, MF_WAVES / 2>), grid4h, dim3(CONV_THREADS / 2), SMEM4, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, res, B, Cin, Cout, H, W, out, (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)outpk, stride)
#define AT4H(A, R) (void)hipFuncSetAttribute((const void*)CONV_KERNEL<A, R, 9, false, 8, 4, MF_WAVES / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4)
            if (!attr4h) {
                AT4H(0, false); AT4H(1, false); AT4H(0, true); AT4H(1, true);
                attr4h = true;
            }
            if (act == 0) { if (res) CV4H(0, true); else CV4H(0, false); }
            else { if (res) CV4H(1, true); else CV4H(1, false); }
#undef CV4H
#undef AT4H
            return gdm_launch_status("conv_mfma16_kernel (3x3, 64-channel tiles, 128 pixels)");
        }
        if (Cin == 64) {
            if (act == 0) { if (res) CV4(0, true, 4); else CV4(0, false, 4); }
            else { if (res) CV4(1, true, 4); else CV4(1, false, 4); }
        } else {
            if (act == 0) { if (res) CV4(0, true, 8); else CV4(0, false, 8); }
            else { if (res) CV4(1, true, 8); else CV4(1, false, 8); }
        }
#undef CV4
#undef AT4
        return gdm_launch_status("conv_mfma16_kernel (3x3, 64-channel tiles)");
    }
    if (Cin == 64) {                                             // one half-filled chunk: only its four non-zero k-steps are run
        if (act == 0) { if (res) CV(0, true, 4); else CV(0, false, 4); }
        else { if (res) CV(1, true, 4); else CV(1, false, 4); }
    } else {
        if (act == 0) { if (res) CV(0, true, 8); else CV(0, false, 8); }
        else { if (res) CV(1, true, 8); else CV(1, false, 8); }
    }
#undef CV
    return gdm_launch_status("conv_mfma16_kernel (3x3)");
}

extern "C" int gdm_conv3x3_packed_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                                      int B, int Cin, int Cout, int H, int W, int act, float* out, void* stream)
{
    return gdm_conv3x3_packed2_hip(xpk, wpk, scale, shift, res, B, Cin, Cout, H, W, act, out, nullptr, stream);
}

// Grouped, gathered GEMM on the same kernel: Y[r, 0:128] = Wpk[tile_co0[r / 256] + 0:128, :] . X[rowidx[r], :] for R rows (R % 256 == 0),
// X packed by gdm_conv3x3_pack_act_hip(x, 1, Cin, 1, M), weights packed by gdm_conv1x1_pack_weight_hip(Cout_total, Cin).
extern "C" int gdm_gemm_grouped_hip(const void* xpk, const void* wpk, const int32_t* rowidx, const int32_t* tile_co0, int R, int M,
                                    int Cin, int Cout_total, float* out, void* stream)
{
    GDM_CHECK_ARG(xpk && wpk && rowidx && tile_co0 && out, "gdm_gemm_grouped_hip: NULL pointer");
    GDM_CHECK_ARG(R >= 256 && R % 256 == 0 && M >= 1 && Cin >= 128 && Cin % 128 == 0 && Cout_total >= 128 && Cout_total % 128 == 0,
                  "gdm_gemm_grouped_hip: R=%d (multiple of 256) M=%d Cin=%d Cout_total=%d (multiples of 128)", R, M, Cin, Cout_total);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        attr = true;
    }
    // the kernel sees B = R "pixels" of a 1 x 1 map for the row bookkeeping and H = 1, W = M for the packed source
    hipLaunchKernelGGL((CONV_KERNEL<0, false, 1, true>), dim3(R / CV_PIX, 1), dim3(CONV_THREADS), 2 * CV_PANEL, (hipStream_t)stream,
                       (const unsigned char*)xpk, (const unsigned char*)wpk, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       R, Cin, Cout_total, 1, M, out, rowidx, tile_co0);
    return gdm_launch_status("gemm_grouped_kernel");
}

// 1x1 convolution / GEMM on the same kernel (one tap): out = act(scale * (W x) + shift), x packed by gdm_conv3x3_pack_act_hip.
// pixel_major != 0 writes out[B*H*W, Cout] (row per pixel) instead of NCHW.
static int conv1x1_launch(const void* xpk, const void* wpk, const float* scale, const float* shift,
                          int B, int Cin, int Cout, int H, int W, int stride, int act, int pixel_major, float* out, void* stream);

extern "C" int gdm_conv1x1_packed_hip(const void* xpk, const void* wpk, const float* scale, const float* shift,
                                      int B, int Cin, int Cout, int H, int W, int act, int pixel_major, float* out, void* stream)
{
    return conv1x1_launch(xpk, wpk, scale, shift, B, Cin, Cout, H, W, 1, act, pixel_major, out, stream);
}

// the same GEMM with one weight set PER IMAGE (include/gdm.h: the split-K parts of the weight-gradient GEMM): H*W % 256 == 0
extern "C" int gdm_conv1x1_packed_wb_hip(const void* xpk, const void* wpk, long w_bstride, int B, int Cin, int Cout, int H, int W, float* out,
                                         void* stream)
{
    GDM_CHECK_ARG(xpk && wpk && out, "gdm_conv1x1_packed_wb_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && Cin >= 128 && Cin % 128 == 0 && Cout >= 1, "gdm_conv1x1_packed_wb_hip: Cin=%d Cout=%d (Cin a multiple of 128)", Cin, Cout);
    GDM_CHECK_ARG(W % 32 == 0 && H >= 1 && (H * W) % CV_PIX == 0, "gdm_conv1x1_packed_wb_hip: W=%d must be a multiple of 32 and H*W=%d of %d",
                  W, H * W, CV_PIX);
    GDM_CHECK_ARG(w_bstride >= 0 && w_bstride % 16 == 0, "gdm_conv1x1_packed_wb_hip: w_bstride=%ld", w_bstride);
    const long ptot = (long)B * H * W;
    dim3 grid(gdm_cdiv(ptot, CV_PIX), gdm_cdiv(Cout, CV_CO));
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        attr = true;
    }
    hipLaunchKernelGGL((CONV_KERNEL<0, false, 1, false>), grid, dim3(CONV_THREADS), 2 * CV_PANEL, (hipStream_t)stream, (const unsigned char*)xpk,
                       (const unsigned char*)wpk, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, B, Cin, Cout, H, W, out,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)nullptr, 1, w_bstride);
    return gdm_launch_status("conv1x1_bf16x3_kernel (per-image weights)");
}

// H, W = OUTPUT size; xpk = the (H stride) x (W stride) input (the downsample branch of a strided residual block reads every other pixel)
extern "C" int gdm_conv1x1_strided_hip(const void* xpk, const void* wpk, const float* scale, const float* shift,
                                       int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* stream)
{
    return conv1x1_launch(xpk, wpk, scale, shift, B, Cin, Cout, H, W, stride, act, 0, out, stream);
}

static int conv1x1_launch(const void* xpk, const void* wpk, const float* scale, const float* shift,
                          int B, int Cin, int Cout, int H, int W, int stride, int act, int pixel_major, float* out, void* stream)
{
    GDM_CHECK_ARG(stride == 1 || stride == 2, "gdm_conv1x1: stride=%d (1 or 2)", stride);
    GDM_CHECK_ARG(xpk && wpk && out, "gdm_conv1x1_packed_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && cin_ok(Cin) && Cout >= 1, "gdm_conv1x1_packed_hip: Cin=%d Cout=%d (Cin a multiple of 128, or 64)", Cin, Cout);
    GDM_CHECK_ARG(Cin != 64 || !pixel_major, "gdm_conv1x1_packed_hip: Cin=64 is built for the NCHW output only");
    GDM_CHECK_ARG(W % 32 == 0 && H >= 1 && (H * W) % CONV_WPIX == 0,
                  "gdm_conv1x1_packed_hip: W=%d must be a multiple of 32 and H*W=%d of %d", W, H * W, CONV_WPIX);
    GDM_CHECK_ARG(act == 0 || act == 1, "gdm_conv1x1_packed_hip: act=%d", act);
    const long ptot = (long)B * H * W;
    const unsigned ptiles = gdm_cdiv(ptot, CV_PIX);
    // 144-channel tiles (nine 16-channel blocks) where they divide Cout and fill the chip's rounds better than 128-channel ones: the
    // tap GEMMs of PSPUpsample have 9 * Cout' outputs -- 2304 at 64 pixel tiles = 1152 tiles of 128 (4.5 rounds of 256 CUs, paid as
    // 5) or 1024 tiles of 144 (4 rounds); 576 at 256 pixel tiles = 1280 tiles of 128, the last of every five half empty, or 1024 of 144
    if (Cin != 64 && !pixel_major && stride == 1 && Cout % 144 == 0 && GDM_CONV_GLDS) {
        const long r128 = gdm_cdiv((long)ptiles * gdm_cdiv(Cout, 128), 256) * 128, r144 = gdm_cdiv((long)ptiles * (Cout / 144), 256) * 144;
        if (r144 < r128) {
            constexpr int PANEL9 = 9 * 16 * ROWB, TILE9 = CV_PIX * (9 * 16 + 4) * 4;
            constexpr int SMEM9 = 2 * PANEL9 > TILE9 ? 2 * PANEL9 : TILE9;
            static bool attr9 = false;
            if (!attr9) {
                (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, false, 8, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM9);
                (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 1, false, 8, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM9);
                attr9 = true;
            }
            const dim3 grid9(ptiles, Cout / 144);
#define C19(A) hipLaunchKernelGGL((CONV_KERNEL<A, false, 1, false, 8, 9>), grid9, dim3(CONV_THREADS), SMEM9, (hipStream_t)stream, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, (const float*)nullptr, B, Cin, Cout, H, W, out, (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)nullptr, stride)
            if (act == 0) C19(0); else C19(1);
#undef C19
            return gdm_launch_status("conv1x1_bf16x3_kernel (144-channel tiles)");
        }
    }
    const bool narrow = Cin != 64 && !pixel_major && narrow_tiles(ptiles, Cout);
    const unsigned ctiles = gdm_cdiv(Cout, narrow ? 64 : CV_CO);    // the last block's rows beyond Cout are zero weights, never stored
    // pixel tiles are the fast grid axis: workgroups that share a pixel tile land on one XCD.  (Channel tiles fastest was measured on the
    // 1024 -> 2304 tap GEMM in round 3: same time, 903 MB instead of 573 MB of fabric traffic -- each XCD then streams every weight panel.)
    const dim3 grid(ptiles, ctiles);
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 1, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CV_PANEL);
        attr = true;
    }
    if (Cin == 64) {                                            // one half-filled chunk: only its four non-zero k-steps are run
#define C1TAIL , (const int32_t*)nullptr, (const int32_t*)nullptr, (unsigned char*)nullptr, stride
#define C1H(A) hipLaunchKernelGGL((CONV_KERNEL<A, false, 1, false, 4>), grid, dim3(CONV_THREADS), 2 * CV_PANEL, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, (const float*)nullptr, B, Cin, Cout, H, W, out C1TAIL)
        if (act == 0) C1H(0); else C1H(1);
#undef C1H
        return gdm_launch_status("conv1x1_bf16x3_kernel");
    }
    if (narrow) {
        const dim3 grid4 = grid;
        static bool attr4 = false;
        if (!attr4) {
            (void)hipFuncSetAttribute((const void*)CONV_KERNEL<0, false, 1, false, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * ROWB);
            (void)hipFuncSetAttribute((const void*)CONV_KERNEL<1, false, 1, false, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * ROWB);
            attr4 = true;
        }
#define C14(A) hipLaunchKernelGGL((CONV_KERNEL<A, false, 1, false, 8, 4>), grid4, dim3(CONV_THREADS), 2 * 64 * ROWB, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, (const float*)nullptr, B, Cin, Cout, H, W, out C1TAIL)
        if (act == 0) C14(0); else C14(1);
#undef C14
        return gdm_launch_status("conv1x1_bf16x3_kernel (64-channel tiles)");
    }
#define C1(A, P) hipLaunchKernelGGL((CONV_KERNEL<A, false, 1, P>), grid, dim3(CONV_THREADS), 2 * CV_PANEL, s, (const unsigned char*)xpk, (const unsigned char*)wpk, scale, shift, (const float*)nullptr, B, Cin, Cout, H, W, out C1TAIL)
    if (act == 0) { if (pixel_major) C1(0, true); else C1(0, false); }
    else { if (pixel_major) C1(1, true); else C1(1, false); }
#undef C1
    return gdm_launch_status("conv1x1_bf16x3_kernel");
}
