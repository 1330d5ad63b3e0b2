// Per-point (1x1) layers of the point branch and of the RGB <-> point fusion, gfx950 (inference).
//
// Replaces, per layer, the chain  torch.cat -> 1x1 Conv1d/Conv2d (a library GEMM) -> BatchNorm -> activation (-> residual add)  of
//   /root/reference/models/pytorch_utils.py:70-124 and models/RandLA/pytorch_utils.py:34-99 (`_ConvBase`, eval mode),
//   Dilated_res_block's tail  lrelu(mlp2(f) + shortcut(x))            /root/reference/models/RandLA/RandLANet.py:685-688,
//   the fusion layers over cat(point features, pooled pixel features)   /root/reference/models/ffb6d.py:224-231,259-265,
//   the decoder layers over cat(skip, nearest_interpolation(deeper))    /root/reference/models/ffb6d.py:246-250,268-272
// by ONE launch:   out[b,:,i] = act( scale * (W . [x0 ; x1 ; x2][b,:,i]) + shift )
// where every input segment is a channel-major tensor f32[B,C,n_src] read either in place (n_src == n) or through a per-point
// index idx[b,i] (the nearest-neighbour interpolation of ffb6d.py:148-163 folded into the load).  The concat is never formed: the
// K loop walks the segments in order.  The residual tail is the same layer over [f ; x] with the two folded BatchNorm scales
// multiplied into the two weight blocks and the shifts added (done once, on the host side of the module cache).
//
// These layers are tiny (8..1024 channels at 128..32768 points: < 2 GFLOP for all ~40 of them per step) and the step is bound by
// their launch count and their latency, not their arithmetic: fp32 FMAs (exact fp32 products, ascending-k summation per partial).
// A wave owns a 64-point x 16-channel tile (lane = 4 points x 4 channels); a workgroup is NW such waves arranged as NW/KS channel
// groups x KS parts of the K axis: (NW, KS) = (4, 1) where the points alone fill the chip, (16, 4) and (16, 8) for the deep levels
// (128..2048 points, K up to 1024), whose only parallelism is K: sixteen waves per CU hide each other's latencies, and the KS
// partial sums are added in fixed order through LDS.  Operands are staged through LDS in 16-deep K chunks, coalesced along the
// points, with the next chunk's global loads in flight during the FMAs.
#include "gdm_common.h"
#include <stdlib.h>

namespace {

constexpr int PT = 64, XS = 68;       // XS: padded row of the x tile (keeps float4 reads aligned, spreads banks)
constexpr int MAXSEG = 4;              // the MFMA form walks up to four segments, the FMA form three

struct PwSegDev {
    const float* x;
    const int32_t* idx;
    int C, n_src;
};

struct PwArgs {
    PwSegDev seg[MAXSEG];
    int nseg;
    const float* wt;          // [K][Cout] (wks = Cout, wcs = 1), K = sum of the segments' channels, rows in segment order; or the layer's
    long wks, wcs;            // weight as the module holds it, [Cout][K] (wks = 1, wcs = K: gdm_pointwise2_hip, the training path)
    const float* scale;       // [Cout] or NULL (1)
    const float* shift;       // [Cout] or NULL (0)
    float* out;
    int n, Cout, outC, out_c0, point_major, act, K;
    float slope;
    long total;               // B * n
};

// address of point (crop b, column i; flat index g) in a channel-major segment: scalars in, so that nothing takes the address of
// the kernel-argument block (which would copy it to scratch memory)
__device__ __forceinline__ const float* seg_column(const float* x, const int32_t* idx, int C, int n_src, long b, int i, long g)
{
    int col = i;
    if (idx) col = min(max(idx[g], 0), n_src - 1);
    return x + b * (long)C * n_src + col;
}

template <int NW, int KS, bool VEC>
__global__ __launch_bounds__(NW * 64) void pointwise_kernel(const PwArgs a)
{
    constexpr int NCG = NW / KS;                 // channel groups (of 16) per workgroup
    constexpr int KT = 16;                       // K rows per chunk and K part
    constexpr int XL = KT / NCG;                 // x rows each wave loads per chunk (the NCG waves of a K part share its x tile)
    constexpr int WL = KT / 4;                   // w rows (of 16 channels) each lane loads per chunk
    static_assert(NCG >= 1 && NCG <= KT && KT % NCG == 0, "tile arrangement");
    __shared__ __attribute__((aligned(16))) float xs[KS][KT][XS];
    __shared__ __attribute__((aligned(16))) float ws[NW][KT][16];
    __shared__ __attribute__((aligned(16))) float red[KS > 1 ? NW : 1][16][64];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int cg = wave / KS, kp = wave % KS;
    const int tx = lane & 15, ty = lane >> 4;
    const long g0 = (long)blockIdx.x * PT;
    const int c0 = ((int)blockIdx.y * NCG + cg) * 16;
    const int n = a.n, Cout = a.Cout, K = a.K;
    const int kpart = (((K + KS - 1) / KS + KT - 1) / KT) * KT;
    const int kbeg = kp * kpart;
    const int nchunk = kpart / KT;

    // this lane's point column in every segment.  Every address formed below is a valid one (points past the end read point 0,
    // rows past K read row K-1 and are zeroed by a select on a wave-uniform condition, channels past Cout read channel Cout-1):
    // the loads carry no branch, so a chunk's loads are all in flight together; what the out-of-range lanes compute is never stored
    const float *xb0, *xb1, *xb2;                 // scalars, not an array: a dynamically indexed array would live in scratch memory
    long xs0, xs1, xs2;
    int ce0, ce1;
    {
        // scalar path: lane = one point of the tile, rows uniform over the wave.  VEC path (no index, n % 4 == 0, Cout % 4 == 0):
        // lane = four consecutive points (lane & 15) of row (lane >> 4): one 16-byte load per lane fetches four rows per wave
        // instruction -- a wave-wide dword load costs the texture path as much as a dwordx4 one, and it is that path, not the FMAs,
        // that bounds the deep layers
        const long g = min(g0 + (VEC ? (lane & 15) * 4 : lane), a.total - (VEC ? 4 : 1));
        const long b = g / n;
        const int i = (int)(g - b * n);
        xb0 = seg_column(a.seg[0].x, a.seg[0].idx, a.seg[0].C, a.seg[0].n_src, b, i, g);
        xs0 = a.seg[0].n_src;
        xb1 = xb2 = xb0;                          // absent segments alias segment 0 (never selected: their range of k is empty)
        xs1 = xs2 = xs0;
        ce0 = a.seg[0].C;
        ce1 = ce0;
        if (a.nseg > 1) {
            xb1 = seg_column(a.seg[1].x, a.seg[1].idx, a.seg[1].C, a.seg[1].n_src, b, i, g);
            xs1 = a.seg[1].n_src;
            ce1 = ce0 + a.seg[1].C;
        }
        if (a.nseg > 2) {
            xb2 = seg_column(a.seg[2].x, a.seg[2].idx, a.seg[2].C, a.seg[2].n_src, b, i, g);
            xs2 = a.seg[2].n_src;
        }
    }
    const float* const wtp = a.wt;
    // segment of row k by ARITHMETIC on 0/1 masks, not by selects: the compiler turns a chain of selects over the three
    // (pointer, stride) pairs into a table in scratch memory indexed per load
    const long e1 = (const char*)xb1 - (const char*)xb0, e2 = (const char*)xb2 - (const char*)xb1;
    const long sd1 = xs1 - xs0, sd2 = xs2 - xs1;
    const int cd1 = ce0, cd2 = ce1 - ce0;
    auto x_addr = [&](int k) -> const float* {   // address of row min(k, K-1) in this lane's column (RAW: rows past K are zeroed where
        const int kc = min(k, K - 1);             // the value is stored to LDS -- a select here would sit right behind the load and
        const long t1 = -(long)(kc >= ce0), t2 = -(long)(kc >= ce1);              // make the prefetch wait); t = 0 or all ones
        const int kl = kc - (cd1 & (int)t1) - (cd2 & (int)t2);
        const long str = xs0 + (sd1 & t1) + (sd2 & t2);
        const float* base = (const float*)((const char*)xb0 + (e1 & t1) + (e2 & t2));
        return base + (long)kl * str;
    };
    // x rows of this wave in a chunk: scalar XL rows cg + NCG r (uniform); VEC XL / 4 loads of rows cg XL + 4 j + (lane >> 4)
    // w tile 16 x 16 of this wave: scalar rows (lane >> 4) + 4 r, channel lane & 15; VEC row lane >> 2, channels 4 (lane & 3) ..
    constexpr int XV = VEC ? XL / 4 : XL;
    constexpr int WV = VEC ? KT / 16 : WL;
    static_assert(!VEC || (XL % 4 == 0 && KT % 16 == 0), "vector loads cover 4 x rows / 16 w rows per instruction");
    const int wc = VEC ? min(c0 + (lane & 3) * 4, Cout - 4) : min(c0 + (lane & 15), Cout - 1);
    auto x_row = [&](int r) -> int { return VEC ? cg * XL + 4 * r + (lane >> 4) : cg + NCG * r; };
    auto w_row = [&](int r) -> int { return VEC ? 16 * r + (lane >> 2) : (lane >> 4) + 4 * r; };

    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;

    float4 xr[XV], wr[WV];                        // scalar path: .x only
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < XV; ++r) {
            const float* p = x_addr(k0 + x_row(r));
            if (VEC) xr[r] = *reinterpret_cast<const float4*>(p);
            else xr[r].x = *p;
        }
#pragma unroll
        for (int r = 0; r < WV; ++r) {
            const float* p = wtp + (long)min(k0 + w_row(r), K - 1) * a.wks + (long)wc * a.wcs;
            if (VEC) wr[r] = *reinterpret_cast<const float4*>(p);
            else wr[r].x = *p;
        }
    };
    fetch(kbeg);
    for (int ch = 0; ch < nchunk; ++ch) {
        const int kc0 = kbeg + ch * KT;
#pragma unroll
        for (int r = 0; r < XV; ++r) {
            const bool in = kc0 + x_row(r) < K;
            if (VEC) *reinterpret_cast<float4*>(&xs[kp][x_row(r)][(lane & 15) * 4]) = in ? xr[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            else xs[kp][x_row(r)][lane] = in ? xr[r].x : 0.f;
        }
#pragma unroll
        for (int r = 0; r < WV; ++r) {
            const bool in = kc0 + w_row(r) < K;
            if (VEC) *reinterpret_cast<float4*>(&ws[wave][w_row(r)][(lane & 3) * 4]) = in ? wr[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            else ws[wave][w_row(r)][lane & 15] = in ? wr[r].x : 0.f;
        }
        __syncthreads();
        if (ch + 1 < nchunk) fetch(kbeg + (ch + 1) * KT);   // next chunk's loads fly during the FMAs below
        float4 xv[2][4], wv[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[0][u] = *reinterpret_cast<const float4*>(&xs[kp][u][tx * 4]);
            wv[0][u] = *reinterpret_cast<const float4*>(&ws[wave][u][ty * 4]);
        }
#pragma unroll
        for (int k4 = 0; k4 < KT / 4; ++k4) {
            const int cur = k4 & 1, nxt = cur ^ 1;
            if (k4 + 1 < KT / 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[nxt][u] = *reinterpret_cast<const float4*>(&xs[kp][(k4 + 1) * 4 + u][tx * 4]);
                    wv[nxt][u] = *reinterpret_cast<const float4*>(&ws[wave][(k4 + 1) * 4 + u][ty * 4]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 x4 = xv[cur][u], w4 = wv[cur][u];
                acc[0][0] = fmaf(w4.x, x4.x, acc[0][0]); acc[0][1] = fmaf(w4.x, x4.y, acc[0][1]);
                acc[0][2] = fmaf(w4.x, x4.z, acc[0][2]); acc[0][3] = fmaf(w4.x, x4.w, acc[0][3]);
                acc[1][0] = fmaf(w4.y, x4.x, acc[1][0]); acc[1][1] = fmaf(w4.y, x4.y, acc[1][1]);
                acc[1][2] = fmaf(w4.y, x4.z, acc[1][2]); acc[1][3] = fmaf(w4.y, x4.w, acc[1][3]);
                acc[2][0] = fmaf(w4.z, x4.x, acc[2][0]); acc[2][1] = fmaf(w4.z, x4.y, acc[2][1]);
                acc[2][2] = fmaf(w4.z, x4.z, acc[2][2]); acc[2][3] = fmaf(w4.z, x4.w, acc[2][3]);
                acc[3][0] = fmaf(w4.w, x4.x, acc[3][0]); acc[3][1] = fmaf(w4.w, x4.y, acc[3][1]);
                acc[3][2] = fmaf(w4.w, x4.z, acc[3][2]); acc[3][3] = fmaf(w4.w, x4.w, acc[3][3]);
            }
        }
        __syncthreads();
    }
    if (KS > 1) {                                  // partial sums of the K parts, added in the fixed order 0, 1, .., KS-1
        if (kp != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave][j * 4 + q][lane] = acc[j][q];
        }
        __syncthreads();
        if (kp != 0) return;
        for (int p = 1; p < KS; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] += red[cg * KS + p][j * 4 + q][lane];
    }

    const long p0 = g0 + tx * 4;
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty * 4 + j;
        const bool live = c < Cout;
        const float sc = (live && a.scale) ? a.scale[c] : 1.f;
        const float sh = (live && a.shift) ? a.shift[c] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = fmaf(acc[j][q], sc, sh);
            if (a.act == 1) y = fmaxf(y, 0.f);
            else if (a.act == 2) y = y > 0.f ? y : y * a.slope;
            v[j][q] = y;
        }
    }
    if (a.point_major) {
        // out[p][out_c0 + c]: the lane's four channels are contiguous
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

// ---------------------------------------------------------------------------------------------------------------------------
// The same layer on v_mfma_f32_16x16x4_f32 (exact fp32 products: the f32-input MFMA is an fmaf chain, MI355X_MICROARCH.md), for the
// layers the FMA kernel runs at one or two waves per SIMD: there the 16 FMAs + 2 LDS reads + address arithmetic per K row of a wave
// are bound by vector-instruction ISSUE, while one MFMA retires 16 x 16 x 4 products per instruction and takes its operands straight
// from global memory in fragment order -- no LDS staging, no barrier in the K loop:
//   A fragment (x):  lane l = point p0 + 16 g + (l & 15), row k0 + (l >> 4)      one dword per lane, 4 rows x 64 B per instruction
//   B fragment (W):  lane l = channel c0 + (l & 15),      row k0 + (l >> 4)      one dword per lane (wt is [K][Cout])
//   accumulator g:   lane l = channel c0 + (l & 15), points p0 + 16 g + 4 (l >> 4) + 0..3   = one 16-byte store channel-major
// A workgroup = KS waves = the KS parts of the K axis of ONE tile of 64 points x 16 channels; partial sums are added through LDS in
// fixed order.  Each wave walks its K range segment by segment (pointer increments only), four K rows per MFMA step.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float pw_f32x4;

// VEC (no indexed segment, n % 4 == 0, 16-byte aligned operands): accumulator block g owns the points p0 + 4 (l & 15) + g instead of
// p0 + 16 g + (l & 15), so ONE 16-byte load per lane fetches a K row's values for all four blocks (a wave-wide dword load costs the
// texture path what a dwordx4 one does: 2 load instructions per step instead of 5), and the four blocks' results for one (row group,
// r) are four consecutive points = one 16-byte store
// one tile of 64 points x 16 channels (points from p0, channels from c0) by the KS waves of the workgroup; `red`: [KS][16][64] floats of LDS
// CB: 16-channel blocks per wave (KS == 1 only).  With CB = 4 a wave forms 64 points x 64 channels from ONE pass over its x rows (the
// training-size layers, ~100 k ... 1.5 M points: with one block per wave every x row was fetched Cout / 16 times); each accumulator
// still sums its K rows in the same order, so the results do not depend on CB.
template <int KS, bool VEC, int CB = 1>
__device__ __forceinline__ void pw_mfma_tile(const PwArgs& a, const long p0, const int c0, float (*red)[16][64])
{
    static_assert(CB == 1 || KS == 1, "several channel blocks per wave only without a K split");
    const int tid = threadIdx.x, wave = KS > 1 ? tid >> 6 : 0, lane = tid & 63;       // wave = this wave's part of the K axis
    const int l16 = lane & 15, kq = lane >> 4;
    const int n = a.n, Cout = a.Cout, K = a.K;
    const int kpart = (((K + KS - 1) / KS + 3) / 4) * 4;
    const int kbeg = wave * kpart, kend = min(K, kbeg + kpart);

    // this lane's four points (one per accumulator block)
    long pb[4];
    int pi[4];
    long pg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // out-of-range points are clamped to valid ones (their results are never stored); VEC lanes load four consecutive points, so
        // the clamp keeps the whole group inside the array (total % 4 == 0 there)
        const long p = VEC ? min(p0 + 4 * l16, a.total - 4) + g : min(p0 + 16 * g + l16, a.total - 1);
        pg[g] = p;
        pb[g] = p / n;
        pi[g] = (int)(p - pb[g] * n);
    }
    long wcoff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wcoff[cb] = (long)min(c0 + 16 * cb + l16, Cout - 1) * a.wcs;

    pw_f32x4 acc[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[cb][g][r] = 0.f;

    int seg_start = 0;
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
        if (s >= a.nseg) break;
        const float* sx = a.seg[s].x;
        const int32_t* sidx = a.seg[s].idx;
        const int sC = a.seg[s].C, sn = a.seg[s].n_src;
        const int seg_end = seg_start + sC;
        const int k0 = max(kbeg, seg_start), k1 = min(kend, seg_end);       // this wave's rows inside this segment
        if (k0 < k1) {
            const float* xp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int col = pi[g];
                if (sidx) col = min(max(sidx[pg[g]], 0), sn - 1);
                xp[g] = sx + (pb[g] * sC + (k0 - seg_start + kq)) * (long)sn + col;
            }
            const float* wp = a.wt + (long)(k0 + kq) * a.wks;
            const long xinc = 4L * sn, winc = 4L * a.wks;
            const int nfull = (k1 - k0) / 4;                               // steps whose four rows all lie inside [k0, k1)
            // software pipeline in groups of four steps: group i + 1's 20 loads are issued BEFORE group i's 16 MFMAs (two register
            // sets, the loop unrolled by two so that both are statically indexed)
            const int ngrp = nfull / 4;
            float xa[4][4], wa[4][CB], xb[4][4], wb[4][CB];
            auto load_grp = [&](float (&xv)[4][4], float (&wv)[4][CB]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) wv[u][cb] = wp[wcoff[cb]];
                    wp += winc;
                    if (VEC) {
                        const float4 t = *reinterpret_cast<const float4*>(xp[0]);
                        xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                        xp[0] += xinc;
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            xv[u][g] = xp[g][0];
                            xp[g] += xinc;
                        }
                    }
                }
            };
            auto mfma_grp = [&](const float (&xv)[4][4], const float (&wv)[4][CB]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][g], wv[u][cb], acc[cb][g], 0, 0, 0);
            };
            int st = 4 * ngrp;
            if (ngrp > 0) {
                load_grp(xa, wa);
                int gi = 0;
                for (; gi + 2 < ngrp; gi += 2) {
                    load_grp(xb, wb);
                    mfma_grp(xa, wa);
                    load_grp(xa, wa);
                    mfma_grp(xb, wb);
                }
                if (gi + 1 < ngrp) {                                        // two groups left: a in flight, then b
                    load_grp(xb, wb);
                    mfma_grp(xa, wa);
                    mfma_grp(xb, wb);
                } else {
                    mfma_grp(xa, wa);
                }
            }
            for (; st < nfull; ++st) {
                float wv[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) wv[cb] = wp[wcoff[cb]];
                wp += winc;
                float xv[4];
                if (VEC) {
                    const float4 t = *reinterpret_cast<const float4*>(xp[0]);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                    xp[0] += xinc;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        xv[g] = xp[g][0];
                        xp[g] += xinc;
                    }
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[g], wv[cb], acc[cb][g], 0, 0, 0);
            }
            if (k0 + 4 * nfull < k1) {                                      // the last, partial step: rows past k1 contribute zero
                const bool in = k0 + 4 * nfull + kq < k1;
                const long back = in ? 0 : (long)(k0 + 4 * nfull + kq - (k1 - 1));      // out-of-range lanes re-read row k1 - 1
                float wv[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const float wraw = wp[wcoff[cb] - back * a.wks];
                    wv[cb] = in ? wraw : 0.f;
                }
                float xraw[4];
                if (VEC) {
                    const float4 t = *reinterpret_cast<const float4*>(xp[0] - back * sn);
                    xraw[0] = t.x; xraw[1] = t.y; xraw[2] = t.z; xraw[3] = t.w;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xraw[g] = xp[g][-back * sn];
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[cb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? xraw[g] : 0.f, wv[cb], acc[cb][g], 0, 0, 0);
            }
        }
        seg_start = seg_end;
    }

    if (KS > 1) {
        if (wave != 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][lane] = acc[0][g][r];
        }
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < KS; ++w)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[0][g][r] += red[w][g * 4 + r][lane];
    }

#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
    const int co = c0 + 16 * cb + l16;
    if (co >= Cout) continue;
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.shift ? a.shift[co] : 0.f;
    auto finish = [&](float y) {
        y = fmaf(y, sc, sh);
        if (a.act == 1) y = fmaxf(y, 0.f);
        else if (a.act == 2) y = y > 0.f ? y : y * a.slope;
        return y;
    };
    // four consecutive points per (j, i): VEC -- block g = point 4 (4 kq + r) + g, so (j, i) = (r, g); else -- point 16 g + 4 kq + r, (j, i) = (g, r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = finish(VEC ? acc[cb][i][j] : acc[cb][j][i]);
        const long q0 = p0 + (VEC ? 16 * kq + 4 * j : 16 * j + 4 * kq);   // first of the four consecutive points
        if (a.point_major) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (q0 + r < a.total) a.out[(q0 + r) * a.outC + a.out_c0 + co] = v[r];
        } else if (n % 4 == 0 && q0 + 3 < a.total) {                       // four points of one crop, 16-byte aligned
            const long b = q0 / n;
            *reinterpret_cast<float4*>(a.out + (b * a.outC + a.out_c0 + co) * n + (q0 - b * n)) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long q = q0 + r;
                if (q >= a.total) break;
                const long b = q / n;
                a.out[(b * a.outC + a.out_c0 + co) * n + (q - b * n)] = v[r];
            }
        }
    }
    }
}

template <int KS, bool VEC, int CB = 1>
__global__ __launch_bounds__(KS * 64) void pointwise_mfma_kernel(const PwArgs a)
{
    __shared__ __attribute__((aligned(16))) float red[KS > 1 ? KS : 1][16][64];
    pw_mfma_tile<KS, VEC, CB>(a, (long)blockIdx.x * PT, (int)blockIdx.y * (16 * CB), red);
}

// No K split, the workgroup's FOUR waves = four channel groups (of 16 CB channels) of ONE 64-point tile: the training-size layers
// (~100 k ... 1.5 M points per launch, 32 ... 256 channels).  With one-wave workgroups the channel groups of a tile were separate
// workgroups far apart in dispatch order and every x row came from HBM Cout / 16 times; here the four waves ask for the same x rows at
// about the same time, through the same L1.  Same sums in the same order as every other form.
template <bool VEC, int CB>
__global__ __launch_bounds__(256) void pointwise_mfma_cw_kernel(const PwArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int c0 = ((int)blockIdx.y * 4 + wave) * (16 * CB);
    if (c0 >= a.Cout) return;
    pw_mfma_tile<1, VEC, CB>(a, (long)blockIdx.x * PT, c0, nullptr);
}

// Up to four INDEPENDENT layers of equal K and Cout in one launch (the four prior products M_k . pool_k(f) of the pyramid-pooling
// module, pspnet.py:17-31: 16 ... 576 points each, every one a latency-bound launch of its own): the tiles of job j are the blocks
// [e_{j-1}, e_j) of the grid's x axis; each job runs exactly the tile function of pointwise_mfma_kernel with the same K split, so
// the results are bit-identical to the separate launches.  The four argument blocks are four kernel parameters (statically
// addressed: indexing an array of them by blockIdx would copy the block to scratch memory).
template <int KS>
__global__ __launch_bounds__(KS * 64) void pointwise_mfma_jobs_kernel(const PwArgs a0, const PwArgs a1, const PwArgs a2, const PwArgs a3,
                                                                      int e0, int e1, int e2, int vecmask)
{
    __shared__ __attribute__((aligned(16))) float red[KS > 1 ? KS : 1][16][64];
    const int bx = (int)blockIdx.x, c0 = (int)blockIdx.y * 16;
    if (bx < e0) {
        if (vecmask & 1) pw_mfma_tile<KS, true>(a0, (long)bx * PT, c0, red);
        else pw_mfma_tile<KS, false>(a0, (long)bx * PT, c0, red);
    } else if (bx < e1) {
        if (vecmask & 2) pw_mfma_tile<KS, true>(a1, (long)(bx - e0) * PT, c0, red);
        else pw_mfma_tile<KS, false>(a1, (long)(bx - e0) * PT, c0, red);
    } else if (bx < e2) {
        if (vecmask & 4) pw_mfma_tile<KS, true>(a2, (long)(bx - e1) * PT, c0, red);
        else pw_mfma_tile<KS, false>(a2, (long)(bx - e1) * PT, c0, red);
    } else {
        if (vecmask & 8) pw_mfma_tile<KS, true>(a3, (long)(bx - e2) * PT, c0, red);
        else pw_mfma_tile<KS, false>(a3, (long)(bx - e2) * PT, c0, red);
    }
}

// Two CHAINED narrow layers per point in one launch: y0 = act0(s0 (W0 x) + b0), y1 = act1(s1 (W1 y0) + b1), both written (the RandLA
// stem: fc0 9 -> 8 and the first block's mlp1 8 -> 16, RandLANet.py:19,683; y0 is also the block's shortcut input).  One thread per
// point; the sums run in the order of pointwise_kernel (k ascending from 0, fmaf), so both outputs equal the two launches' bit for bit.
struct PwChain2 {
    const float* x;                      // [B, C0, n]
    const float *w0t, *s0, *b0;          // [C0, C1], [C1] or NULL
    const float *w1t, *s1, *b1;          // [C1, C2]
    float *y0, *y1;                      // [B, C1, n], [B, C2, n]
    int n, C0, C1, C2, act0, act1;
    float slope0, slope1;
    long total;
};

__device__ __forceinline__ float pw_act(float y, int act, float slope)
{
    if (act == 1) return fmaxf(y, 0.f);
    if (act == 2) return y > 0.f ? y : y * slope;
    return y;
}

__global__ __launch_bounds__(256) void pointwise_chain2_kernel(const PwChain2 a)
{
    constexpr int CMAX0 = 16, CMAX1 = 16, CMAX2 = 32;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.total) return;
    const long b = g / a.n;
    const int i = (int)(g - b * a.n);
    float xin[CMAX0];
#pragma unroll
    for (int k = 0; k < CMAX0; ++k) xin[k] = k < a.C0 ? a.x[(b * a.C0 + k) * a.n + i] : 0.f;
    float h[CMAX1];
#pragma unroll
    for (int c = 0; c < CMAX1; ++c) {
        float acc = 0.f;
        if (c < a.C1) {
#pragma unroll
            for (int k = 0; k < CMAX0; ++k)
                if (k < a.C0) acc = fmaf(a.w0t[k * a.C1 + c], xin[k], acc);
            acc = pw_act(fmaf(acc, a.s0 ? a.s0[c] : 1.f, a.b0 ? a.b0[c] : 0.f), a.act0, a.slope0);
            a.y0[(b * a.C1 + c) * a.n + i] = acc;
        }
        h[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < CMAX2; ++c) {
        if (c < a.C2) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < CMAX1; ++k)
                if (k < a.C1) acc = fmaf(a.w1t[k * a.C2 + c], h[k], acc);
            a.y1[(b * a.C2 + c) * a.n + i] = pw_act(fmaf(acc, a.s1 ? a.s1[c] : 1.f, a.b1 ? a.b1[c] : 0.f), a.act1, a.slope1);
        }
    }
}

bool seg_ok(const gdm_pw_seg& s, int n)
{
    return s.x && s.C >= 1 && s.n_src >= 1 && (s.idx || s.n_src == n);
}

} // namespace

extern "C" int gdm_pointwise_hip(const gdm_pw_seg* segs, int nseg, const float* wt, const float* scale, const float* shift,
                                 int B, int n, int Cout, int act, float slope, float* out, int out_C, int out_c0, int point_major,
                                 void* stream)
{
    return gdm_pointwise2_hip(segs, nseg, wt, 0, scale, shift, B, n, Cout, act, slope, out, out_C, out_c0, point_major, stream);
}

extern "C" int gdm_pointwise2_hip(const gdm_pw_seg* segs, int nseg, const float* wt, int w_rowmajor, const float* scale, const float* shift,
                                  int B, int n, int Cout, int act, float slope, float* out, int out_C, int out_c0, int point_major,
                                  void* stream)
{
    GDM_CHECK_ARG(segs && wt && out, "gdm_pointwise_hip: NULL pointer");
    GDM_CHECK_ARG(nseg >= 1 && nseg <= MAXSEG, "gdm_pointwise_hip: nseg=%d not in [1,%d]", nseg, MAXSEG);
    GDM_CHECK_ARG(B >= 1 && n >= 1 && Cout >= 1, "gdm_pointwise_hip: bad shape B=%d n=%d Cout=%d", B, n, Cout);
    GDM_CHECK_ARG(out_c0 >= 0 && out_c0 + Cout <= out_C, "gdm_pointwise_hip: channels [%d, %d) outside the output's %d", out_c0,
                  out_c0 + Cout, out_C);
    GDM_CHECK_ARG(act >= 0 && act <= 2, "gdm_pointwise_hip: act=%d", act);
    PwArgs a;
    a.K = 0;
    for (int s = 0; s < MAXSEG; ++s) {
        if (s < nseg) {
            GDM_CHECK_ARG(seg_ok(segs[s], n), "gdm_pointwise_hip: segment %d: NULL / empty, or n_src=%d != n=%d without an index", s,
                          segs[s].n_src, n);
            a.seg[s] = PwSegDev{segs[s].x, segs[s].idx, segs[s].C, segs[s].n_src};
            a.K += segs[s].C;
        } else
            a.seg[s] = PwSegDev{nullptr, nullptr, 0, 0};
    }
    GDM_CHECK_ARG(((uintptr_t)out & 15) == 0, "gdm_pointwise_hip: out must be 16-byte aligned");
    a.nseg = nseg;
    a.wt = wt;
    a.wks = w_rowmajor ? 1 : Cout;
    a.wcs = w_rowmajor ? a.K : 1;
    a.scale = scale;
    a.shift = shift;
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
    GDM_CHECK_ARG(tiles <= 0x7fffffffL, "gdm_pointwise_hip: grid too large");
    hipStream_t st = (hipStream_t)stream;
    bool vec = (n % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)wt & 15) == 0) && !w_rowmajor;      // (the FMA form's 16-byte weight loads run along Cout)
    for (int s = 0; s < nseg; ++s) vec = vec && !segs[s].idx && (((uintptr_t)segs[s].x & 15) == 0);
#define GDM_PW_LAUNCH(NW, KS, COUT_PER_WG)                                                                                          \
    do {                                                                                                                            \
        const dim3 grid((unsigned)tiles, gdm_cdiv(Cout, COUT_PER_WG));                                                              \
        if (vec) hipLaunchKernelGGL((pointwise_kernel<NW, KS, true>), grid, dim3(NW * 64), 0, st, a);                               \
        else hipLaunchKernelGGL((pointwise_kernel<NW, KS, false>), grid, dim3(NW * 64), 0, st, a);                                  \
    } while (0)
    if (a.K >= 32 && gdm_cdiv(Cout, 16) <= 65535) {
        // K parts per tile: as many as leave each wave >= 16 rows, at most 8 (= waves of the workgroup)
        int ks = 1;
        while (ks < 8 && a.K / (2 * ks) >= 16 && tiles * gdm_cdiv(Cout, 16) * ks < 4096) ks *= 2;
        const dim3 grid((unsigned)tiles, gdm_cdiv(Cout, 16));
        bool mvec = (n % 4 == 0);
        for (int sgi = 0; sgi < nseg; ++sgi) mvec = mvec && !segs[sgi].idx && (((uintptr_t)segs[sgi].x & 15) == 0) && segs[sgi].n_src % 4 == 0;
#define GDM_PWM(KSV)                                                                                                    \
        do {                                                                                                             \
            if (mvec) hipLaunchKernelGGL((pointwise_mfma_kernel<KSV, true>), grid, dim3(KSV * 64), 0, st, a);            \
            else hipLaunchKernelGGL((pointwise_mfma_kernel<KSV, false>), grid, dim3(KSV * 64), 0, st, a);                \
        } while (0)
        // no K split and still >= 2048 workgroups with four (two) channel blocks per wave: each x row is then fetched Cout / 64 (/ 32)
        // times instead of Cout / 16
        if (ks == 1 && Cout >= 32 && tiles >= 1024) {
            // training sizes (B = 24: 64 -> 64 channels at 393 k points 95 -> 52 us, 576 -> 64 755 -> 430 us; two channel blocks per wave
            // measured equal): tools/bench_pointwise_train.py
            const dim3 gridc((unsigned)tiles, gdm_cdiv(Cout, 64));
            if (mvec) hipLaunchKernelGGL((pointwise_mfma_cw_kernel<true, 1>), gridc, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((pointwise_mfma_cw_kernel<false, 1>), gridc, dim3(256), 0, st, a);
            return gdm_launch_status("pointwise_mfma_cw_kernel");
        }
        if (ks == 1) GDM_PWM(1);
        else if (ks == 2) GDM_PWM(2);
        else if (ks == 4) GDM_PWM(4);
        else GDM_PWM(8);
#undef GDM_PWM
        return gdm_launch_status("pointwise_mfma_kernel");
    }
    // K < 32 (or more than 2^20 output channels): the FMA form, four waves per 64 x 64 tile, no K split
    GDM_CHECK_ARG(nseg <= 3, "gdm_pointwise_hip: four segments need K >= 32 (the MFMA form)");
    GDM_PW_LAUNCH(4, 1, 64);
#undef GDM_PW_LAUNCH
    return gdm_launch_status("pointwise_kernel");
}

extern "C" int gdm_pointwise_jobs_hip(const gdm_pw_job* jobs, int njobs, int B, int K, int Cout, void* stream)
{
    GDM_CHECK_ARG(jobs && njobs >= 1 && njobs <= 4, "gdm_pointwise_jobs_hip: njobs=%d not in [1,4]", njobs);
    GDM_CHECK_ARG(B >= 1 && K >= 32 && Cout >= 1 && gdm_cdiv(Cout, 16) <= 65535, "gdm_pointwise_jobs_hip: bad shape B=%d K=%d Cout=%d", B, K, Cout);
    // the K split of gdm_pointwise_hip is chosen PER JOB there (from its own grid size); a job launched here must agree with its separate
    // launch bit for bit, so jobs are launched together only with the jobs that take the same split: one launch per distinct split
    // (the pyramid-pooling products at batch 16: all four take eight parts -> one launch; at batch 32 the 36-bin job takes four -> two)
    int kjob[4] = {0, 0, 0, 0};
    for (int j = 0; j < njobs; ++j) {
        const gdm_pw_job& jb = jobs[j];
        GDM_CHECK_ARG(jb.x && jb.wt && jb.out && jb.n >= 1, "gdm_pointwise_jobs_hip: job %d: NULL pointer or n=%d", j, jb.n);
        GDM_CHECK_ARG(((uintptr_t)jb.out & 15) == 0, "gdm_pointwise_jobs_hip: job %d: out must be 16-byte aligned", j);
        const long tj = ((long)B * jb.n + PT - 1) / PT;
        int kj = 1;
        while (kj < 8 && K / (2 * kj) >= 16 && tj * gdm_cdiv(Cout, 16) * kj < 4096) kj *= 2;
        kjob[j] = kj;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int ks = 1; ks <= 8; ks *= 2) {
        PwArgs a[4];
        int ends[4];
        int vecmask = 0, m = 0;
        long tiles = 0;
        const gdm_pw_job* last = nullptr;
        for (int j = 0; j < njobs; ++j) {
            if (kjob[j] != ks) continue;
            const gdm_pw_job& jb = jobs[j];
            last = &jb;
            for (int sgi = 0; sgi < MAXSEG; ++sgi) a[m].seg[sgi] = PwSegDev{nullptr, nullptr, 0, 0};
            a[m].seg[0] = PwSegDev{jb.x, nullptr, K, jb.n};
            a[m].nseg = 1;
            a[m].wt = jb.wt;
            a[m].wks = Cout;
            a[m].wcs = 1;
            a[m].scale = nullptr;
            a[m].shift = nullptr;
            a[m].out = jb.out;
            a[m].n = jb.n;
            a[m].Cout = Cout;
            a[m].outC = Cout;
            a[m].out_c0 = 0;
            a[m].point_major = 0;
            a[m].act = 0;
            a[m].slope = 0.f;
            a[m].K = K;
            a[m].total = (long)B * jb.n;
            tiles += (a[m].total + PT - 1) / PT;
            if (jb.n % 4 == 0 && ((uintptr_t)jb.x & 15) == 0) vecmask |= 1 << m;
            ends[m] = (int)tiles;
            ++m;
        }
        if (m == 0) continue;
        for (int q = m; q < 4; ++q) {                                  // unused slots: valid pointers, no blocks
            a[q] = a[m - 1];
            ends[q] = (int)tiles;
        }
        (void)last;
        GDM_CHECK_ARG(tiles <= 0x7fffffffL, "gdm_pointwise_jobs_hip: grid too large");
        const dim3 grid((unsigned)tiles, gdm_cdiv(Cout, 16));
#define GDM_PWJ(KSV) hipLaunchKernelGGL((pointwise_mfma_jobs_kernel<KSV>), grid, dim3(KSV * 64), 0, st, a[0], a[1], a[2], a[3], ends[0], ends[1], ends[2], vecmask)
        if (ks == 1) GDM_PWJ(1);
        else if (ks == 2) GDM_PWJ(2);
        else if (ks == 4) GDM_PWJ(4);
        else GDM_PWJ(8);
#undef GDM_PWJ
        const int rc = gdm_launch_status("pointwise_mfma_jobs_kernel");
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gdm_pointwise_chain2_hip(const float* x, const float* w0t, const float* s0, const float* b0, int act0, float slope0,
                                        const float* w1t, const float* s1, const float* b1, int act1, float slope1,
                                        int B, int n, int C0, int C1, int C2, float* y0, float* y1, void* stream)
{
    GDM_CHECK_ARG(x && w0t && w1t && y0 && y1, "gdm_pointwise_chain2_hip: NULL pointer");
    GDM_CHECK_ARG(B >= 1 && n >= 1 && C0 >= 1 && C0 <= 16 && C1 >= 1 && C1 <= 16 && C2 >= 1 && C2 <= 32,
                  "gdm_pointwise_chain2_hip: B=%d n=%d C0=%d (<= 16) C1=%d (<= 16) C2=%d (<= 32)", B, n, C0, C1, C2);
    GDM_CHECK_ARG(act0 >= 0 && act0 <= 2 && act1 >= 0 && act1 <= 2, "gdm_pointwise_chain2_hip: act=%d,%d", act0, act1);
    PwChain2 a;
    a.x = x; a.w0t = w0t; a.s0 = s0; a.b0 = b0; a.w1t = w1t; a.s1 = s1; a.b1 = b1; a.y0 = y0; a.y1 = y1;
    a.n = n; a.C0 = C0; a.C1 = C1; a.C2 = C2; a.act0 = act0; a.act1 = act1; a.slope0 = slope0; a.slope1 = slope1;
    a.total = (long)B * n;
    GDM_CHECK_ARG(gdm_cdiv(a.total, 256) <= 0x7fffffff, "gdm_pointwise_chain2_hip: grid too large");
    hipLaunchKernelGGL(pointwise_chain2_kernel, dim3(gdm_cdiv(a.total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return gdm_launch_status("pointwise_chain2_kernel");
}
