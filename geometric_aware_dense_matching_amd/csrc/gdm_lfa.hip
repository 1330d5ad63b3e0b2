// RandLA-Net local feature aggregation, one attentive-pooling stage per launch, gfx950 (inference).
//
// Replaces, per stage, the chain of /root/reference/models/RandLA/RandLANet.py:
//   relative_pos_encoding :720-727  ->  mlp1 (10 -> d/2, BN, LeakyReLU) :702  [-> mlp2 (d/2 -> d/2) :713 in the second stage]
//   gather_neighbour :729-738       ->  cat([f_neighbours, f_xyz]) :705/:716
//   Att_pooling.forward :747-754        fc (d x d, no bias), softmax over K, feature * score, sum over K, mlp (d -> out, BN, LeakyReLU)
// The reference (and the unfused path of this package) materialises five [B, d, n, K] tensors per stage for these steps (gathered
// features, encoded positions, their concat, the attention logits, the softmax) -- the memory-bound part of SURVEY.md row a4.
// Here a stage reads the point features [B, d/2, n], xyz and the neighbour index once, keeps the d x K block of a point in LDS
// and writes [B, out, n]; nothing of size n*K goes to HBM.  Two launches per Building_block instead of eighteen.
//
// Thread = (point slot p, channel c): a workgroup of 256 threads owns 256/D points.  Weights come pre-transposed ([in][out]),
// so the load of W^T[j][c] is contiguous over c; the d x K block is broadcast-read from LDS as 16-byte quads; the softmax over
// K and the weighted sum are in-thread.  fp32 FMAs; K = 16.
#include "gdm_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int LK = 16;

struct LfaArgs {
    const float* xyz;      // [B, n, 3]
    const int32_t* idx;    // [B, n, 16]
    const float* feat;     // [B, D/2, n]
    const float* w1t;      // [10, D/2]        mlp1 weight transposed
    const float* s1;       // [D/2]            folded BN scale / shift
    const float* b1;
    const float* w2t;      // [D/2, D/2] or NULL (first stage)
    const float* s2;
    const float* b2;
    const float* wft;      // [D, D]           Att_pooling.fc transposed
    const float* wmt;      // [D, OUT]         Att_pooling.mlp transposed
    const float* sm;       // [OUT]
    const float* bm;
    float* out;            // [B, OUT, n]
    int n, OUT;
    float slope;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// (Rounds 1-2 ran the stage as `lfa_stage_kernel`, thread = (point, channel) with every product as fp32 FMAs on broadcast LDS reads,
// 35-63 us per launch; it stayed behind GDM_LFA_MFMA=0 as an A/B leg until round 4 and was removed with that switch: no caller or test
// reached it.  The phases the MFMA kernel below did not change -- position encoding, mlp1 + gather, the final mlp -- are written out in it.)

// ---------------------------------------------------------------------------------------------------------------------------
// The same stage with its two large products -- mlp2 (d/2 x d/2 per neighbour) and the attention logits (d x d per neighbour) -- on
// v_mfma_f32_16x16x4_f32 (exact fp32 products).  In the FMA form every thread read the whole d x 16 block of its point from
// LDS (four 16-byte broadcast reads per 16 FMAs): the stage was bound by LDS issue, 35-63 us per launch for ~1 GFLOP.  Here a tile is
// 16 channels x the 16 neighbours of one point:
//   A fragment (block in LDS):   lane l = neighbour l & 15, input row j + (l >> 4)                     (256 contiguous bytes, conflict-free)
//   B fragment (W^T [in][out]):  lane l = output channel c0 + (l & 15), input row j + (l >> 4)         (global, 64-byte runs)
//   accumulator:                 lane l = output channel c0 + (l & 15), register r = neighbour 4 (l >> 4) + r
// so a lane holds four neighbours of ONE channel: the softmax over the 16 neighbours is three in-lane steps plus two cross-row
// exchanges (lanes l, l ^ 16, l ^ 32, l ^ 48), and the feature the weighted sum needs -- fcat[c][4 (l >> 4) .. + 4] -- is one
// conflict-free 16-byte LDS read; mlp2's output goes back to the block as one 16-byte LDS write per lane.
// A workgroup owns P = 512 / D points (phases 1, 2 and 5 as above with two points per thread group); its 4 waves tile (channel tiles) x
// (points).  One LDS dword read + one global dword load per 1024 products instead of 64 bytes of LDS per 16.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float lfa_f32x4;

// all-reduce over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (same l & 15), in one fixed order for every lane
__device__ __forceinline__ float quadrow_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float quadrow_sum(float v, int kq)
{
    float o = __shfl_xor(v, 16);
    v = (kq & 1) ? o + v : v + o;
    o = __shfl_xor(v, 32);
    v = (kq & 2) ? o + v : v + o;
    return v;
}

// acc[i][j] += W^T[:, row tile rt0 + WR i]^T . block of point pt0 + WP j, over KD input rows; src = LDS blocks [P][KD rows][16], wt = [KD][ROWS]
template <int ROWS, int KD, int NRT, int NPT, int WR, int WP>
__device__ __forceinline__ void lfa_tiles(const float* __restrict__ wt, const float* src, int src_pstride, int rt0, int pt0, int l16, int kq,
                                          lfa_f32x4 (&acc)[NRT][NPT])
{
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NPT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const float* wl = wt + (long)kq * ROWS + rt0 * 16 + l16;
    const float* sl = src + (long)pt0 * src_pstride + kq * LK + l16;
    // groups of G k-steps (4 G input rows); the next group's W fragments are in flight (L2 latency) behind this group's MFMAs
    constexpr int G = 4;
    static_assert(KD % (4 * G) == 0, "KD in whole groups");
    float an[G][NRT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < NRT; ++i) an[g][i] = wl[(long)(4 * g) * ROWS + i * WR * 16];
    for (int jj = 0; jj < KD; jj += 4 * G) {
        float ac[G][NRT], bv[G][NPT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < NRT; ++i) ac[g][i] = an[g][i];
        if (jj + 4 * G < KD) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < NRT; ++i) an[g][i] = wl[(long)(jj + 4 * G + 4 * g) * ROWS + i * WR * 16];
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < NPT; ++j) bv[g][j] = sl[(jj + 4 * g) * LK + j * WP * src_pstride];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < NRT; ++i)
#pragma unroll
                for (int j = 0; j < NPT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[g][j], ac[g][i], acc[i][j], 0, 0, 0);
    }
}

template <int D, int PP>
__global__ __launch_bounds__(256) void lfa_stage_mfma_kernel(const LfaArgs a)
{
    constexpr int H = D / 2;
    constexpr int P = (256 / D) * PP;                  // points per workgroup
    __shared__ __attribute__((aligned(16))) float fcat[P][D][LK];
    __shared__ __attribute__((aligned(16))) float fx1[P][H][LK];
    __shared__ float pe[P][10][LK];
    __shared__ int nidx[P][LK];
    __shared__ float aggv[P][D];

    const int tid = threadIdx.x;
    const int slot = tid / D, c = tid - slot * D;
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y;
    const int n = a.n;
    const float slope = a.slope;
    int pidx[PP];
    bool live[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int raw = (int)blockIdx.x * P + slot * PP + pp;
        live[pp] = raw < n;
        pidx[pp] = min(raw, n - 1);
    }

    // 1. relative position encoding (as lfa_stage_kernel)
    if (c < LK) {
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const int p = slot * PP + pp, i = pidx[pp];
            int jn = a.idx[((long)b * n + i) * LK + c];
            jn = min(max(jn, 0), n - 1);
            const float* pi = a.xyz + ((long)b * n + i) * 3;
            const float* pj = a.xyz + ((long)b * n + jn) * 3;
            const float ax = pi[0], ay = pi[1], az = pi[2];
            const float bx = pj[0], by = pj[1], bz = pj[2];
            const float rx = __fsub_rn(ax, bx), ry = __fsub_rn(ay, by), rz = __fsub_rn(az, bz);
            float s = __fmul_rn(rx, rx);
            s = __fadd_rn(s, __fmul_rn(ry, ry));
            s = __fadd_rn(s, __fmul_rn(rz, rz));
            pe[p][0][c] = __fsqrt_rn(s);
            pe[p][1][c] = rx; pe[p][2][c] = ry; pe[p][3][c] = rz;
            pe[p][4][c] = ax; pe[p][5][c] = ay; pe[p][6][c] = az;
            pe[p][7][c] = bx; pe[p][8][c] = by; pe[p][9][c] = bz;
            nidx[p][c] = jn;
        }
    }
    __syncthreads();

    // 2. mlp1 on the encoding and the neighbour gather (as lfa_stage_kernel)
    {
        const int j = c % H, kb = (c / H) * 8;
        float w[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) w[q] = a.w1t[q * H + j];
        const float sc = a.s1[j], sh = a.b1[j];
        const float* frow = a.feat + ((long)b * H + j) * n;
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const int p = slot * PP + pp;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = kb + kk;
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 10; ++q) v = fmaf(w[q], pe[p][q][k], v);
                v = lrelu(fmaf(v, sc, sh), slope);
                if (a.w2t) fx1[p][j][k] = v;
                else fcat[p][H + j][k] = v;
                fcat[p][j][k] = frow[nidx[p][k]];
            }
        }
    }
    __syncthreads();

    // 2b. second stage: mlp2 on the encoded positions, tiles of (16 channels of H) x (point)
    if (a.w2t) {
        constexpr int RT = H / 16, WR = RT < 4 ? RT : 4, WP = 4 / WR, NRT = RT / WR, NPT = P / WP;
        static_assert(RT >= 1 && NRT * WR == RT && NPT * WP == P, "mlp2 tiling");
        const int wr = wave % WR, wp = wave / WR;
        lfa_f32x4 acc[NRT][NPT];
        lfa_tiles<H, H, NRT, NPT, WR, WP>(a.w2t, &fx1[0][0][0], H * LK, wr, wp, l16, kq, acc);
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int ch = (wr + WR * i) * 16 + l16;
            const float sc = a.s2[ch], sh = a.b2[ch];
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                float4 o;
                o.x = lrelu(fmaf(acc[i][j][0], sc, sh), slope); o.y = lrelu(fmaf(acc[i][j][1], sc, sh), slope);
                o.z = lrelu(fmaf(acc[i][j][2], sc, sh), slope); o.w = lrelu(fmaf(acc[i][j][3], sc, sh), slope);
                *reinterpret_cast<float4*>(&fcat[wp + WP * j][H + ch][4 * kq]) = o;
            }
        }
        __syncthreads();
    }

    // 3 + 4. attention logits, softmax over the 16 neighbours (= the 16 lanes of a row), feature * score, sum
    {
        constexpr int RT = D / 16, WR = RT < 4 ? RT : 4, WP = 4 / WR, NRT = RT / WR, NPT = P / WP;
        static_assert(NRT * WR == RT && NPT * WP == P, "attention tiling");
        const int wr = wave % WR, wp = wave / WR;
        lfa_f32x4 acc[NRT][NPT];
        lfa_tiles<D, D, NRT, NPT, WR, WP>(a.wft, &fcat[0][0][0], D * LK, wr, wp, l16, kq, acc);
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int ch = (wr + WR * i) * 16 + l16;
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int p = wp + WP * j;
                const lfa_f32x4 t = acc[i][j];
                const float mx = quadrow_max(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])));
                const float e0 = __expf(t[0] - mx), e1 = __expf(t[1] - mx), e2 = __expf(t[2] - mx), e3 = __expf(t[3] - mx);
                const float den = quadrow_sum((e0 + e1) + (e2 + e3), kq);
                const float4 f = *reinterpret_cast<const float4*>(&fcat[p][ch][4 * kq]);
                const float rden = 1.0f / den;
                const float num = quadrow_sum((f.x * (e0 * rden) + f.y * (e1 * rden)) + (f.z * (e2 * rden) + f.w * (e3 * rden)), kq);
                if (kq == 0) aggv[p][ch] = num;
            }
        }
    }
    __syncthreads();

    // 5. mlp on the pooled feature (as lfa_stage_kernel)
    const int OUT = a.OUT;
    if (c < OUT) {
        float o[PP];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) o[pp] = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < D; ++jj) {
            const float w = a.wmt[jj * OUT + c];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) o[pp] = fmaf(w, aggv[slot * PP + pp][jj], o[pp]);
        }
        const float sm = a.sm[c], bm = a.bm[c];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
            if (live[pp]) a.out[((long)b * OUT + c) * n + pidx[pp]] = lrelu(fmaf(o[pp], sm, bm), slope);
    }
}

} // namespace

extern "C" int gdm_lfa_stage_hip(const float* xyz, const int32_t* idx, const float* feat, const float* w1t, const float* s1, const float* b1,
                                 const float* w2t, const float* s2, const float* b2, const float* wft, const float* wmt, const float* sm,
                                 const float* bm, int B, int n, int K, int D, int OUT, float slope, float* out, void* stream)
{
    GDM_CHECK_ARG(xyz && idx && feat && w1t && s1 && b1 && wft && wmt && sm && bm && out, "gdm_lfa_stage_hip: NULL pointer");
    GDM_CHECK_ARG(!w2t || (s2 && b2), "gdm_lfa_stage_hip: w2t without s2/b2");
    GDM_CHECK_ARG(K == LK, "gdm_lfa_stage_hip: K=%d, only K=16 is built", K);
    GDM_CHECK_ARG(D == 32 || D == 64 || D == 128 || D == 256, "gdm_lfa_stage_hip: D=%d not in {32,64,128,256}", D);
    GDM_CHECK_ARG(OUT >= 1 && OUT <= D, "gdm_lfa_stage_hip: OUT=%d must be in [1, D=%d]", OUT, D);
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && n >= 1, "gdm_lfa_stage_hip: bad shape");
    LfaArgs a{xyz, idx, feat, w1t, s1, b1, w2t, s2, b2, wft, wmt, sm, bm, out, n, OUT, slope};
    hipStream_t s = (hipStream_t)stream;
    // the MFMA form at every level, one point per thread group at D = 32 / 64 and two above (two points per group at the deep levels
    // for the FMA phases was measured slower: 71 -> 82 us and 75 -> 120 us per block of two stages at batch 16).  Per block of two
    // stages at batch 16 (FMA form of rounds 1-2 -> this): 109 -> 96, 82 -> 69, 70 -> 60, 73 -> 60 us
    const int pp = D <= 64 ? 1 : 2;
    const int P = (256 / D) * pp;
    dim3 g(gdm_cdiv(n, P), B);
    if (D == 32) hipLaunchKernelGGL((lfa_stage_mfma_kernel<32, 1>), g, dim3(256), 0, s, a);
    else if (D == 64) hipLaunchKernelGGL((lfa_stage_mfma_kernel<64, 1>), g, dim3(256), 0, s, a);
    else if (D == 128) hipLaunchKernelGGL((lfa_stage_mfma_kernel<128, 2>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lfa_stage_mfma_kernel<256, 2>), g, dim3(256), 0, s, a);
    return gdm_launch_status("lfa_stage_mfma_kernel");
}
