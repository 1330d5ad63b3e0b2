// Training matching loss WITHOUT the similarity matrix, gfx950: cosine similarity of the selected scene points against the model
// vertices on the matrix cores, with the circle loss's two masked log-sum-exps formed in the accumulator registers (forward) and
// the similarity tile recomputed in the backward kernels, flash-attention style.  Nothing of size [n_sel, M+1] touches HBM.
//
// Replaces, for all selected points of a batch at once,
//   /root/reference/models/geoMatch.py:117-136   normalise, cat(-1 column), matmul -> similarity [n_i, M+1]      (per item)
//   /root/reference/models/geoMatch.py:55-83     matching_loss: 3-D radius test -> boolean mask [n_i, M+1]
//   /root/reference/models/geoMatch.py:86-100    matching_loss_sys: two positive columns per row (symmetric objects)
//   /root/reference/models/loss.py:441-459,470-494  CircleLoss: masked LSEs, softplus
// and their autograd backward (two more [n_i, M+1] x 128 GEMMs per item).
//
// Row r = a selected scene point (unit descriptor x_r, batch item b_r), column c = a model vertex (unit descriptor y_c), column M
// = the reference's padding column (every component -1/sqrt(128)):
//   s = <x_r, y_c>          in = column c is a positive of row r        (constants m = 0.2, gamma = 16)
//   logit = gamma * a * d,   in : a = max(1 + m - s, 0), d = (1 - m) - s        out: a = max(s + m, 0), d = s - m
//   loss_r = softplus(LSE_in(logit) + LSE_out(logit));   dloss_r/ds = sigmoid(.) * softmax_within_its_set * (in ? -a : a) * gamma
// (a is a constant for the gradient, as the reference detaches it).  An empty positive set gives loss 0 / gradient 0.
// Positives: non-symmetric objects -- vertex c is visible in item b_r AND within `radius` of the ground-truth vertex g_r
// (sqrt(|xyz_g - xyz_c|^2 + 1e-7) < radius, utils/basic_utils.py:86-89); the radius test depends on the model only, so it is a bit
// table nbr[M][M/32] built once per model (cm_nbr_kernel, the reference's fp32 arithmetic), ANDed with the item's visibility
// bits.  Symmetric objects -- columns c1_r and c2_r.  The padding column is positive iff the row has no ground-truth vertex.
//
// One kernel template, three modes.  "Owner" items live in registers (32 per wave, whole K = 128 as split-bf16 fragments), the
// other side is streamed through LDS in 64-item stages:
//   MODE 0  forward   owner = scene rows, stream = vertices: S tile (24 MFMAs) -> both running sums per lane; out: lse_p, lse_n, loss
//   MODE 1  grad x    owner = scene rows, stream = vertices: S tile again, G = dloss/ds in the accumulator layout, then
//                     gX^T[d, r] += Y^T[d, c] G[c, r] (24 MFMAs) with G taken STRAIGHT from the accumulator registers as the B
//                     operand: the stream's d-major copy is packed in the k-order the accumulator layout dictates (cm_pack_kernel)
//   MODE 2  grad y    owner = vertices, stream = scene rows (a slice of them per workgroup; partial sums reduced by the caller)
// Products are split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate), as in the inference matching kernel.
#include "gdm_common.h"
#include <math.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int ROWB = 512;                       // packed row: 128 bf16 hi | 128 bf16 lo
constexpr int CM_THREADS = 256;                 // 4 waves
constexpr int CM_OWN = 128;                     // owner items per workgroup (32 per wave)
constexpr int CM_ST = 64;                       // streamed items per LDS stage (two 32-item sub-tiles)
constexpr int TP_G = 128 * 64;                  // bytes of one plane of a d-major 32-item sub-tile in global memory
constexpr int TP_LSTRIDE = 80;                  // LDS bytes per d row of it (64 + 16 pad: conflict-free ds_read_b128)
constexpr int TP_L = 128 * TP_LSTRIDE;
constexpr int LDS_ROWS = CM_ST * ROWB;                       // 32 KiB
constexpr int LDS_TP = (CM_ST / 32) * 2 * TP_L;              // 40 KiB
constexpr int LDS_RD = 4 * CM_ST * 16;                       // MODE 2: per-wave row data

__device__ __forceinline__ unsigned pack2(float a, float b)
{
    // plain casts: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float hi_of(float a) { return (float)(__bf16)a; }

__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = pack2(v[2 * j], v[2 * j + 1]);
        lo[j] = pack2(v[2 * j] - hi_of(v[2 * j]), v[2 * j + 1] - hi_of(v[2 * j + 1]));
    }
}

__device__ __forceinline__ int swz(int row, int ch) { return row * ROWB + (((ch & 16) | ((ch ^ row) & 15)) << 4); }

// accumulator register q (0..7) of k-step ks, lane half h  ->  streamed index inside a 32-item sub-tile
__host__ __device__ __forceinline__ int acc_row(int ks, int h, int q) { return (q & 3) + 8 * (2 * ks + (q >> 2)) + 4 * h; }

// x f32[n,128] (unit rows) -> rows[npad] (512-B split-bf16 rows, zero beyond n), tp[npad/32][2 planes][128 d][4 x 16 B] (the same
// values d-major, the 8 values of a 16-B piece in accumulator order: piece (ks, h) holds items acc_row(ks, h, 0..7)), rowsum[npad].
__global__ __launch_bounds__(256) void cm_pack_kernel(const float* __restrict__ x, int n, unsigned char* __restrict__ rows,
                                                      unsigned char* __restrict__ tp, float* __restrict__ rowsum)
{
    __shared__ float t[32][129];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x;
    for (int e = tid; e < 32 * 128; e += 256) {
        const int r = e >> 7, d = e & 127;
        const long gr = (long)tile * 32 + r;
        t[r][d] = gr < n ? x[gr * 128 + d] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {                       // row-major: (row, 16-B chunk of 8 channels)
        const int r = e >> 4, ch = e & 15;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t[r][ch * 8 + j];
        u32x4 hi, lo;
        split8(v, hi, lo);
        unsigned char* o = rows + ((long)tile * 32 + r) * ROWB;
        *reinterpret_cast<u32x4*>(o + ch * 16) = hi;
        *reinterpret_cast<u32x4*>(o + 256 + ch * 16) = lo;
    }
    for (int e = tid; e < 512; e += 256) {                       // d-major: (d, piece = 2 ks + h)
        const int d = e >> 2, pc = e & 3;
        const int ks = pc >> 1, h = pc & 1;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = t[acc_row(ks, h, q)][d];
        u32x4 hi, lo;
        split8(v, hi, lo);
        unsigned char* o = tp + (long)tile * 2 * TP_G + d * 64 + pc * 16;
        *reinterpret_cast<u32x4*>(o) = hi;
        *reinterpret_cast<u32x4*>(o + TP_G) = lo;
    }
    if (tid < 32) {
        float s = 0.f;
        for (int d = 0; d < 128; ++d) s += t[tid][d];
        rowsum[(long)tile * 32 + tid] = s;
    }
}

// nbr[g][w] bit k = sqrt(|xyz_g - xyz_c|^2 + 1e-7) < radius for c = 32 w + k < M   (the reference's arithmetic, basic_utils.py:88-89)
__global__ __launch_bounds__(256) void cm_nbr_kernel(const float* __restrict__ xyz, int M, int W, float radius, unsigned* __restrict__ nbr)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * W) return;
    const int g = (int)(i / W), w = (int)(i - (long)g * W);
    const float gx = xyz[3 * g], gy = xyz[3 * g + 1], gz = xyz[3 * g + 2];
    unsigned bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int c = 32 * w + k;
        if (c >= M) break;
        const float dx = gx - xyz[3 * c], dy = gy - xyz[3 * c + 1], dz = gz - xyz[3 * c + 2];
        float d2 = __fmul_rn(dx, dx);
        d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
        d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
        if (__fsqrt_rn(__fadd_rn(d2, 1e-7f)) < radius) bits |= 1u << k;
    }
    nbr[i] = bits;
}

// The same table PER ITEM with a per-vertex radius (geoMatch_DGCNN.py:62-70: positive_r / 1000 * z of the posed vertex):
// nbr[b][g][w] bit k = sqrt(|xyz_g - xyz_c|^2 + 1e-7) < rad[b][c]
__global__ __launch_bounds__(256) void cm_nbr_items_kernel(const float* __restrict__ xyz, int M, int W, const float* __restrict__ rad, int B,
                                                           unsigned* __restrict__ nbr)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * M * W) return;
    const int w = (int)(i % W);
    const long bg = i / W;
    const int g = (int)(bg % M), b = (int)(bg / M);
    const float gx = xyz[3 * g], gy = xyz[3 * g + 1], gz = xyz[3 * g + 2];
    const float* rb = rad + (long)b * M;
    unsigned bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int c = 32 * w + k;
        if (c >= M) break;
        const float dx = gx - xyz[3 * c], dy = gy - xyz[3 * c + 1], dz = gz - xyz[3 * c + 2];
        float d2 = __fmul_rn(dx, dx);
        d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
        d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
        if (__fsqrt_rn(__fadd_rn(d2, 1e-7f)) < rb[c]) bits |= 1u << k;
    }
    nbr[i] = bits;
}

// vis u8[B,M] (nonzero = visible) -> bits[B][W]
__global__ __launch_bounds__(256) void cm_visbits_kernel(const unsigned char* __restrict__ vis, int B, int M, int W, unsigned* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * W) return;
    const int b = (int)(i / W), w = (int)(i - (long)b * W);
    unsigned bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int c = 32 * w + k;
        if (c < M && vis[(long)b * M + c]) bits |= 1u << k;
    }
    out[i] = bits;
}

struct CmArgs {
    const unsigned char* xrows;     // scene rows, packed [Rp]
    const unsigned char* xtp;       // scene d-major tiles
    const unsigned char* yrows;     // vertex rows, packed [Mp]
    const unsigned char* ytp;
    const float* xsum;              // pad_e0 == 0: sum_d x[r][d] (padding column = every component -1/sqrt(128)); pad_e0: x[r][0] (column e0)
    const int32_t* g;               // [Rp] ground-truth vertex (M = none) / first positive column (symmetric)
    const int32_t* c2;              // [Rp] second positive column (symmetric) or null
    const int32_t* item;            // [Rp]
    const unsigned* nbr;            // [M][W] (non-symmetric)
    const unsigned* visb;           // [B][W] (non-symmetric)
    float* lse_p;                   // [Rp]  forward out / backward in
    float* lse_n;
    float* loss;                    // [Rp]  forward out
    const float* coef;              // [Rp]  backward: upstream gradient x sigmoid(lse_p + lse_n), 0 for padding / empty rows
    float* gout;                    // MODE 1: gX [Rp,128]; MODE 2: partial gY [P][Mp,128]
    int R, Rp, M, Mp, W, P;
    float gamma, m, offp, offn;
    long nbr_istride;               // words between two items' neighbour tables (0: one table, the model's)
    int pad_e0;                     // the padding column is the unit vector e0 (geoMatch_DGCNN.py:96-99) instead of -1/sqrt(128) everywhere
};

// word of positives of scene row (g, c2, item) inside the 32-vertex tile t
template <bool SYM>
__device__ __forceinline__ unsigned pos_word(const CmArgs& a, int g, int c2, int item, int t)
{
    if (SYM) {
        unsigned w = 0;
        if (g < a.M && (g >> 5) == t) w |= 1u << (g & 31);
        if (c2 >= 0 && c2 < a.M && (c2 >> 5) == t) w |= 1u << (c2 & 31);
        return w;
    }
    if (g >= a.M || t >= a.W) return 0u;
    return a.nbr[(long)item * a.nbr_istride + (long)g * a.W + t] & a.visb[(long)item * a.W + t];
}

template <int MODE, bool SYM>
__global__ __launch_bounds__(CM_THREADS, 2) void circle_mm_kernel(const CmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lrows = smem;
    unsigned char* ltp = smem + LDS_ROWS;
    float4* lrd = reinterpret_cast<float4*>(smem + LDS_ROWS + LDS_TP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const bool own_is_x = MODE != 2;
    const int nown_blocks = (own_is_x ? a.Rp : a.Mp) / CM_OWN;
    const int ob = blockIdx.x % nown_blocks;            // owner block
    const int part = blockIdx.x / nown_blocks;          // MODE 2: slice of the stream
    const int own0 = ob * CM_OWN + wave * 32;           // this wave's first owner item
    const unsigned char* orows = own_is_x ? a.xrows : a.yrows;
    const unsigned char* srows = own_is_x ? a.yrows : a.xrows;
    const unsigned char* stp = own_is_x ? a.ytp : a.xtp;
    const int nstage = (own_is_x ? a.Mp : a.Rp) / CM_ST;

    // owner operand: 8 k-steps x (hi, lo)
    u32x4 ohi[8], olo[8];
    {
        const unsigned char* r = orows + (long)(own0 + j) * ROWB + h * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            ohi[s] = *reinterpret_cast<const u32x4*>(r + s * 32);
            olo[s] = *reinterpret_cast<const u32x4*>(r + 256 + s * 32);
        }
    }
    // per-lane constants of the owner row (MODE 0 / 1)
    int rg = 0, rc2 = -1, ritem = 0;
    float rlp = 0.f, rln = 0.f, rcoef = 0.f;
    if (own_is_x) {
        rg = a.g[own0 + j];
        rc2 = (SYM && a.c2) ? a.c2[own0 + j] : -1;
        ritem = a.item[own0 + j];
        if (MODE == 1) {
            rlp = a.lse_p[own0 + j];
            rln = a.lse_n[own0 + j];
            rcoef = a.coef[own0 + j];
        }
    }
    const float gam = a.gamma, mm = a.m;
    float sum_p = 0.f, sum_n = 0.f;

    f32x16 outacc[4];
    if (MODE != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) outacc[c][i] = 0.f;
    }

    for (int st = (MODE == 2 ? part : 0); st < nstage; st += (MODE == 2 ? a.P : 1)) {
        __syncthreads();                                            // the previous stage's readers are done
        {
            const unsigned char* src = srows + (long)st * CM_ST * ROWB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gch = i * CM_THREADS + tid;               // 2048 chunks of 16 B
                *reinterpret_cast<u32x4*>(lrows + swz(gch >> 5, gch & 31)) = *reinterpret_cast<const u32x4*>(src + (long)gch * 16);
            }
            if (MODE != 0) {
                const unsigned char* tsrc = stp + (long)st * (CM_ST / 32) * 2 * TP_G;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int gch = i * CM_THREADS + tid;           // (sub*2 + plane) * 512 + d * 4 + piece
                    const int sp = gch >> 9, d = (gch >> 2) & 127, pc = gch & 3;
                    *reinterpret_cast<u32x4*>(ltp + sp * TP_L + d * TP_LSTRIDE + pc * 16) = *reinterpret_cast<const u32x4*>(tsrc + (long)gch * 16);
                }
            }
            if (MODE == 2) {                                        // per streamed scene row: lse_p, lse_n, coef, positives word
                const int r = st * CM_ST + lane;                    //   of this wave's vertex tile
                const unsigned w = pos_word<SYM>(a, a.g[r], (SYM && a.c2) ? a.c2[r] : -1, a.item[r], own0 >> 5);
                lrd[wave * CM_ST + lane] = make_float4(a.lse_p[r], a.lse_n[r], a.coef[r], __uint_as_float(w));
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int sub = 0; sub < CM_ST / 32; ++sub) {
            const int t32 = st * (CM_ST / 32) + sub;                // index of this 32-item sub-tile in the stream
            unsigned word = 0;
            if (own_is_x) word = pos_word<SYM>(a, rg, rc2, ritem, t32);
            // ---- S tile: acc[i][j] = <stream_i, owner_j> ----
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16x8 sh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lrows + swz(sub * 32 + j, 2 * s + h)));
                const bf16x8 sl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lrows + swz(sub * 32 + j, 16 + 2 * s + h)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh, __builtin_bit_cast(bf16x8, olo[s]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl, __builtin_bit_cast(bf16x8, ohi[s]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh, __builtin_bit_cast(bf16x8, ohi[s]), acc, 0, 0, 0);
            }
            // ---- element-wise: register r <-> streamed item i = acc_row(r >> 3, h, r & 7), lane <-> owner item j ----
            const int cbase = t32 * 32;                             // MODE 0/1: first vertex of the sub-tile
            float G[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = acc_row(r >> 3, h, r & 7);
                const float s = acc[r];
                bool in, valid;
                float lp, ln, cf;
                if (own_is_x) {
                    in = (word >> i) & 1u;
                    valid = cbase + i < a.M;
                    lp = rlp; ln = rln; cf = rcoef;
                } else {
                    const float4 rd = lrd[wave * CM_ST + sub * 32 + i];
                    in = (__float_as_uint(rd.w) >> j) & 1u;
                    valid = true;                                   // padding rows carry coef = 0, padding vertices are dropped
                    lp = rd.x; ln = rd.y; cf = rd.z;
                }
                const float av = fmaxf(in ? (1.f + mm) - s : s + mm, 0.f);
                const float dv = in ? (1.f - mm) - s : s - mm;
                const float logit = av * dv * gam;
                if (MODE == 0) {
                    const float e = valid ? __expf(logit - (in ? a.offp : a.offn)) : 0.f;
                    sum_p += in ? e : 0.f;
                    sum_n += in ? 0.f : e;
                } else {
                    const float w = __expf(logit - (in ? lp : ln)) * (in ? -av : av) * gam * cf;
                    G[r] = (valid && cf != 0.f) ? w : 0.f;
                }
            }
            if (MODE == 0) continue;
            // ---- out^T[d][j] += sum_i stream^T[d][i] G[i][j]: G from the accumulator registers as the B operand ----
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 gh, gl;
                split8(&G[8 * ks], gh, gl);
                const bf16x8 bgh = __builtin_bit_cast(bf16x8, gh), bgl = __builtin_bit_cast(bf16x8, gl);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const unsigned char* p = ltp + (sub * 2) * TP_L + (db * 32 + j) * TP_LSTRIDE + (ks * 2 + h) * 16;
                    const bf16x8 th = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                    const bf16x8 tl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + TP_L));
                    outacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, bgl, outacc[db], 0, 0, 0);
                    outacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, bgh, outacc[db], 0, 0, 0);
                    outacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, bgh, outacc[db], 0, 0, 0);
                }
            }
        }
    }

    const float inv_sqrt_d = 0.08838834764831845f;                  // every component of the normalised padding column is -1/sqrt(128)
    if (MODE == 0) {
        sum_p += __shfl_xor(sum_p, 32, 64);                         // lanes j and j + 32 hold the two halves of row j's columns
        sum_n += __shfl_xor(sum_n, 32, 64);
        const float s = a.pad_e0 ? a.xsum[own0 + j] : -a.xsum[own0 + j] * inv_sqrt_d;
        const bool in = SYM ? (rg == a.M || rc2 == a.M) : (rg >= a.M);
        const float av = fmaxf(in ? (1.f + mm) - s : s + mm, 0.f);
        const float dv = in ? (1.f - mm) - s : s - mm;
        const float e = __expf(av * dv * gam - (in ? a.offp : a.offn));
        sum_p += in ? e : 0.f;
        sum_n += in ? 0.f : e;
        if (h == 0 && own0 + j < a.Rp) {
            const float lp = sum_p > 0.f ? __logf(sum_p) + a.offp : -INFINITY;
            const float ln = sum_n > 0.f ? __logf(sum_n) + a.offn : -INFINITY;
            const float z = lp + ln;
            a.lse_p[own0 + j] = lp;
            a.lse_n[own0 + j] = ln;
            a.loss[own0 + j] = (own0 + j < a.R && sum_p > 0.f) ? (z > 20.f ? z : log1pf(expf(z))) : 0.f;
        }
        return;
    }
    // ---- MODE 1 / 2: outacc[db][r] = grad[owner j][d = db*32 + acc_row(r)] ----
    float padg = 0.f, padg0 = 0.f;                                  // added to every channel / to channel 0 only
    if (MODE == 1) {                                                // the padding column's share: dS_pad * (-1/sqrt(128)) on every channel,
        const float s = a.pad_e0 ? a.xsum[own0 + j] : -a.xsum[own0 + j] * inv_sqrt_d;       // or dS_pad on channel 0 (column e0)
        const bool in = SYM ? (rg == a.M || rc2 == a.M) : (rg >= a.M);
        const float av = fmaxf(in ? (1.f + mm) - s : s + mm, 0.f);
        const float dv = in ? (1.f - mm) - s : s - mm;
        const float w = __expf(av * dv * gam - (in ? rlp : rln)) * (in ? -av : av) * gam * rcoef;
        if (a.pad_e0) padg0 = rcoef != 0.f ? w : 0.f;
        else padg = rcoef != 0.f ? -w * inv_sqrt_d : 0.f;
    }
    float* ob_out = a.gout + (MODE == 2 ? (long)part * a.Mp * 128 : 0L) + (long)(own0 + j) * 128;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int d = db * 32 + 8 * q4 + 4 * h;                 // registers 4 q4 .. 4 q4 + 3 = four consecutive channels
            *reinterpret_cast<float4*>(ob_out + d) = make_float4(outacc[db][4 * q4] + padg + (d == 0 ? padg0 : 0.f), outacc[db][4 * q4 + 1] + padg,
                                                                 outacc[db][4 * q4 + 2] + padg, outacc[db][4 * q4 + 3] + padg);
        }
}

template <int MODE>
int launch_mode(const CmArgs& a, bool sym, hipStream_t stream)
{
    const int lds = LDS_ROWS + (MODE != 0 ? LDS_TP : 0) + (MODE == 2 ? LDS_RD : 0);
    const int grid = MODE == 2 ? (a.Mp / CM_OWN) * a.P : a.Rp / CM_OWN;
    if (sym) {
        GDM_HIP(hipFuncSetAttribute((const void*)circle_mm_kernel<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((circle_mm_kernel<MODE, true>), dim3(grid), dim3(CM_THREADS), lds, stream, a);
    } else {
        GDM_HIP(hipFuncSetAttribute((const void*)circle_mm_kernel<MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((circle_mm_kernel<MODE, false>), dim3(grid), dim3(CM_THREADS), lds, stream, a);
    }
    return gdm_launch_status("circle_mm_kernel");
}

} // namespace

extern "C" size_t gdm_circle_match_rows_bytes(int n) { return n < 1 ? 0 : (size_t)((n + 127) / 128 * 128) * ROWB; }
extern "C" size_t gdm_circle_match_tp_bytes(int n) { return n < 1 ? 0 : (size_t)((n + 127) / 128 * 128) / 32 * 2 * TP_G; }

extern "C" int gdm_circle_match_pack_hip(const float* x, int n, void* rows, void* tp, float* rowsum, void* stream)
{
    GDM_CHECK_ARG(x && rows && tp && rowsum && n >= 1, "gdm_circle_match_pack_hip: NULL pointer or n=%d", n);
    const int np = (n + 127) / 128 * 128;
    hipLaunchKernelGGL(cm_pack_kernel, dim3(np / 32), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned char*)rows, (unsigned char*)tp, rowsum);
    return gdm_launch_status("cm_pack_kernel");
}

extern "C" int gdm_circle_match_nbr_hip(const float* xyz, int M, float radius, uint32_t* nbr, void* stream)
{
    GDM_CHECK_ARG(xyz && nbr && M >= 1, "gdm_circle_match_nbr_hip: NULL pointer or M=%d", M);
    const int W = (M + 31) / 32;
    hipLaunchKernelGGL(cm_nbr_kernel, dim3(gdm_cdiv((long)M * W, 256)), dim3(256), 0, (hipStream_t)stream, xyz, M, W, radius, nbr);
    return gdm_launch_status("cm_nbr_kernel");
}

extern "C" int gdm_circle_match_nbr_items_hip(const float* xyz, int M, const float* rad, int B, uint32_t* nbr, void* stream)
{
    GDM_CHECK_ARG(xyz && rad && nbr && M >= 1 && B >= 1, "gdm_circle_match_nbr_items_hip: NULL pointer or M=%d B=%d", M, B);
    const int W = (M + 31) / 32;
    hipLaunchKernelGGL(cm_nbr_items_kernel, dim3(gdm_cdiv((long)B * M * W, 256)), dim3(256), 0, (hipStream_t)stream, xyz, M, W, rad, B, nbr);
    return gdm_launch_status("cm_nbr_items_kernel");
}

extern "C" int gdm_circle_match_visbits_hip(const uint8_t* vis, int B, int M, uint32_t* bits, void* stream)
{
    GDM_CHECK_ARG(vis && bits && B >= 1 && M >= 1, "gdm_circle_match_visbits_hip: NULL pointer or bad shape");
    const int W = (M + 31) / 32;
    hipLaunchKernelGGL(cm_visbits_kernel, dim3(gdm_cdiv((long)B * W, 256)), dim3(256), 0, (hipStream_t)stream, vis, B, M, W, bits);
    return gdm_launch_status("cm_visbits_kernel");
}

static int fill_args(CmArgs& a, const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp, int R, int M,
                     const int32_t* g, const int32_t* c2, const int32_t* item, const uint32_t* nbr, const uint32_t* visb,
                     float gamma, float m, const char* who)
{
    GDM_CHECK_ARG(xrows && xtp && xsum && yrows && ytp && g && item, "%s: NULL pointer", who);
    GDM_CHECK_ARG(R >= 1 && M >= 1, "%s: R=%d M=%d", who, R, M);
    GDM_CHECK_ARG(c2 || (nbr && visb), "%s: either the symmetric columns (c2) or the neighbour / visibility bit tables are needed", who);
    GDM_CHECK_ARG(gamma > 0.f && m >= 0.f && m < 1.f && gamma * (2.f + m) * (2.f - m) < 150.f, "%s: gamma=%g m=%g outside the fp32 exp range", who, gamma, m);
    a.xrows = (const unsigned char*)xrows; a.xtp = (const unsigned char*)xtp; a.xsum = xsum;
    a.yrows = (const unsigned char*)yrows; a.ytp = (const unsigned char*)ytp;
    a.g = g; a.c2 = c2; a.item = item; a.nbr = nbr; a.visb = visb;
    a.R = R; a.Rp = (R + 127) / 128 * 128; a.M = M; a.Mp = (M + 127) / 128 * 128; a.W = (M + 31) / 32; a.P = 1;
    a.gamma = gamma; a.m = m;
    a.offp = 0.5f * gamma * (2.f + m) * (2.f - m);              // logits of positives span [-0.04 gamma .., gamma (2+m)(2-m)], negatives
    a.offn = 0.5f * gamma * (1.f + m) * (1.f - m);              // [.., gamma (1+m)(1-m)]: centred, exp() stays far inside fp32
    a.lse_p = a.lse_n = a.loss = nullptr; a.coef = nullptr; a.gout = nullptr;
    a.nbr_istride = 0; a.pad_e0 = 0;
    return 0;
}

// Forward.  All per-row arrays have Rp = round_up(R, 128) entries (g = M, item = 0 for the padding rows).
extern "C" int gdm_circle_match_fwd2_hip(const void* xrows, const void* xtp, const float* xpad, const void* yrows, const void* ytp,
                                         int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                                         const uint32_t* nbr, int nbr_per_item, const uint32_t* visb, int pad_e0, float gamma, float m,
                                         float* lse_p, float* lse_n, float* loss, void* stream)
{
    CmArgs a;
    int rc = fill_args(a, xrows, xtp, xpad, yrows, ytp, R, M, g, c2, item, nbr, visb, gamma, m, "gdm_circle_match_fwd_hip");
    if (rc) return rc;
    GDM_CHECK_ARG(lse_p && lse_n && loss, "gdm_circle_match_fwd_hip: NULL output");
    a.lse_p = lse_p; a.lse_n = lse_n; a.loss = loss;
    a.nbr_istride = nbr_per_item ? (long)M * a.W : 0;
    a.pad_e0 = pad_e0 ? 1 : 0;
    return launch_mode<0>(a, c2 != nullptr, (hipStream_t)stream);
}

extern "C" int gdm_circle_match_fwd_hip(const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp,
                                        int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                                        const uint32_t* nbr, const uint32_t* visb, float gamma, float m,
                                        float* lse_p, float* lse_n, float* loss, void* stream)
{
    return gdm_circle_match_fwd2_hip(xrows, xtp, xsum, yrows, ytp, R, M, g, c2, item, nbr, 0, visb, 0, gamma, m, lse_p, lse_n, loss, stream);
}

extern "C" int gdm_circle_match_bwd_parts(int R, int M)
{
    if (R < 1 || M < 1) return 0;
    const int vb = (M + 127) / 128, nst = (R + 127) / 128 * 128 / CM_ST;
    int P = (768 + vb - 1) / vb;                                   // ~3 workgroups per CU in flight
    if (P > nst) P = nst;
    return P < 1 ? 1 : P;
}

// Backward: gx f32[Rp,128] and gy_part f32[P][Mp,128] (P = gdm_circle_match_bwd_parts; the caller sums over P).
extern "C" int gdm_circle_match_bwd2_hip(const void* xrows, const void* xtp, const float* xpad, const void* yrows, const void* ytp,
                                         int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                                         const uint32_t* nbr, int nbr_per_item, const uint32_t* visb, int pad_e0, float gamma, float m,
                                         const float* lse_p, const float* lse_n, const float* coef, float* gx, float* gy_part, void* stream);

extern "C" int gdm_circle_match_bwd_hip(const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp,
                                        int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                                        const uint32_t* nbr, const uint32_t* visb, float gamma, float m,
                                        const float* lse_p, const float* lse_n, const float* coef, float* gx, float* gy_part, void* stream)
{
    return gdm_circle_match_bwd2_hip(xrows, xtp, xsum, yrows, ytp, R, M, g, c2, item, nbr, 0, visb, 0, gamma, m, lse_p, lse_n, coef, gx, gy_part,
                                     stream);
}

extern "C" int gdm_circle_match_bwd2_hip(const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp,
                                         int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                                         const uint32_t* nbr, int nbr_per_item, const uint32_t* visb, int pad_e0, float gamma, float m,
                                         const float* lse_p, const float* lse_n, const float* coef, float* gx, float* gy_part, void* stream)
{
    CmArgs a;
    int rc = fill_args(a, xrows, xtp, xsum, yrows, ytp, R, M, g, c2, item, nbr, visb, gamma, m, "gdm_circle_match_bwd_hip");
    if (rc) return rc;
    a.nbr_istride = nbr_per_item ? (long)M * a.W : 0;
    a.pad_e0 = pad_e0 ? 1 : 0;
    GDM_CHECK_ARG(lse_p && lse_n && coef && gx && gy_part, "gdm_circle_match_bwd_hip: NULL pointer");
    a.lse_p = const_cast<float*>(lse_p); a.lse_n = const_cast<float*>(lse_n); a.coef = coef;
    a.gout = gx;
    if ((rc = launch_mode<1>(a, c2 != nullptr, (hipStream_t)stream))) return rc;
    a.gout = gy_part;
    a.P = gdm_circle_match_bwd_parts(R, M);
    return launch_mode<2>(a, c2 != nullptr, (hipStream_t)stream);
}
