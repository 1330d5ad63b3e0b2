// N x M descriptor matching for gfx950 (MI355X): cosine similarity + row arg-max, optionally
// the materialised similarity matrix.
//
// Replaces (reference, /root/reference):
//   evaluator.py:87-93    F.normalize(rows) ; F.normalize(mesh, dim=0) ; matmul ; max(dim=1)
//   models/geoMatch.py:117-136 uses the same normalise + matmul with the (M+1)-column padded mesh
//
// Two kernels:
//   1. pack_rows_kernel   [R, D=128, n] channel-major fp32  ->  R*n packed rows of 512 B:
//        L2-normalise over D (x / max(|x|, 1e-12), F.normalize semantics) and store either
//          BF16X3: 128 bf16 "hi" | 128 bf16 "lo"   (x = hi + lo + O(2^-18 |x|))
//          F32   : 128 fp32
//        Transposes through LDS so that global reads are contiguous along n and global writes
//        are contiguous 512-B rows.
//   2. match_kernel<PREC, WRITE_SIM>  one workgroup = 4 waves = 128 scene rows; each wave keeps
//        its 32 rows' whole K=128 operand in registers (64 VGPRs) for the life of the kernel and
//        streams 64-column model tiles through a double-buffered, XOR-swizzled LDS image
//        (global -> registers before the MFMAs, registers -> LDS after them: issue-early /
//        write-late).  Products run on the matrix cores:
//          BF16X3: v_mfma_f32_32x32x16_bf16, three per k-step: hi*hi + hi*lo + lo*hi
//          F32   : v_mfma_f32_32x32x2_f32 (exact fp32 products, one rounding per FMA)
//        The epilogue keeps a running (max, first arg-max) per row in registers, reduces it
//        across the 32 lanes of a half-wave with shuffles, and (WRITE_SIM) stores the tile.
//        The M axis can be split over blockIdx.y; a small kernel merges the partial maxima.
//
// Roofline: materialised form moves 4D(BN+M) + 4BNM bytes; fused form 4D(BN+M)+8BN.
#include "gdm_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int D = 128;                 // descriptor length (config/lmo_cfg.py:125 feat_dim)
constexpr int ROW_BYTES = 512;         // one packed row
constexpr int MT_ROWS = 128;           // scene rows per workgroup
constexpr int MT_COLS = 64;            // model columns per LDS tile
constexpr int TILE_BYTES = MT_COLS * ROW_BYTES;   // 32 KiB

// ------------------------------------------------------------------------------------------
// pack: [R, 128, n] -> R*n rows
// ------------------------------------------------------------------------------------------

// one block: 64 points (from p0) of row set r
template <int PREC>
__device__ __forceinline__ void pack_rows_tile(const float* __restrict__ x, int n, unsigned char* __restrict__ out, const int r, const int p0,
                                               float (*t)[65], float (*part)[64])
{
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int p = min(p0 + lane, n - 1);
    const float* xr = x + (long)r * D * n;
    float ss = 0.f;
    for (int c = w; c < D; c += 4) {
        const float v = xr[(long)c * n + p];
        t[c][lane] = v;
        ss += v * v;
    }
    part[w][lane] = ss;
    __syncthreads();
    // every thread recomputes the norm of the points it will write
    for (int it = 0; it < 4; ++it) {
        const int item = it * 256 + threadIdx.x;       // 64 points x 16 chunks of 8 channels
        const int ch = item & 15;
        const int pl = item >> 4;
        if (p0 + pl >= n) continue;
        const float nrm = sqrtf((part[0][pl] + part[1][pl]) + (part[2][pl] + part[3][pl]));
        const float inv_den = fmaxf(nrm, 1e-12f);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t[ch * 8 + j][pl] / inv_den;
        unsigned char* row = out + ((long)r * n + p0 + pl) * ROW_BYTES;
        if (PREC == GDM_MATCH_BF16X3) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gdm_split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
            }
            *reinterpret_cast<uint4*>(row + ch * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(row + 256 + ch * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        } else {
            *reinterpret_cast<float4*>(row + ch * 32) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(row + ch * 32 + 16) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ x, int n, unsigned char* __restrict__ out)
{
    __shared__ float t[D][65];
    __shared__ float part[4][64];
    pack_rows_tile<PREC>(x, n, out, blockIdx.y, blockIdx.x * 64, t, part);
}

// two independent packs in one launch (the scene descriptors [R1, 128, n1] and the model descriptors [R2, 128, n2] of one step):
// blockIdx.y < R1 walks the first, the rest the second; the same tile function, so the same bytes as two launches
template <int PREC>
__global__ __launch_bounds__(256) void pack_rows2_kernel(const float* __restrict__ x1, int R1, int n1, unsigned char* __restrict__ out1,
                                                         const float* __restrict__ x2, int n2, unsigned char* __restrict__ out2)
{
    __shared__ float t[D][65];
    __shared__ float part[4][64];
    const int p0 = blockIdx.x * 64;
    const int y = blockIdx.y;
    if (y < R1) {
        if (p0 < n1) pack_rows_tile<PREC>(x1, n1, out1, y, p0, t, part);
    } else {
        if (p0 < n2) pack_rows_tile<PREC>(x2, n2, out2, y - R1, p0, t, part);
    }
}

// ------------------------------------------------------------------------------------------
// match
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_chunk_off(int col, int ch)
{
    // 512-B rows; XOR the low four bits of the 16-B chunk index with the column, so that the 16
    // lanes of every ds_read_b128 group (distinct columns mod 16, same logical chunk) fall on 16
    // different 16-B slots of the 256-B bank row.
    return col * ROW_BYTES + (((ch & 16) | ((ch ^ col) & 15)) << 4);
}

template <int PREC, bool WRITE_SIM>
__global__ __launch_bounds__(256) void match_kernel(const unsigned char* __restrict__ apk,   // [R] packed scene rows
                                                    const unsigned char* __restrict__ bpk,   // [M] packed model rows
                                                    int R, int M, int cols_per_split,
                                                    float* __restrict__ sim,                  // [R, M] or NULL
                                                    float* __restrict__ pval,                 // [splits, R]
                                                    int32_t* __restrict__ pidx)               // [splits, R]
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 31;        // row (A) / column (B) within a 32-block
    const int h = lane >> 5;         // k half

    const int row0 = blockIdx.x * MT_ROWS + wave * 32;
    const int split = blockIdx.y;
    const int col_begin = split * cols_per_split;
    const int col_end = min(col_begin + cols_per_split, M);
    const int ntiles = (col_end - col_begin + MT_COLS - 1) / MT_COLS;

    // ---- A operand: this lane's 16 chunks, resident for the whole kernel ----
    u32x4 areg[16];
    {
        const unsigned char* arow = apk + (long)min(row0 + lr, R - 1) * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int ch;
            if (PREC == GDM_MATCH_BF16X3) ch = (i < 8) ? (2 * i + h) : (16 + 2 * (i - 8) + h);   // hi[0..7], lo[0..7]
            else ch = 16 * h + i;                                                                // k = 64h + 4i..4i+3
            areg[i] = *reinterpret_cast<const u32x4*>(arow + ch * 16);
        }
    }

    float best[16];
    int bidx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        best[i] = -INFINITY;
        bidx[i] = 0;
    }

    // ---- tile staging: 2048 chunks per tile, 8 per thread ----
    u32x4 stage[8];
    auto stage_load = [&](int tile) {
        const int c0 = col_begin + tile * MT_COLS;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = i * 256 + tid;
            const int col = g >> 5, ch = g & 31;
            const int gc = min(c0 + col, M - 1);
            stage[i] = *reinterpret_cast<const u32x4*>(bpk + (long)gc * ROW_BYTES + ch * 16);
        }
    };
    auto stage_store = [&](int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = i * 256 + tid;
            const int col = g >> 5, ch = g & 31;
            *reinterpret_cast<u32x4*>(base + lds_chunk_off(col, ch)) = stage[i];
        }
    };

    if (ntiles > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const bool has_next = tile + 1 < ntiles;
        if (has_next) stage_load(tile + 1);                 // global loads in flight under the MFMAs

        const unsigned char* base = smem + buf * TILE_BYTES;
        const int tcol0 = col_begin + tile * MT_COLS;

#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int col = cb * 32 + lr;
            if (PREC == GDM_MATCH_BF16X3) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(base + lds_chunk_off(col, 2 * s + h));
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(base + lds_chunk_off(col, 16 + 2 * s + h));
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, areg[s]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, areg[8 + s]);
                    const bf16x8 vbh = __builtin_bit_cast(bf16x8, bh);
                    const bf16x8 vbl = __builtin_bit_cast(bf16x8, bl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vbl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vbh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vbh, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32x4 bv = *reinterpret_cast<const u32x4*>(base + lds_chunk_off(col, 16 * h + i));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].x), __uint_as_float(bv.x), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].y), __uint_as_float(bv.y), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].z), __uint_as_float(bv.z), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].w), __uint_as_float(bv.w), acc, 0, 0, 0);
                }
            }
            // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
            const int gcol = tcol0 + col;
            const bool col_ok = gcol < col_end;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = acc[reg];
                if (col_ok && v > best[reg]) {              // strict: first maximum per lane (ascending columns)
                    best[reg] = v;
                    bidx[reg] = gcol;
                }
                if (WRITE_SIM) {
                    const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (col_ok && grow < R) sim[(long)grow * M + gcol] = v;
                }
            }
        }

        if (has_next) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- reduce (max, lowest arg) across the 32 lanes of each half-wave ----
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float v = best[reg];
        int ix = bidx[reg];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const float ov = __shfl_xor(v, m, 64);
            const int oi = __shfl_xor(ix, m, 64);
            if (ov > v || (ov == v && oi < ix)) {
                v = ov;
                ix = oi;
            }
        }
        if (lr == 0) {
            const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (grow < R) {
                pval[(long)split * R + grow] = v;
                pidx[(long)split * R + grow] = ix;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// match v2: model PANEL resident in LDS, scene row-blocks streamed through registers
// ------------------------------------------------------------------------------------------
// One workgroup = 8 waves (2 per SIMD) owns a panel of 256 model columns (128 KiB of packed rows, the
// whole LDS budget of a CU but for a sliver) for its lifetime and walks over row-blocks of 256 scene rows
// (32 per wave).  After the one-time panel fill there is NO barrier and no LDS write in the loop: a wave
// loads its 32 rows' operand (16 x 16 B per lane, straight to registers, next block prefetched), runs
// 8 column blocks x 24 (BF16X3) MFMAs against fragments read from the swizzled panel, and stores / arg-max
// reduces its own 32 x 256 outputs.  Tile stores of one wave overlap the MFMAs of its SIMD partner.
// Grid: blockIdx.x -> (g = bid % G, panel = bid / G): workgroups that read the same scene rows share
// `bid % 8`, i.e. (observed, speed only) one XCD's L2.
constexpr int PANEL_COLS = 256;
constexpr int PANEL_BYTES = PANEL_COLS * ROW_BYTES;     // 128 KiB
constexpr int V2_THREADS = 512;
constexpr int V2_ROWS = 256;                            // scene rows per workgroup iteration

template <int PREC>
__device__ __forceinline__ void load_a_rows(const unsigned char* __restrict__ apk, int row, int R, int h, u32x4 (&a)[16])
{
    const unsigned char* arow = apk + (long)min(row, R - 1) * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int ch;
        if (PREC == GDM_MATCH_BF16X3) ch = (i < 8) ? (2 * i + h) : (16 + 2 * (i - 8) + h);
        else ch = 16 * h + i;
        a[i] = *reinterpret_cast<const u32x4*>(arow + ch * 16);
    }
}

template <int PREC, bool WRITE_SIM>
__global__ __launch_bounds__(V2_THREADS) void match_panel_kernel(const unsigned char* __restrict__ apk,
                                                                  const unsigned char* __restrict__ bpk,
                                                                  int R, int M, int G,
                                                                  float* __restrict__ sim,
                                                                  float* __restrict__ pval,      // [panels, R]
                                                                  int32_t* __restrict__ pidx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // PANEL_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 31;
    const int h = lane >> 5;
    const int g = blockIdx.x % G;
    const int panel = blockIdx.x / G;
    const int col0 = panel * PANEL_COLS;
    const int ncols = min(PANEL_COLS, M - col0);

    // ---- one-time panel fill: 256 columns x 32 chunks, 16 chunks per thread, coalesced 16-B loads ----
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int gi = i * V2_THREADS + tid;
        const int col = gi >> 5, ch = gi & 31;
        const int gc = min(col0 + col, M - 1);
        const u32x4 v = *reinterpret_cast<const u32x4*>(bpk + (long)gc * ROW_BYTES + ch * 16);
        *reinterpret_cast<u32x4*>(smem + lds_chunk_off(col, ch)) = v;
    }
    __syncthreads();

    const int nrb = (R + V2_ROWS - 1) / V2_ROWS;
    u32x4 anext[16];
    if (g < nrb) load_a_rows<PREC>(apk, g * V2_ROWS + wave * 32 + lr, R, h, anext);

    for (int rb = g; rb < nrb; rb += G) {
        u32x4 areg[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) areg[i] = anext[i];
        if (rb + G < nrb) load_a_rows<PREC>(apk, (rb + G) * V2_ROWS + wave * 32 + lr, R, h, anext);
        const int row0 = rb * V2_ROWS + wave * 32;

        float best[16];
        int bidx[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            best[i] = -INFINITY;
            bidx[i] = 0;
        }

#pragma unroll 1
        for (int cp = 0; cp < PANEL_COLS / 64; ++cp) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0[i] = 0.f;
                acc1[i] = 0.f;
            }
            const int c0 = cp * 64 + lr, c1 = c0 + 32;
            if (PREC == GDM_MATCH_BF16X3) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bf16x8 bh0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0, 2 * s + h)));
                    const bf16x8 bl0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0, 16 + 2 * s + h)));
                    const bf16x8 bh1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c1, 2 * s + h)));
                    const bf16x8 bl1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c1, 16 + 2 * s + h)));
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, areg[s]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, areg[8 + s]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32x4 b0 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0, 16 * h + i));
                    const u32x4 b1 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c1, 16 * h + i));
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].x), __uint_as_float(b0.x), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].x), __uint_as_float(b1.x), acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].y), __uint_as_float(b0.y), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].y), __uint_as_float(b1.y), acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].z), __uint_as_float(b0.z), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].z), __uint_as_float(b1.z), acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].w), __uint_as_float(b0.w), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[i].w), __uint_as_float(b1.w), acc1, 0, 0, 0);
                }
            }
            const int gc0 = col0 + c0, gc1 = col0 + c1;
            const bool ok0 = c0 < ncols, ok1 = c1 < ncols;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v0 = acc0[reg], v1 = acc1[reg];
                if (ok0 && v0 > best[reg]) {
                    best[reg] = v0;
                    bidx[reg] = gc0;
                }
                if (ok1 && v1 > best[reg]) {
                    best[reg] = v1;
                    bidx[reg] = gc1;
                }
                if (WRITE_SIM) {
                    const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (grow < R) {
                        float* o = sim + (long)grow * M;
                        if (ok0) __builtin_nontemporal_store(v0, o + gc0);
                        if (ok1) __builtin_nontemporal_store(v1, o + gc1);
                    }
                }
            }
        }

        // ---- (max, lowest arg) across the 32 lanes of each half-wave, halving the row set every step:
        //      after step t each lane carries 16>>(t+1) rows, so 16+8+4+2 (+2) shuffles instead of 160 ----
        float v8[8];  int i8[8];
        {
            const bool up = lane & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sv = up ? best[i] : best[i + 8];
                const int si = up ? bidx[i] : bidx[i + 8];
                const float ov = __shfl_xor(sv, 1, 64);
                const int oi = __shfl_xor(si, 1, 64);
                const float mv = up ? best[i + 8] : best[i];
                const int mi = up ? bidx[i + 8] : bidx[i];
                const bool take = ov > mv || (ov == mv && oi < mi);
                v8[i] = take ? ov : mv;
                i8[i] = take ? oi : mi;
            }
        }
        float v4[4];  int i4[4];
        {
            const bool up = lane & 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sv = up ? v8[i] : v8[i + 4];
                const int si = up ? i8[i] : i8[i + 4];
                const float ov = __shfl_xor(sv, 2, 64);
                const int oi = __shfl_xor(si, 2, 64);
                const float mv = up ? v8[i + 4] : v8[i];
                const int mi = up ? i8[i + 4] : i8[i];
                const bool take = ov > mv || (ov == mv && oi < mi);
                v4[i] = take ? ov : mv;
                i4[i] = take ? oi : mi;
            }
        }
        float v2[2];  int i2[2];
        {
            const bool up = lane & 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float sv = up ? v4[i] : v4[i + 2];
                const int si = up ? i4[i] : i4[i + 2];
                const float ov = __shfl_xor(sv, 4, 64);
                const int oi = __shfl_xor(si, 4, 64);
                const float mv = up ? v4[i + 2] : v4[i];
                const int mi = up ? i4[i + 2] : i4[i];
                const bool take = ov > mv || (ov == mv && oi < mi);
                v2[i] = take ? ov : mv;
                i2[i] = take ? oi : mi;
            }
        }
        float v1;  int i1;
        {
            const bool up = lane & 8;
            const float sv = up ? v2[0] : v2[1];
            const int si = up ? i2[0] : i2[1];
            const float ov = __shfl_xor(sv, 8, 64);
            const int oi = __shfl_xor(si, 8, 64);
            const float mv = up ? v2[1] : v2[0];
            const int mi = up ? i2[1] : i2[0];
            const bool take = ov > mv || (ov == mv && oi < mi);
            v1 = take ? ov : mv;
            i1 = take ? oi : mi;
        }
        {
            const float ov = __shfl_xor(v1, 16, 64);
            const int oi = __shfl_xor(i1, 16, 64);
            if (ov > v1 || (ov == v1 && oi < i1)) {
                v1 = ov;
                i1 = oi;
            }
        }
        // lane bits 0..3 chose the upper half of the remaining row set at steps 1..4:
        // register index reg = 8*b0 + 4*b1 + 2*b2 + b3
        if ((lane & 16) == 0) {
            const int reg = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
            const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (grow < R) {
                pval[(long)panel * R + grow] = v1;
                pidx[(long)panel * R + grow] = i1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// match v3: v2's LDS-resident panel, software-pipelined inside the wave (BF16X3, full tiles only)
// ------------------------------------------------------------------------------------------
// PMC on v2 (profiles/r01_match_pmc.md): the MFMA pipe is busy 65 % (fused) / 39 % (materialised) of the SIMD's cycles; the two
// waves of a SIMD fall into phase (both issue MFMAs, then both run their ~130-instruction arg-max/store epilogue), so the pipe
// idles during every epilogue.  v3 removes the dependence on the partner wave:
//   * a wave's step t issues the 48 MFMAs of 64-column block t into one accumulator pair WHILE it drains block t-1 from the
//     other pair (running arg-max, stores), with the fragment reads of the next k-step (and of the next block's first k-step)
//     in flight; sched_group_barrier pins the interleave MFMA : ds_read : VALU : store, one scheduling region per step;
//   * the MFMA operand roles are swapped (srcA = model columns from the LDS panel, srcB = scene rows from registers), so a
//     lane holds 16 columns of ONE scene row per accumulator: the running (max, arg) is a single register pair per lane, the
//     row-block reduction is one shuffle, and the tile goes out as dwordx4 stores (4 consecutive columns per lane; plain
//     stores, merged to full lines in L2 -- tools/micro/store_pattern.hip: 5.95 TB/s for this pattern, non-temporal 1.8);
//   * the next row block's scene operand is prefetched into a second register set at the start of the row block, i.e. ahead of
//     that row block's stores in vmcnt order (loads issued behind ~100 stores wait for all of them).
// Requires R % 256 == 0 and M % 256 == 0 (no per-lane predication, so every step is one basic block); other shapes run v2.
// Same products, same accumulation order as v2: bit-identical results.
struct BFrag { u32x4 h0, l0, h1, l1; };

__device__ __forceinline__ BFrag read_bfrag(const unsigned char* smem, int c0, int s, int h)
{
    BFrag f;
    f.h0 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0, 2 * s + h));
    f.l0 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0, 16 + 2 * s + h));
    f.h1 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0 + 32, 2 * s + h));
    f.l1 = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c0 + 32, 16 + 2 * s + h));
    return f;
}

// (max, first arg-max) of the 32 values a lane holds for its row in one 64-column block, as a tournament: codes are
// compile-time constants at the leaves, every node takes its right child only if strictly greater (lower column wins ties).
// code = 16*acc + reg; column within the block = (reg&3) + 8*(reg>>2) + 32*acc (+ 4*h), ascending in code for a fixed lane.
__device__ __forceinline__ void block_argmax(const f32x16& p0, const f32x16& p1, float& bm, int& bc)
{
    // four independent scan chains over ascending code ranges (8 values each), merged left to right
    float m[4];
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x16& p = (j < 2) ? p0 : p1;
        const int r0 = (j & 1) * 8;
        m[j] = p[r0];
        c[j] = j * 8;
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            const float v = p[r0 + i];
            const bool t = v > m[j];
            m[j] = t ? v : m[j];
            c[j] = t ? j * 8 + i : c[j];
        }
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const bool t = m[j] > m[0];
        m[0] = t ? m[j] : m[0];
        c[0] = t ? c[j] : c[0];
    }
    bm = m[0];
    bc = c[0];
}

// One pipeline step: MFMAs of column block CP into (n0, n1) | drain (p0, p1) = column block PCP of row `prow`.
//   fr        in: fragments of (CP, k-step 0); out: fragments of ((CP+1)%4, k-step 0)
//   PREFETCH  : after k-step s, reload areg[s], areg[8+s] with the NEXT row block's operand (their last use in this row block)
//   DRAIN     : false only for the very first step of a workgroup (nothing to drain yet)
// acc[reg] of lane (lr, h) = sim[row lr][column (reg&3) + 8*(reg>>2) + 4*h of the 32-column block]
template <int CP, int PCP, bool PREFETCH, bool DRAIN>
__device__ __forceinline__ void pipe_step(const unsigned char* smem, int lr, int h, u32x4 (&areg)[16],
                                          BFrag& fr, f32x16& n0, f32x16& n1, const f32x16& p0, const f32x16& p1,
                                          float& best, int& bcode, const unsigned char* __restrict__ arow_next)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        n0[i] = 0.f;
        n1[i] = 0.f;
    }
    const int c0 = CP * 64 + lr;
    const int c0n = ((CP + 1) & 3) * 64 + lr;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const BFrag nx = (s < 7) ? read_bfrag(smem, c0, s + 1, h) : read_bfrag(smem, c0n, 0, h);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, areg[s]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, areg[8 + s]);
        const bf16x8 bh0 = __builtin_bit_cast(bf16x8, fr.h0), bl0 = __builtin_bit_cast(bf16x8, fr.l0);
        const bf16x8 bh1 = __builtin_bit_cast(bf16x8, fr.h1), bl1 = __builtin_bit_cast(bf16x8, fr.l1);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl0, ah, n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl1, ah, n1, 0, 0, 0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, al, n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, al, n1, 0, 0, 0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah, n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah, n1, 0, 0, 0);
        if (PREFETCH) {
            areg[s] = *reinterpret_cast<const u32x4*>(arow_next + (2 * s + h) * 16);
            areg[8 + s] = *reinterpret_cast<const u32x4*>(arow_next + (16 + 2 * s + h) * 16);
        }
        fr = nx;
    }
    if (DRAIN) {
        float lbest;
        int lcode;
        block_argmax(p0, p1, lbest, lcode);
        const bool take = lbest > best;                    // earlier blocks hold lower columns: strict '>' keeps the first maximum
        best = take ? lbest : best;
        bcode = take ? lcode + PCP * 32 : bcode;
    }
    // interleave: every MFMA is followed by a fragment read and a slice of the epilogue (one pipeline = one sync id per step)
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, CP);
        if (i < 32) __builtin_amdgcn_sched_group_barrier(0x100, 1, CP);
        if (PREFETCH && (i % 6) == 5) __builtin_amdgcn_sched_group_barrier(0x020, 2, CP);
        if (DRAIN) __builtin_amdgcn_sched_group_barrier(0x002, 2, CP);
    }
    __builtin_amdgcn_sched_barrier(0);          // one scheduling region per step: the interleave pipeline is matched per region
}

__global__ __launch_bounds__(V2_THREADS) void match_pipe_kernel(const unsigned char* __restrict__ apk,
                                                                 const unsigned char* __restrict__ bpk,
                                                                 int R, int M, int G,
                                                                 float* __restrict__ sim,
                                                                 float* __restrict__ pval,      // [panels, R]
                                                                 int32_t* __restrict__ pidx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // PANEL_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31;
    const int h = lane >> 5;
    const int g = blockIdx.x % G;
    const int panel = blockIdx.x / G;
    const int col0 = panel * PANEL_COLS;

#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int gi = i * V2_THREADS + tid;
        const int col = gi >> 5, ch = gi & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(bpk + (long)(col0 + col) * ROW_BYTES + ch * 16);
        *reinterpret_cast<u32x4*>(smem + lds_chunk_off(col, ch)) = v;
    }
    __syncthreads();

    const int nrb = R / V2_ROWS;
    if (g >= nrb) return;
    u32x4 areg[16];
    {
        const unsigned char* arow = apk + (long)(g * V2_ROWS + wave * 32 + lr) * ROW_BYTES;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            areg[s] = *reinterpret_cast<const u32x4*>(arow + (2 * s + h) * 16);
            areg[8 + s] = *reinterpret_cast<const u32x4*>(arow + (16 + 2 * s + h) * 16);
        }
    }
    float best = -INFINITY;
    int bcode = 0;
    f32x16 a0, a1, b0, b1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        b0[i] = 0.f;
        b1[i] = 0.f;
    }
    BFrag fr = read_bfrag(smem, lr, 0, h);

    // step 0 of the first row block: nothing to drain
    pipe_step<0, 0, false, false>(smem, lr, h, areg, fr, a0, a1, b0, b1, best, bcode, apk);

    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * V2_ROWS + wave * 32;                        // uniform
        const bool has_next = rb + G < nrb;
        const int nrow = (has_next ? (rb + G) * V2_ROWS : rb * V2_ROWS) + wave * 32 + lr;
        const unsigned char* arow_next = apk + (long)nrow * ROW_BYTES;
        pipe_step<1, 0, false, true>(smem, lr, h, areg, fr, b0, b1, a0, a1, best, bcode, arow_next);
        pipe_step<2, 1, false, true>(smem, lr, h, areg, fr, a0, a1, b0, b1, best, bcode, arow_next);
        pipe_step<3, 2, true, true>(smem, lr, h, areg, fr, b0, b1, a0, a1, best, bcode, arow_next);
        if (has_next) {
            pipe_step<0, 3, false, true>(smem, lr, h, areg, fr, a0, a1, b0, b1, best, bcode, arow_next);
        } else {
            float lbest;
            int lcode;
            block_argmax(b0, b1, lbest, lcode);
            const bool take = lbest > best;
            best = take ? lbest : best;
            bcode = take ? lcode + 96 : bcode;
        }
        // row-block result: code -> column, merge the two half-waves (lower column wins ties), one lane per row writes
        {
            int ix = col0 + ((bcode & 3) | (((bcode >> 2) & 3) << 3) | ((bcode >> 4) << 5)) + 4 * h;
            float v = best;
            const float ov = __shfl_xor(v, 32, 64);
            const int oi = __shfl_xor(ix, 32, 64);
            if (ov > v || (ov == v && oi < ix)) {
                v = ov;
                ix = oi;
            }
            if (h == 0) {
                pval[(long)panel * R + row0 + lr] = v;
                pidx[(long)panel * R + row0 + lr] = ix;
            }
        }
        best = -INFINITY;
        bcode = 0;
    }
}

__device__ __forceinline__ void finalize_rows(const float (&best)[16], const int (&bidx)[16], int lane, int h, int row0, int R,
                                              int panel, float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    // (max, lowest arg) across the 32 lanes of each half-wave, halving the row set every step (as v2)
    float v8[8];  int i8[8];
    {
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sv = up ? best[i] : best[i + 8];
            const int si = up ? bidx[i] : bidx[i + 8];
            const float ov = __shfl_xor(sv, 1, 64);
            const int oi = __shfl_xor(si, 1, 64);
            const float mv = up ? best[i + 8] : best[i];
            const int mi = up ? bidx[i + 8] : bidx[i];
            const bool take = ov > mv || (ov == mv && oi < mi);
            v8[i] = take ? ov : mv;
            i8[i] = take ? oi : mi;
        }
    }
    float v4[4];  int i4[4];
    {
        const bool up = lane & 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sv = up ? v8[i] : v8[i + 4];
            const int si = up ? i8[i] : i8[i + 4];
            const float ov = __shfl_xor(sv, 2, 64);
            const int oi = __shfl_xor(si, 2, 64);
            const float mv = up ? v8[i + 4] : v8[i];
            const int mi = up ? i8[i + 4] : i8[i];
            const bool take = ov > mv || (ov == mv && oi < mi);
            v4[i] = take ? ov : mv;
            i4[i] = take ? oi : mi;
        }
    }
    float v2[2];  int i2[2];
    {
        const bool up = lane & 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sv = up ? v4[i] : v4[i + 2];
            const int si = up ? i4[i] : i4[i + 2];
            const float ov = __shfl_xor(sv, 4, 64);
            const int oi = __shfl_xor(si, 4, 64);
            const float mv = up ? v4[i + 2] : v4[i];
            const int mi = up ? i4[i + 2] : i4[i];
            const bool take = ov > mv || (ov == mv && oi < mi);
            v2[i] = take ? ov : mv;
            i2[i] = take ? oi : mi;
        }
    }
    float v1;  int i1;
    {
        const bool up = lane & 8;
        const float sv = up ? v2[0] : v2[1];
        const int si = up ? i2[0] : i2[1];
        const float ov = __shfl_xor(sv, 8, 64);
        const int oi = __shfl_xor(si, 8, 64);
        const float mv = up ? v2[1] : v2[0];
        const int mi = up ? i2[1] : i2[0];
        const bool take = ov > mv || (ov == mv && oi < mi);
        v1 = take ? ov : mv;
        i1 = take ? oi : mi;
    }
    {
        const float ov = __shfl_xor(v1, 16, 64);
        const int oi = __shfl_xor(i1, 16, 64);
        if (ov > v1 || (ov == v1 && oi < i1)) {
            v1 = ov;
            i1 = oi;
        }
    }
    if ((lane & 16) == 0) {
        const int reg = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
        const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        pval[(long)panel * R + grow] = v1;
        pidx[(long)panel * R + grow] = i1;
    }
}

// Materialised form of v3: original operand roles (lane = column, register = row), so one dword store instruction covers
// 2 rows x 128 contiguous bytes and may stay non-temporal; the swapped layout's 16-byte pieces need write-back merging in L2
// and lose under load (347 vs 256 us).  Running (max, arg) per register, v2's butterfly per row block.
// Steps of 32 columns (one accumulator in flight, one draining): 24 MFMAs | 16 values drained per step.
#ifndef GDM_MATCH_EXP
#define GDM_MATCH_EXP 0     // development: 1 = no operand reload, 2 = no matrix stores (wrong results)
#endif
struct BFrag1 { u32x4 h, l; };

__device__ __forceinline__ BFrag1 read_bfrag1(const unsigned char* smem, int c, int s, int h)
{
    BFrag1 f;
    f.h = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c, 2 * s + h));
    f.l = *reinterpret_cast<const u32x4*>(smem + lds_chunk_off(c, 16 + 2 * s + h));
    return f;
}

template <int CP, int PCP, bool RELOAD, bool DRAIN>
__device__ __forceinline__ void pipe_step_sim(const unsigned char* smem, int lr, int h, u32x4 (&areg)[16], BFrag1& fr,
                                              f32x16& n, const f32x16& p, float (&best)[16], int (&bidx)[16], int pgc,
                                              float* __restrict__ pbase, unsigned pvoff, long M,
                                              const unsigned char* __restrict__ arow_next)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) n[i] = 0.f;
    const int c = CP * 32 + lr;
    const int cn = ((CP + 1) & 7) * 32 + lr;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const BFrag1 nx = (s < 7) ? read_bfrag1(smem, c, s + 1, h) : read_bfrag1(smem, cn, 0, h);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, areg[s]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, areg[8 + s]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, fr.h), bl = __builtin_bit_cast(bf16x8, fr.l);
        n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, n, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, n, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, n, 0, 0, 0);
        if (RELOAD && !(GDM_MATCH_EXP & 1)) {
            areg[s] = *reinterpret_cast<const u32x4*>(arow_next + (2 * s + h) * 16);
            areg[8 + s] = *reinterpret_cast<const u32x4*>(arow_next + (16 + 2 * s + h) * 16);
        }
        fr = nx;
    }
    if (DRAIN) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float v = p[reg];
            const bool t = v > best[reg];
            best[reg] = t ? v : best[reg];
            bidx[reg] = t ? pgc : bidx[reg];
            float* o = pbase + (long)((reg & 3) + 8 * (reg >> 2)) * M + PCP * 32;     // uniform row base, per-lane 32-bit offset
            if (!(GDM_MATCH_EXP & 2)) __builtin_nontemporal_store(v, o + pvoff);
        }
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, CP);
        if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, CP);
        if (DRAIN) __builtin_amdgcn_sched_group_barrier(0x002, 2, CP);
        if (DRAIN && i >= 4 && i < 20) __builtin_amdgcn_sched_group_barrier(0x040, 1, CP);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(V2_THREADS) void match_pipe_sim_kernel(const unsigned char* __restrict__ apk,
                                                                     const unsigned char* __restrict__ bpk,
                                                                     int R, int M, int G,
                                                                     float* __restrict__ sim,
                                                                     float* __restrict__ pval,      // [panels, R]
                                                                     int32_t* __restrict__ pidx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // PANEL_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31;
    const int h = lane >> 5;
    const int g = blockIdx.x % G;
    const int panel = blockIdx.x / G;
    const int col0 = panel * PANEL_COLS;

#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int gi = i * V2_THREADS + tid;
        const int col = gi >> 5, ch = gi & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(bpk + (long)(col0 + col) * ROW_BYTES + ch * 16);
        *reinterpret_cast<u32x4*>(smem + lds_chunk_off(col, ch)) = v;
    }
    __syncthreads();

    const int nrb = R / V2_ROWS;
    if (g >= nrb) return;
    u32x4 areg[16];
    {
        const unsigned char* arow = apk + (long)(g * V2_ROWS + wave * 32 + lr) * ROW_BYTES;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            areg[s] = *reinterpret_cast<const u32x4*>(arow + (2 * s + h) * 16);
            areg[8 + s] = *reinterpret_cast<const u32x4*>(arow + (16 + 2 * s + h) * 16);
        }
    }
    float best[16];
    int bidx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        best[i] = -INFINITY;
        bidx[i] = 0;
    }
    f32x16 a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 0.f;
    BFrag1 fr = read_bfrag1(smem, lr, 0, h);
    const long Ml = M;
    const unsigned pvoff = (unsigned)(4 * h) * (unsigned)M + (unsigned)(col0 + lr);   // per-lane part of the store offset

    pipe_step_sim<0, 0, false, false>(smem, lr, h, areg, fr, a, b, best, bidx, 0, sim, 0u, Ml, apk);

    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * V2_ROWS + wave * 32;                        // uniform
        float* pbase = sim + (long)row0 * Ml;
        const bool has_next = rb + G < nrb;
        const int nrow = (has_next ? (rb + G) * V2_ROWS : rb * V2_ROWS) + wave * 32 + lr;
        const unsigned char* arow_next = apk + (long)nrow * ROW_BYTES;
        const int gc = col0 + lr;
        pipe_step_sim<1, 0, false, true>(smem, lr, h, areg, fr, b, a, best, bidx, gc, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<2, 1, false, true>(smem, lr, h, areg, fr, a, b, best, bidx, gc + 32, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<3, 2, false, true>(smem, lr, h, areg, fr, b, a, best, bidx, gc + 64, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<4, 3, false, true>(smem, lr, h, areg, fr, a, b, best, bidx, gc + 96, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<5, 4, false, true>(smem, lr, h, areg, fr, b, a, best, bidx, gc + 128, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<6, 5, false, true>(smem, lr, h, areg, fr, a, b, best, bidx, gc + 160, pbase, pvoff, Ml, arow_next);
        pipe_step_sim<7, 6, true, true>(smem, lr, h, areg, fr, b, a, best, bidx, gc + 192, pbase, pvoff, Ml, arow_next);
        if (has_next) {
            pipe_step_sim<0, 7, false, true>(smem, lr, h, areg, fr, a, b, best, bidx, gc + 224, pbase, pvoff, Ml, arow_next);
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = b[reg];
                const bool t = v > best[reg];
                best[reg] = t ? v : best[reg];
                bidx[reg] = t ? gc + 224 : bidx[reg];
                float* o = pbase + (long)((reg & 3) + 8 * (reg >> 2)) * Ml + 224;
                __builtin_nontemporal_store(v, o + pvoff);
            }
        }
        finalize_rows(best, bidx, lane, h, row0, R, panel, pval, pidx);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            best[i] = -INFINITY;
            bidx[i] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void merge_splits_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                           int splits, int R, float* __restrict__ oval, int32_t* __restrict__ oidx)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float v = pval[r];
    int ix = pidx[r];
    for (int s = 1; s < splits; ++s) {               // ascending column ranges: strict '>' keeps the first maximum
        const float ov = pval[(long)s * R + r];
        if (ov > v) {
            v = ov;
            ix = pidx[(long)s * R + r];
        }
    }
    oval[r] = v;
    oidx[r] = ix;
}

int pick_splits(int R, int M, bool write_sim)
{
    const int row_tiles = gdm_cdiv(R, MT_ROWS);
    const int max_splits = gdm_cdiv(M, MT_COLS);
    int want;
    if (write_sim) want = gdm_cdiv(M, 512);                 // ~512 columns per workgroup: thousands of workgroups
    else want = gdm_cdiv(1024, row_tiles);                  // >= ~4 workgroups per CU when the batch is small
    if (want < 1) want = 1;
    if (want > max_splits) want = max_splits;
    if (want > 64) want = 64;
    return want;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// 1 = tile-streaming kernel, 2 = LDS-resident panel kernel, 3 (default) = 2 + in-wave software pipeline where the shape allows;
// GDM_MATCH_KERNEL overrides, for A/B runs.
int match_kernel_version()
{
    const char* e = getenv("GDM_MATCH_KERNEL");
    if (e && e[0] == '1') return 1;
    if (e && e[0] == '2') return 2;
    return 3;
}

} // namespace

extern "C" size_t gdm_match_rows_bytes(int rows)
{
    return rows < 1 ? 0 : align256((size_t)rows * ROW_BYTES);
}

extern "C" size_t gdm_match_partial_bytes(int B, int N)
{
    if (B < 1 || N < 1) return 0;
    return 2 * align256(64 * (size_t)B * N * sizeof(float));
}

extern "C" size_t gdm_match_workspace_bytes(int B, int N, int M)
{
    if (B < 1 || N < 1 || M < 1) return 0;
    return gdm_match_rows_bytes(B * N) + gdm_match_rows_bytes(M) + gdm_match_partial_bytes(B, N) + 256;
}

extern "C" int gdm_match_pack_hip(const float* x, int R, int Dd, int n, int precision, void* rows, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    GDM_CHECK_ARG(x && rows, "gdm_match_pack_hip: NULL pointer");
    GDM_CHECK_ARG(Dd == D, "gdm_match_pack_hip: D=%d, only D=128 is built", Dd);
    GDM_CHECK_ARG(R >= 1 && R <= 65535 && n >= 1, "gdm_match_pack_hip: bad shape R=%d n=%d", R, n);
    GDM_CHECK_ARG(precision == GDM_MATCH_BF16X3 || precision == GDM_MATCH_F32, "gdm_match_pack_hip: precision=%d", precision);
    GDM_CHECK_ARG(((uintptr_t)rows & 15) == 0, "gdm_match_pack_hip: rows must be 16-byte aligned");
    dim3 grid(gdm_cdiv(n, 64), R);
    if (precision == GDM_MATCH_BF16X3)
        hipLaunchKernelGGL(pack_rows_kernel<GDM_MATCH_BF16X3>, grid, dim3(256), 0, stream, x, n, (unsigned char*)rows);
    else
        hipLaunchKernelGGL(pack_rows_kernel<GDM_MATCH_F32>, grid, dim3(256), 0, stream, x, n, (unsigned char*)rows);
    return gdm_launch_status("pack_rows_kernel");
}

extern "C" int gdm_match_pack2_hip(const float* x1, int R1, int n1, void* rows1, const float* x2, int R2, int n2, void* rows2,
                                   int Dd, int precision, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    GDM_CHECK_ARG(x1 && rows1 && x2 && rows2, "gdm_match_pack2_hip: NULL pointer");
    GDM_CHECK_ARG(Dd == D, "gdm_match_pack2_hip: D=%d, only D=128 is built", Dd);
    GDM_CHECK_ARG(R1 >= 1 && R2 >= 1 && R1 + R2 <= 65535 && n1 >= 1 && n2 >= 1, "gdm_match_pack2_hip: bad shape R=%d+%d n=%d,%d", R1, R2, n1, n2);
    GDM_CHECK_ARG(precision == GDM_MATCH_BF16X3 || precision == GDM_MATCH_F32, "gdm_match_pack2_hip: precision=%d", precision);
    GDM_CHECK_ARG((((uintptr_t)rows1 | (uintptr_t)rows2) & 15) == 0, "gdm_match_pack2_hip: rows must be 16-byte aligned");
    dim3 grid(gdm_cdiv(n1 > n2 ? n1 : n2, 64), R1 + R2);
    if (precision == GDM_MATCH_BF16X3)
        hipLaunchKernelGGL(pack_rows2_kernel<GDM_MATCH_BF16X3>, grid, dim3(256), 0, stream, x1, R1, n1, (unsigned char*)rows1, x2, n2, (unsigned char*)rows2);
    else
        hipLaunchKernelGGL(pack_rows2_kernel<GDM_MATCH_F32>, grid, dim3(256), 0, stream, x1, R1, n1, (unsigned char*)rows1, x2, n2, (unsigned char*)rows2);
    return gdm_launch_status("pack_rows2_kernel");
}

extern "C" int gdm_match_packed_hip(const void* scene_rows, const void* model_rows, int R, int M, int precision,
                                    int32_t* best_idx, float* best_sim, float* sim,
                                    void* partial, size_t partial_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    GDM_CHECK_ARG(scene_rows && model_rows && best_idx && best_sim && partial, "gdm_match_packed_hip: NULL pointer");
    GDM_CHECK_ARG(R >= 1 && M >= 1, "gdm_match_packed_hip: bad shape R=%d M=%d", R, M);
    GDM_CHECK_ARG(precision == GDM_MATCH_BF16X3 || precision == GDM_MATCH_F32, "gdm_match_packed_hip: precision=%d", precision);
    const size_t half = align256(64 * (size_t)R * sizeof(float));
    if (partial_bytes < 2 * half) {
        gdm_set_error("gdm_match_packed_hip: partial workspace %zu < %zu bytes", partial_bytes, 2 * half);
        return GDM_ENOMEM;
    }
    const unsigned char* apk = (const unsigned char*)scene_rows;
    const unsigned char* bpk = (const unsigned char*)model_rows;
    float* pval = (float*)partial;
    int32_t* pidx = (int32_t*)((unsigned char*)partial + half);

    const bool ws_sim = sim != nullptr;
    int rc;
    const int panels = gdm_cdiv(M, PANEL_COLS);
    int nsplit;
    const int mkv = match_kernel_version();
    if (mkv >= 2 && panels <= 64) {
        // ---- v2: LDS-resident model panel, one persistent-ish workgroup per (panel, row group) ----
        const int nrb = gdm_cdiv(R, V2_ROWS);
        const bool pipe = mkv == 3 && precision == GDM_MATCH_BF16X3 && R % V2_ROWS == 0 && M % PANEL_COLS == 0 &&
                          (long)R * M < (1L << 40) && (long)32 * M < (1L << 30);
        // one workgroup per CU (the panel takes the LDS): v2 queues two rounds; v3 runs one round of <= 256 workgroups,
        // which the store path prefers (tools/micro/store_pattern.hip: 5.9 vs 4.9 TB/s)
        int G = (pipe ? 256 : 512) / panels;
        if (G < 1) G = 1;
        if (G > nrb) G = nrb;
        if (G >= 8) G &= ~7;                                    // keep same-rows workgroups on one bid%8 class
        nsplit = panels;
        dim3 grid(panels * G);
        float* ov = nsplit == 1 ? best_sim : pval;
        int32_t* oi = nsplit == 1 ? best_idx : pidx;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)match_panel_kernel<GDM_MATCH_BF16X3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            (void)hipFuncSetAttribute((const void*)match_panel_kernel<GDM_MATCH_BF16X3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            (void)hipFuncSetAttribute((const void*)match_panel_kernel<GDM_MATCH_F32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            (void)hipFuncSetAttribute((const void*)match_panel_kernel<GDM_MATCH_F32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            (void)hipFuncSetAttribute((const void*)match_pipe_sim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            (void)hipFuncSetAttribute((const void*)match_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
            attr_set = true;
        }
#define LAUNCH2(P, W) hipLaunchKernelGGL((match_panel_kernel<P, W>), grid, dim3(V2_THREADS), PANEL_BYTES, stream, apk, bpk, R, M, G, sim, ov, oi)
        if (pipe) {
            if (ws_sim) hipLaunchKernelGGL(match_pipe_sim_kernel, grid, dim3(V2_THREADS), PANEL_BYTES, stream, apk, bpk, R, M, G, sim, ov, oi);
            else hipLaunchKernelGGL(match_pipe_kernel, grid, dim3(V2_THREADS), PANEL_BYTES, stream, apk, bpk, R, M, G, sim, ov, oi);
        } else if (precision == GDM_MATCH_BF16X3) {
            if (ws_sim) LAUNCH2(GDM_MATCH_BF16X3, true); else LAUNCH2(GDM_MATCH_BF16X3, false);
        } else {
            if (ws_sim) LAUNCH2(GDM_MATCH_F32, true); else LAUNCH2(GDM_MATCH_F32, false);
        }
#undef LAUNCH2
        rc = gdm_launch_status("match_panel_kernel");
        if (rc) return rc;
    } else {
        const int splits = pick_splits(R, M, ws_sim);
        int cps = gdm_cdiv(gdm_cdiv(M, splits), MT_COLS) * MT_COLS;
        nsplit = gdm_cdiv(M, cps);
        dim3 grid(gdm_cdiv(R, MT_ROWS), nsplit);
        float* ov = nsplit == 1 ? best_sim : pval;
        int32_t* oi = nsplit == 1 ? best_idx : pidx;
        const size_t lds = 2 * TILE_BYTES;
#define LAUNCH(P, W) hipLaunchKernelGGL((match_kernel<P, W>), grid, dim3(256), lds, stream, apk, bpk, R, M, cps, sim, ov, oi)
        if (precision == GDM_MATCH_BF16X3) {
            if (ws_sim) LAUNCH(GDM_MATCH_BF16X3, true); else LAUNCH(GDM_MATCH_BF16X3, false);
        } else {
            if (ws_sim) LAUNCH(GDM_MATCH_F32, true); else LAUNCH(GDM_MATCH_F32, false);
        }
#undef LAUNCH
        rc = gdm_launch_status("match_kernel");
        if (rc) return rc;
    }
    if (nsplit > 1) {
        hipLaunchKernelGGL(merge_splits_kernel, dim3(gdm_cdiv(R, 256)), dim3(256), 0, stream, pval, pidx, nsplit, R, best_sim, best_idx);
        rc = gdm_launch_status("merge_splits_kernel");
    }
    return rc;
}

extern "C" int gdm_match_hip(const float* scene, const float* model, int B, int Dd, int N, int M, int precision,
                             int32_t* best_idx, float* best_sim, float* sim,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    GDM_CHECK_ARG(scene && model && best_idx && best_sim && workspace, "gdm_match_hip: NULL pointer");
    GDM_CHECK_ARG(Dd == D, "gdm_match_hip: D=%d, only D=128 is built", Dd);
    GDM_CHECK_ARG(B >= 1 && N >= 1 && M >= 1, "gdm_match_hip: bad shape B=%d N=%d M=%d", B, N, M);
    GDM_CHECK_ARG((long)B * N < (1L << 22), "gdm_match_hip: B*N too large");
    GDM_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "gdm_match_hip: workspace must be 16-byte aligned");
    if (workspace_bytes < gdm_match_workspace_bytes(B, N, M)) {
        gdm_set_error("gdm_match_hip: workspace %zu < %zu bytes", workspace_bytes, gdm_match_workspace_bytes(B, N, M));
        return GDM_ENOMEM;
    }
    unsigned char* ws = (unsigned char*)workspace;
    unsigned char* apk = ws;
    unsigned char* bpk = apk + gdm_match_rows_bytes(B * N);
    unsigned char* part = bpk + gdm_match_rows_bytes(M);
    int rc = gdm_match_pack2_hip(scene, B, N, apk, model, 1, M, bpk, Dd, precision, stream);
    if (rc) return rc;
    return gdm_match_packed_hip(apk, bpk, B * N, M, precision, best_idx, best_sim, sim, part, gdm_match_partial_bytes(B, N), stream);
}
