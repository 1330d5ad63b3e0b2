// Exact K-nearest-neighbour search on gfx950 (MI355X): brute force, fp32, canonical order.
//
// Replaces (reference, /root/reference):
//   models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135  cpp_knn_batch_omp
//   nanoflann.hpp:323-348 (distance arithmetic), :115-139 (result set)
//   22 calls per crop from datasets/lm/linemod_pbr.py:534-569
//
// Design (VALU-bound compare-and-select work, no MFMA):
//   * one launch serves a whole TABLE of independent searches (all pyramid calls x all crops);
//     the table travels as a by-value kernel argument, so a launch needs no device allocation
//     and can be captured in a hipGraph.
//   * K == 1 (knn_kernel<1>): a query is owned by T = 2^t lanes of one wave (T from the support size):
//     lane t scans support points t, t+T, ... from an LDS tile (float4 per point, one ds_read_b128 per
//     pair), keeps its best in registers, and the T partial results are merged with wave shuffles.
//   * K > 1 (knn_wave_kernel): ONE QUERY PER WAVE, 64 pairs per step, and NO sorted list in the scan loop.
//     A candidate is admitted by one compare against the query's current K-th (d2, index) bound and
//     appended (ballot + mbcnt) to a 64-slot buffer in LDS; only when the buffer is full the wave sorts it
//     across its lanes (bitonic, 21 compare-exchange stages), keeps the K smallest and tightens the bound.
//     Support points are visited in a hashed order (tile t = points t, t+ntiles, ...; slots inside a tile
//     permuted by an odd multiplier), so raster-ordered pixel grids behave like a random stream: a query
//     appends ~64 + 48 log4(S/64) candidates and sorts ~2 + log4(S/64) times, instead of paying a sorted
//     insertion (~15-25 dependent wave instructions) for each of them.  Round 1's lane-distributed list
//     spent 75 of its 93 wave instructions per 64 pairs on insertions; here the loop body is the distance
//     (8 VALU) + one compare + one branch.
//   * K > 1 against an ORGANISED support (a pixel grid of a depth map, gdm_knn_job.grid_w > 0; knn_grid_kernel): the points of one
//     pixel column share x/z, those of one row y/z (pinhole projection), so the distance from a query to ANY point of a column is
//     at least its distance to the plane through the origin that holds the column's rays: (x - a z)^2 / (1 + a^2), a = x'/z';
//     likewise for rows.  A pre-kernel measures, per crop, the range [lo, hi] of that ratio over the valid pixels of every
//     column and row (so nothing is ASSUMED about the data: an unstructured map just gets wide ranges and no pruning).  A query
//     then takes its first 64 candidates around the column / row whose range holds its own ratio, and afterwards only scans
//     the window of columns and rows whose lower bound is below its K-th distance: ~100 candidates instead of 16384, results
//     identical to the exhaustive search (depth holes -- points at the origin -- are covered by a |q|^2 test that widens the
//     window to the whole grid).
//   * distances are ((dx*dx)+dy*dy)+dz*dz with every operation rounded to fp32 (no FMA
//     contraction: __fmul_rn/__fadd_rn and -ffp-contract=off), exactly the reference's
//     arithmetic, so indices are bit-exact on tie-free inputs; ties are ordered by ascending
//     support index (the reference orders them by KD-tree traversal, which is not reproducible).
#include "gdm_common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int KNN_BLOCK = 256;
constexpr int KNN_TILE = 1024;          // support points per LDS tile (16 KiB)
constexpr int IDX_EMPTY = 0x7fffffff;

struct KnnJobDev {
    const float* support;
    const float* query;
    int32_t* idx;
    float* d2;
    long long support_bstride;
    long long query_bstride;
    int S, Q, K, logT;
    int blocks_per_b;                   // blocks that cover the Q queries of one batch item
    int block_begin;                    // first blockIdx.x of this job
    const float4* packed;               // K > 1 only: the support set as hashed float4 tiles [B][ntiles][npad] (workspace), or null
    const float* ranges;                // organised supports: per crop [col_lo W][col_hi W][row_lo H][row_hi H][bad, hole, -, -] (workspace)
    int grid_w, grid_h;
};

struct KnnTable {
    KnnJobDev jobs[GDM_KNN_MAX_JOBS];
    int njobs;
    int B;
};

__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float px, float py, float pz)
{
    // nanoflann.hpp:343-346 for dim == 3: result = ((0 + d0*d0) + d1*d1) + d2*d2, diff = query - point.  Scalar, explicitly rounded
    // operations (a packed x / y form -- v_pk_add_f32 / v_pk_mul_f32, same bits -- was measured: no faster; and a packed-fp32 read of a
    // register straight behind the LDS wait that retires it is not safe on this part: profiles/r04_knn1_pair_root_cause.md)
    const float d0 = __fsub_rn(qx, px);
    const float d1 = __fsub_rn(qy, py);
    const float d2 = __fsub_rn(qz, pz);
    float r = __fmul_rn(d0, d0);
    r = __fadd_rn(r, __fmul_rn(d1, d1));
    r = __fadd_rn(r, __fmul_rn(d2, d2));
    return r;
}

template <int KMAX>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const KnnTable tab)
{
    __shared__ float4 tile[KNN_TILE];

    // ---- which job / batch item / query block (all wave-uniform) ----
    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.njobs && bid >= tab.jobs[j + 1].block_begin) ++j;
    const KnnJobDev& job = tab.jobs[j];
    const int local = bid - job.block_begin;
    const int b = local / job.blocks_per_b;
    const int qb = local - b * job.blocks_per_b;
    const int logT = job.logT;
    const int T = 1 << logT;
    const int S = job.S, Q = job.Q, K = job.K;

    const int tid = threadIdx.x;
    const int t = tid & (T - 1);
    const int q = qb * (KNN_BLOCK >> logT) + (tid >> logT);
    const bool valid = q < Q;
    const int qc = valid ? q : Q - 1;

    const float* sup = job.support + (long long)b * job.support_bstride;
    const float* qry = job.query + (long long)b * job.query_bstride + (long long)qc * 3;
    const float qx = qry[0], qy = qry[1], qz = qry[2];

    float dl[KMAX];
    int il[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        dl[i] = INFINITY;
        il[i] = IDX_EMPTY;
    }

    for (int tile0 = 0; tile0 < S; tile0 += KNN_TILE) {
        __syncthreads();
        const int npt = min(KNN_TILE, S - tile0);
        for (int p = tid; p < KNN_TILE; p += KNN_BLOCK) {
            float4 v;
            if (p < npt) {
                const float* s3 = sup + (long long)(tile0 + p) * 3;
                v = make_float4(s3[0], s3[1], s3[2], 0.f);
            } else {
                v = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
            }
            tile[p] = v;
        }
        __syncthreads();

        const int steps = (npt + T - 1) >> logT;          // uniform trip count
        for (int s = 0; s < steps; ++s) {
            const int p = (s << logT) + t;                // < KNN_TILE: padded slots hold +inf
            const float4 v = tile[p];
            const float d = dist2_ref(qx, qy, qz, v.x, v.y, v.z);
            if (d < dl[KMAX - 1]) {                       // strict: an equal distance has a larger index
                // Branch-free sorted insertion, every slot independent of the others (no swap chain):
                //   new[i] = old[i-1] > d ? old[i-1] : (old[i] > d ? d : old[i])
                // '>' is strict, so the newcomer lands AFTER stored entries of equal distance (they have
                // lower indices).  Walking i downwards reads old[i-1] before it is overwritten.
                const int pi = tile0 + p;
                bool gt_hi = true;                        // old[KMAX-1] > d holds (admission test)
#pragma unroll
                for (int i = KMAX - 1; i > 0; --i) {
                    const bool gt_lo = dl[i - 1] > d;
                    const float dn = gt_lo ? dl[i - 1] : (gt_hi ? d : dl[i]);
                    const int in = gt_lo ? il[i - 1] : (gt_hi ? pi : il[i]);
                    dl[i] = dn;
                    il[i] = in;
                    gt_hi = gt_lo;
                }
                dl[0] = gt_hi ? d : dl[0];
                il[0] = gt_hi ? pi : il[0];
            }
        }
    }

    int32_t* out_i = job.idx + ((long long)b * Q + qc) * K;
    float* out_d = job.d2 ? job.d2 + ((long long)b * Q + qc) * K : nullptr;

    if (T == 1) {
        if (valid) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    out_i[k] = il[k] == IDX_EMPTY ? 0 : il[k];
                    if (out_d) out_d[k] = isinf(dl[k]) ? 3.402823466e+38f : dl[k];
                }
            }
        }
        return;
    }

    // ---- merge the T sorted lists of this query: K rounds of lexicographic (d, idx) arg-min ----
    for (int k = 0; k < K; ++k) {
        float bd = dl[0];
        int bi = il[0];
        for (int m = 1; m < T; m <<= 1) {
            const float od = __shfl_xor(bd, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (od < bd || (od == bd && oi < bi)) {
                bd = od;
                bi = oi;
            }
        }
        if (dl[0] == bd && il[0] == bi) {                 // the owner pops its head
#pragma unroll
            for (int i = 0; i < KMAX - 1; ++i) {
                dl[i] = dl[i + 1];
                il[i] = il[i + 1];
            }
            dl[KMAX - 1] = INFINITY;
            il[KMAX - 1] = IDX_EMPTY;
        }
        if (valid && t == (k & (T - 1))) {
            out_i[k] = bi == IDX_EMPTY ? 0 : bi;
            if (out_d) out_d[k] = isinf(bd) ? 3.402823466e+38f : bd;
        }
    }
}

// ---- K > 1: one query per wave, buffered candidates, lazy selection ---------------------------------------
constexpr int KW_WAVES = 8;             // queries per workgroup (one per wave)
constexpr int KW_BLOCK = KW_WAVES * 64;
constexpr int KW_TILE = 1024;           // support points per LDS tile (16 KiB)
constexpr int KW_HASH = 397;            // odd: p -> (p * KW_HASH) mod 2^k is a bijection of the tile's slots

__device__ __forceinline__ bool lex_less(float d, int i, float td, int ti) { return d < td || (d == td && i < ti); }

typedef unsigned long long u64;

// (d2, index) as ONE unsigned 64-bit key: d2 >= 0 (or +inf), so its float bits order like unsigned integers, index in the low
// word breaks ties: lexicographic (d2, index) order == integer order, one v_cmp_lt_u64 per comparison.
__device__ __forceinline__ u64 make_key(float d, int i) { return ((u64)(unsigned)__float_as_int(d) << 32) | (unsigned)i; }
constexpr u64 KEY_EMPTY = ((u64)0x7f800000u << 32) | (unsigned)IDX_EMPTY;

// lane i <- lane i ^ J of a 32-bit value.  J < 16 stays inside a row of 16 lanes: DPP moves (VALU, no LDS round trip);
// J = 16, 32 go through ds_bpermute.
template <int J>
__device__ __forceinline__ int xor_lane(int x)
{
    if constexpr (J == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, false);                // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, false);           // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_mov_dpp(x, 0x104, 0xF, 0x5, false);                          // banks 0,2 <- lane i+4 (row_shl:4)
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                           // banks 1,3 <- lane i-4 (row_shr:4)
    } else if constexpr (J == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, false);        // row_ror:8
    else return __shfl_xor(x, J, 64);
}

template <int K2, int J>
__device__ __forceinline__ void sort_stage(u64& key, int lane)
{
    const unsigned lo = (unsigned)xor_lane<J>((int)(unsigned)key);
    const unsigned hi = (unsigned)xor_lane<J>((int)(unsigned)(key >> 32));
    const u64 o = ((u64)hi << 32) | lo;
    const bool keep_min = ((lane & K2) == 0) == ((lane & J) == 0);
    key = ((o < key) == keep_min) ? o : key;
}

template <int K2, int J>
__device__ __forceinline__ void sort_merge(u64& key, int lane)
{
    sort_stage<K2, J>(key, lane);
    if constexpr (J > 1) sort_merge<K2, J / 2>(key, lane);
}

// Ascending bitonic sort of one key per lane over the 64 lanes of the wave: 21 branch-free compare-exchange stages
// (2 cross-lane moves, 1 compare, 1 scalar mask op, 2 selects each; 18 of them on DPP).  Keys are distinct except KEY_EMPTY padding.
__device__ __forceinline__ void wave_sort64(u64& key, int lane)
{
    sort_merge<2, 1>(key, lane);
    sort_merge<4, 2>(key, lane);
    sort_merge<8, 4>(key, lane);
    sort_merge<16, 8>(key, lane);
    sort_merge<32, 16>(key, lane);
    sort_merge<64, 32>(key, lane);
}

struct PackEntry {
    const float* support;
    float4* packed;
    long long support_bstride;
    int S, ntiles, npad;
    int block_begin;                    // first blockIdx.x of this entry; an entry has B * ntiles * npad / 256 blocks
};

struct PackTable {
    PackEntry e[GDM_KNN_MAX_JOBS];
    int n;
    int B;
};

__host__ __device__ inline void tile_geometry(int S, int& ntiles, int& npad)
{
    ntiles = (S + KW_TILE - 1) / KW_TILE;
    const int npt_max = (S + ntiles - 1) / ntiles;       // points in tile 0 (<= KW_TILE)
    npad = 64;
    while (npad < npt_max) npad <<= 1;                   // slots per tile: power of two in [64, KW_TILE]
}

// slot p of tile t holds support point ((p * KW_HASH) mod npad) * ntiles + t, or padding
__device__ __forceinline__ float4 hashed_point(const float* sup, int S, int ntiles, int npad, int t, int p)
{
    const int pp = (p * KW_HASH) & (npad - 1);
    const int npt = (S - t + ntiles - 1) / ntiles;       // valid points of this tile
    if (pp < npt) {
        const int gi = pp * ntiles + t;
        const float* s3 = sup + (long long)gi * 3;
        return make_float4(s3[0], s3[1], s3[2], __int_as_float(gi));
    }
    return make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(IDX_EMPTY));
}

// Every support set of a launch, once: [B][S][3] -> hashed float4 tiles [B][ntiles][npad] in the workspace, so that the search
// blocks stream whole tiles with 16-byte coalesced loads.  (Filling tiles straight from the [S][3] array costs one cache line
// per point and tile: 1024 blocks x 16 tiles x 196 KB = 3.2 GB of L2 traffic for the 16384-pixel supports of one batch.)
__global__ __launch_bounds__(256) void knn_pack_kernel(const PackTable tab)
{
    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.n && bid >= tab.e[j + 1].block_begin) ++j;
    const PackEntry& e = tab.e[j];
    const int per_b = e.ntiles * e.npad;
    const long long i = (long long)(bid - e.block_begin) * 256 + threadIdx.x;
    if (i >= (long long)tab.B * per_b) return;
    const int b = (int)(i / per_b);
    const int r = (int)(i - (long long)b * per_b);
    const int t = r / e.npad, p = r - t * e.npad;
    e.packed[i] = hashed_point(e.support + (long long)b * e.support_bstride, e.S, e.ntiles, e.npad, t, p);
}

__global__ __launch_bounds__(KW_BLOCK) void knn_wave_kernel(const KnnTable tab)
{
    __shared__ float4 tile[KW_TILE];
    __shared__ u64 cand[KW_WAVES][64];

    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.njobs && bid >= tab.jobs[j + 1].block_begin) ++j;
    const KnnJobDev& job = tab.jobs[j];
    const int local = bid - job.block_begin;
    const int b = local / job.blocks_per_b;
    const int qb = local - b * job.blocks_per_b;
    const int S = job.S, Q = job.Q, K = job.K;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = qb * KW_WAVES + wave;
    const bool valid = q < Q;                            // wave-uniform
    const int qc = valid ? q : Q - 1;
    const float* sup = job.support + (long long)b * job.support_bstride;
    const float* qry = job.query + (long long)b * job.query_bstride + (long long)qc * 3;
    const float qx = qry[0], qy = qry[1], qz = qry[2];
    u64* buf = cand[wave];

    float td = INFINITY;                                 // current K-th (d2, index): admission bound
    int ti = IDX_EMPTY;
    int cnt = 0;                                         // buffered candidates (wave-uniform)

    // keep the K smallest of the buffer (sorted, in slots 0..K-1) and tighten the bound
    auto compact = [&]() -> u64 {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        u64 key = lane < cnt ? buf[lane] : KEY_EMPTY;
        wave_sort64(key, lane);
        if (lane < K) buf[lane] = key;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt = min(cnt, K);
        td = __int_as_float(__shfl((int)(unsigned)(key >> 32), K - 1, 64));   // (inf, IDX_EMPTY) while fewer than K are known
        ti = __shfl((int)(unsigned)key, K - 1, 64);
        return key;
    };

    int ntiles, npad;
    tile_geometry(S, ntiles, npad);
    const int steps = npad >> 6;
    const float4* packed = job.packed ? job.packed + (long long)b * ntiles * npad : nullptr;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (packed) {
            for (int p = tid; p < npad; p += KW_BLOCK) tile[p] = packed[(long long)t * npad + p];
        } else {
            for (int p = tid; p < npad; p += KW_BLOCK) tile[p] = hashed_point(sup, S, ntiles, npad, t, p);
        }
        __syncthreads();
        if (!valid) continue;
        // admission of one step's 64 candidates (d2 in d, support index in di)
        auto admit = [&](float d, int di) {
            if (__ballot(d <= td) == 0ull) return;       // the common case: nobody is inside the bound
            bool pass = lex_less(d, di, td, ti);
            unsigned long long bal = __ballot(pass);
            while (bal) {                                // wave-uniform
                const int n = __builtin_popcountll(bal);
                if (cnt + n > 64 && cnt > K) {           // no room: select, tighten, re-test
                    compact();
                    pass = pass && lex_less(d, di, td, ti);
                    bal = __ballot(pass);
                    continue;
                }
                const int room = 64 - cnt;               // >= 32 after a compaction (K <= 32)
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                const bool now = pass && pos < room;
                if (now) buf[cnt + pos] = make_key(d, di);
                cnt += min(n, room);
                pass = pass && !now;
                bal = __ballot(pass);
            }
        };
        const float4* tl = tile + lane;
        int s = 0;
        for (; s + 4 <= steps; s += 4) {                 // four steps' reads and distances in flight before the first test
            float4 v4[4];
            float d4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = tl[(s + u) << 6];
#pragma unroll
            for (int u = 0; u < 4; ++u) d4[u] = dist2_ref(qx, qy, qz, v4[u].x, v4[u].y, v4[u].z);
#pragma unroll
            for (int u = 0; u < 4; ++u) admit(d4[u], __float_as_int(v4[u].w));
        }
        for (; s < steps; ++s) {                         // steps = 1 or 2 (supports of <= 128 points per tile)
            const float4 v = tl[s << 6];
            admit(dist2_ref(qx, qy, qz, v.x, v.y, v.z), __float_as_int(v.w));
        }
    }
    if (!valid) return;
    const u64 key = compact();
    if (lane < K) {
        const long long o = ((long long)b * Q + q) * K + lane;
        const int si = (int)(unsigned)key;
        const float sd = __int_as_float((int)(unsigned)(key >> 32));
        job.idx[o] = si == IDX_EMPTY ? 0 : si;
        if (job.d2) job.d2[o] = isinf(sd) ? 3.402823466e+38f : sd;
    }
}

// ---- K > 1, organised support ---------------------------------------------------------------------------------------------
constexpr int KG_MAXDIM = 256;          // grid width / height handled (4 columns or rows per lane)

struct RangeEntry {
    const float* support;
    float* ranges;
    long long support_bstride;
    int W, H;
    int nblocks;                        // B * ceil(H / 16)
};

struct RangeTable {
    RangeEntry e[GDM_KNN_MAX_JOBS];
    int n;
    int B;
};

constexpr int KG_ROWS = 16;
__host__ __device__ inline int range_chunks(int H) { return (H + KG_ROWS - 1) / KG_ROWS; }
// [key(-lo) | key(hi)] per column, the same per row, then two flag words PER BLOCK of the table's kernel (a block = 16 rows)
__host__ __device__ inline int range_floats(int W, int H) { return 2 * W + 2 * H + 2 * range_chunks(H); }

// Ratio ranges of every pixel column (x/z) and row (y/z) over the valid pixels (z > 0) of a crop.  The table is stored as ordered
// unsigned keys (0 = nothing recorded): [key(-lo) | key(hi)] per column, then per row, then two flag words per block: bad = a point
// with z <= 0 that is not the origin, or a non-finite coordinate (no pruning for this crop); hole = a point at the origin.  One
// block per (entry, crop, 16 rows): it reduces ITS 16 rows (and their flags), and -- over all H rows -- ITS slice of the columns
// (W / blocks-per-crop of them), so every word of the table is written exactly once by plain stores: no zero fill in front of
// the kernel, no atomics (a max is exact in any order: the same table the atomicMax form made).
__device__ __forceinline__ unsigned ord_key(float v)
{
    const unsigned u = (unsigned)__float_as_int(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);         // monotone float -> unsigned; 0 is below every real key
}
__device__ __forceinline__ float ord_val(unsigned k)           // inverse; key 0 (nothing recorded) -> -inf
{
    if (k == 0u) return -INFINITY;
    return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k));
}

__global__ __launch_bounds__(256) void knn_grid_ranges_kernel(const RangeTable tab)
{
    __shared__ float snlo[16][17], shi[16][17];
    int blk = blockIdx.x, ei = 0;
    while (ei + 1 < tab.n && blk >= tab.e[ei].nblocks) { blk -= tab.e[ei].nblocks; ++ei; }
    const RangeEntry& e = tab.e[ei];
    const int W = e.W, H = e.H;
    const int chunks = (H + KG_ROWS - 1) / KG_ROWS;
    const int b = blk / chunks, v0 = (blk - b * chunks) * KG_ROWS;
    const float* sup = e.support + (long long)b * e.support_bstride;
    unsigned* out = reinterpret_cast<unsigned*>(e.ranges) + (long long)b * range_floats(W, H);
    const int tid = threadIdx.x;
    unsigned flags = 0;
    // rows v0 .. v0 + 15: 16 lanes per row
    {
        const int v = v0 + (tid >> 4), l = tid & 15;
        float nlo = -INFINITY, hi = -INFINITY;                  // max of -a and of a
        if (v < H)
            for (int u = l; u < W; u += 16) {
                const float* p = sup + ((long long)v * W + u) * 3;
                const float x = p[0], y = p[1], z = p[2];
                if (!(isfinite(x) && isfinite(y) && isfinite(z))) flags |= 1u;
                else if (z > 0.f) {
                    const float a = y / z;
                    nlo = fmaxf(nlo, -a);
                    hi = fmaxf(hi, a);
                } else if (x == 0.f && y == 0.f && z == 0.f) flags |= 2u;
                else flags |= 1u;
            }
        for (int m = 1; m < 16; m <<= 1) {
            nlo = fmaxf(nlo, __shfl_xor(nlo, m, 64));
            hi = fmaxf(hi, __shfl_xor(hi, m, 64));
        }
        if (v < H && l == 0) {
            out[2 * W + v] = isinf(nlo) ? 0u : ord_key(nlo);
            out[2 * W + H + v] = isinf(hi) ? 0u : ord_key(hi);
        }
    }
    // this block's flag words
    {
        const int bad = __syncthreads_or((int)(flags & 1u)), hole = __syncthreads_or((int)(flags & 2u));
        if (tid == 0) {
            const int c = v0 / KG_ROWS;
            out[2 * W + 2 * H + 2 * c] = bad ? 1u : 0u;
            out[2 * W + 2 * H + 2 * c + 1] = hole ? 1u : 0u;
        }
    }
    // this block's slice of the columns over ALL rows: thread = (row group rg of 16, column lane cl of 16)
    {
        const int c = v0 / KG_ROWS;
        const int cb = (int)((long long)c * W / chunks), ce = (int)((long long)(c + 1) * W / chunks);
        const int rg = tid >> 4, cl = tid & 15;
        for (int u0 = cb; u0 < ce; u0 += 16) {                          // block-uniform trip count
            const int u = u0 + cl;
            float nlo = -INFINITY, hi = -INFINITY;
            if (u < ce)
                for (int v = rg; v < H; v += 16) {
                    const float* p = sup + ((long long)v * W + u) * 3;
                    const float x = p[0], z = p[2];
                    if (z > 0.f && isfinite(x) && isfinite(z)) {
                        const float a = x / z;
                        nlo = fmaxf(nlo, -a);
                        hi = fmaxf(hi, a);
                    }
                }
            snlo[rg][cl] = nlo;
            shi[rg][cl] = hi;
            __syncthreads();
            if (tid < 16 && u0 + tid < ce) {
                float a = snlo[0][tid], b2 = shi[0][tid];
                for (int r = 1; r < 16; ++r) {
                    a = fmaxf(a, snlo[r][tid]);
                    b2 = fmaxf(b2, shi[r][tid]);
                }
                out[u0 + tid] = isinf(a) ? 0u : ord_key(a);
                out[W + u0 + tid] = isinf(b2) ? 0u : ord_key(b2);
            }
            __syncthreads();
        }
    }
}

// lower bound of the squared distance from (p, z) -- one coordinate and the depth of the query -- to any point whose ratio
// coordinate / depth lies in [lo, hi]: 0 inside the range, else the nearer of the two bounding planes.  f(a) = (p - a z)^2 / (1 + a^2)
// is unimodal away from a = p / z on either side up to the perpendicular direction and falls beyond it, so its minimum over an
// interval that does not hold p / z is at an end point.  An empty range (lo > hi: no valid pixel) gives +inf.
__device__ __forceinline__ float plane_bound(float p, float z, float lo, float hi)
{
    if (!(lo <= hi)) return INFINITY;
    const float r = p / z;
    if (r >= lo && r <= hi) return 0.f;
    const float dl = p - lo * z, dh = p - hi * z;
    return fminf(dl * dl / (1.f + lo * lo), dh * dh / (1.f + hi * hi));
}

__global__ __launch_bounds__(KW_BLOCK) void knn_grid_kernel(const KnnTable tab)
{
    __shared__ u64 cand[KW_WAVES][64];

    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.njobs && bid >= tab.jobs[j + 1].block_begin) ++j;
    const KnnJobDev& job = tab.jobs[j];
    const int local = bid - job.block_begin;
    const int b = local / job.blocks_per_b;
    const int qb = local - b * job.blocks_per_b;
    const int Q = job.Q, K = job.K, W = job.grid_w, H = job.grid_h;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = qb * KW_WAVES + wave;
    if (q >= Q) return;                                   // wave-uniform; no block-level barrier in this kernel
    const float* sup = job.support + (long long)b * job.support_bstride;
    const float* qry = job.query + (long long)b * job.query_bstride + (long long)q * 3;
    const float qx = qry[0], qy = qry[1], qz = qry[2];
    const unsigned* rk = reinterpret_cast<const unsigned*>(job.ranges) + (long long)b * range_floats(W, H);
    u64* buf = cand[wave];

    float td = INFINITY;
    int ti = IDX_EMPTY;
    int cnt = 0;
    auto compact = [&]() -> u64 {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        u64 key = lane < cnt ? buf[lane] : KEY_EMPTY;
        wave_sort64(key, lane);
        if (lane < K) buf[lane] = key;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt = min(cnt, K);
        td = __int_as_float(__shfl((int)(unsigned)(key >> 32), K - 1, 64));
        ti = __shfl((int)(unsigned)key, K - 1, 64);
        return key;
    };
    // admission of one step's candidates; `incl`: the bound itself passes (second phase, after the buffer was emptied)
    auto admit = [&](float d, int di, bool live, bool incl) {
        bool pass = live && (lex_less(d, di, td, ti) || (incl && d == td && di == ti));
        unsigned long long bal = __ballot(pass);
        while (bal) {
            const int n = __builtin_popcountll(bal);
            if (cnt + n > 64 && cnt > K) {
                compact();
                pass = pass && (lex_less(d, di, td, ti) || (incl && d == td && di == ti));
                bal = __ballot(pass);
                continue;
            }
            const int room = 64 - cnt;
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            const bool now = pass && pos < room;
            if (now) buf[cnt + pos] = make_key(d, di);
            cnt += min(n, room);
            pass = pass && !now;
            bal = __ballot(pass);
        }
    };
    auto eval = [&](int r, int c, bool live, bool incl) {  // grid cell (r, c) as a candidate
        const int gi = live ? r * W + c : 0;
        const float* p = sup + (long long)gi * 3;
        admit(dist2_ref(qx, qy, qz, p[0], p[1], p[2]), gi, live, incl);
    };

    unsigned any_bad = 0u, any_hole = 0u;                                      // the table's flag words, one pair per 16 rows
    for (int c = 0; c < range_chunks(H); ++c) {
        any_bad |= rk[2 * W + 2 * H + 2 * c];
        any_hole |= rk[2 * W + 2 * H + 2 * c + 1];
    }
    const bool structured = any_bad == 0u && qz > 0.f && isfinite(qx) && isfinite(qy) && isfinite(qz);
    const bool has_hole = any_hole != 0u;
    const float q2 = dist2_ref(qx, qy, qz, 0.f, 0.f, 0.f);
    // this lane's columns lane, lane + 64, ... and rows: plane bounds (0 where the query's own ratio lies inside the range)
    float lbc[KG_MAXDIM / 64], lbr[KG_MAXDIM / 64];
#pragma unroll
    for (int i = 0; i < KG_MAXDIM / 64; ++i) {
        const int u = lane + 64 * i;
        lbc[i] = (structured && u < W) ? plane_bound(qx, qz, -ord_val(rk[u]), ord_val(rk[W + u])) : (u < W ? 0.f : INFINITY);
        lbr[i] = (structured && u < H) ? plane_bound(qy, qz, -ord_val(rk[2 * W + u]), ord_val(rk[2 * W + H + u])) : (u < H ? 0.f : INFINITY);
    }
    // centre cell: the column / row with the smallest bound (lowest index among equals)
    auto arg_min = [&](const float (&lb)[KG_MAXDIM / 64]) {
        float bv = lb[0];
        int bi = lane;
#pragma unroll
        for (int i = 1; i < KG_MAXDIM / 64; ++i)
            if (lb[i] < bv) { bv = lb[i]; bi = lane + 64 * i; }
        for (int m = 1; m < 64; m <<= 1) {
            const float ov = __shfl_xor(bv, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        return bi;
    };
    int c0 = 0, c1 = W - 1, r0 = 0, r1 = H - 1;
    if (structured && W >= 8 && H >= 8) {
        // phase 1: the 8 x 8 cells around the centre -> a first K-th distance
        const int uc = min(max(arg_min(lbc) - 4, 0), W - 8), vc = min(max(arg_min(lbr) - 4, 0), H - 8);
        eval(vc + (lane >> 3), uc + (lane & 7), true, false);
        compact();
        // phase 2: only columns / rows whose bound (less a rounding allowance, scale-free) can still be below the K-th distance
        const float slack = 2e-6f * sqrtf(q2);
        int cl = W, ch = -1, rl = H, rh = -1;
#pragma unroll
        for (int i = 0; i < KG_MAXDIM / 64; ++i) {
            const int u = lane + 64 * i;
            const float sc = sqrtf(lbc[i]) - slack, sr = sqrtf(lbr[i]) - slack;
            const bool kc = u < W && (sc <= 0.f || sc * sc * 0.999f <= td);
            const bool kr = u < H && (sr <= 0.f || sr * sr * 0.999f <= td);
            if (kc) { cl = min(cl, u); ch = max(ch, u); }
            if (kr) { rl = min(rl, u); rh = max(rh, u); }
        }
        for (int m = 1; m < 64; m <<= 1) {
            cl = min(cl, __shfl_xor(cl, m, 64)); ch = max(ch, __shfl_xor(ch, m, 64));
            rl = min(rl, __shfl_xor(rl, m, 64)); rh = max(rh, __shfl_xor(rh, m, 64));
        }
        const bool holes_matter = has_hole && q2 * 0.999f <= td;   // points at the origin lie outside every plane bound
        if (!holes_matter && ch >= cl && rh >= rl) { c0 = cl; c1 = ch; r0 = rl; r1 = rh; }
        cnt = 0;                                          // the window holds the K best again: start over, bound inclusive
    }
    const bool incl = structured && W >= 8 && H >= 8;
    const int wc = c1 - c0 + 1, total = wc * (r1 - r0 + 1);
    const float inv_wc = 1.f / (float)wc;
    for (int e0 = 0; e0 < total; e0 += 64) {
        const int e = e0 + lane;
        int r = (int)((float)e * inv_wc);                 // e / wc, corrected for the float estimate
        r -= (r * wc > e);
        r += ((r + 1) * wc <= e);
        const int c = e - r * wc;
        eval(r0 + r, c0 + c, e < total, incl);
    }
    const u64 key = compact();
    if (lane < K) {
        const long long o = ((long long)b * Q + q) * K + lane;
        const int si = (int)(unsigned)key;
        const float sd = __int_as_float((int)(unsigned)(key >> 32));
        job.idx[o] = si == IDX_EMPTY ? 0 : si;
        if (job.d2) job.d2[o] = isinf(sd) ? 3.402823466e+38f : sd;
    }
}

// ---- K > 1, large unorganised support: uniform (x, y) cell lists ---------------------------------------------------------------
// The cloud's own K = 16 search (2048 x 2048 per crop, linemod_pbr.py:535) was the one search left that evaluated every pair: ~110
// admitted candidates and 3-4 sorts per query on top of 32 scan steps.  Here the support of a crop is binned ONCE into G x G cells (kc_grid)
// of its (x, y) bounding box (knn_cells_bin_kernel: counting sort in LDS, one workgroup per crop) and a query visits the cells ring
// by ring around its own -- 3 x 3 first, one wave step per cell row -- until the nearest edge of the visited window lies farther than
// the current K-th distance: dx (or dy) alone already exceeds it for every unvisited point.  Exact for any data (same (d2, index)
// keys, the window just grows to the whole grid in the worst case); the bound carries a 0.1 % allowance on d2 for the rounding of the
// cell assignment.
constexpr int KC_MIN_S = 1024;          // supports below this stay on knn_wave_kernel
constexpr int KC_GMAX = 32;             // cells per axis: 16 below 8192 points, 32 from there (kc_grid).  Measured on the pyramid's 2048-point clouds
                                        // (tools/knn_cells_stats.py): 16 -> 1.8 rings, 2.6 row batches, 2.7 sorts per query, 64 us per batch of
                                        // 16 crops; 32 -> 2.2 rings, 4.9 row batches, 2.5 sorts, 99 us: the row batches (L2 round trips), not the
                                        // distance evaluations, are what a query pays for
constexpr int KC_CELLS_MAX = KC_GMAX * KC_GMAX;
constexpr int KC_FOFF = (KC_CELLS_MAX + 1 + 3) & ~3;     // words per crop: cell_start[G * G + 1], then at KC_FOFF xmin, ymin, cw, ch, 1/cw, 1/ch, -, -
constexpr int KC_META = KC_FOFF + 8;
constexpr int KC_BIN_THREADS = 1024;
__host__ __device__ inline int kc_grid(int S) { return S >= 8192 ? 32 : 16; }

struct CellEntry {
    const float* support;
    float4* sorted;                     // [B][S]: (x, y, z, index) in cell order
    int* meta;                          // [B][KC_META]
    long long support_bstride;
    int S;
};

struct CellTable {
    CellEntry e[GDM_KNN_MAX_JOBS];
    int n;
    int B;
};

__device__ __forceinline__ int cell_coord(float v, float lo, float inv, int G)
{
    const float f = (v - lo) * inv;                      // NaN / -inf compare false -> cell 0; +inf -> the last cell
    return f >= 0.f ? (f < (float)G ? (int)f : G - 1) : 0;
}

__global__ __launch_bounds__(KC_BIN_THREADS) void knn_cells_bin_kernel(const CellTable tab)
{
    __shared__ float red[4][KC_BIN_THREADS / 64];
    __shared__ int cnt[KC_CELLS_MAX], cur[KC_CELLS_MAX];
    __shared__ float box[4];
    const int ei = blockIdx.x / tab.B, b = blockIdx.x - ei * tab.B;
    const CellEntry& e = tab.e[ei];
    const int S = e.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = kc_grid(S), cells = G * G;
    const float* sup = e.support + (long long)b * e.support_bstride;
    float4* sorted = e.sorted + (long long)b * S;
    int* meta = e.meta + (long long)b * KC_META;
    // 1. bounding box of the finite (x, y)
    float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int p = tid; p < S; p += KC_BIN_THREADS) {
        const float x = sup[(long long)p * 3], y = sup[(long long)p * 3 + 1];
        if (isfinite(x)) { xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); }
        if (isfinite(y)) { ylo = fminf(ylo, y); yhi = fmaxf(yhi, y); }
    }
    for (int m = 1; m < 64; m <<= 1) {
        xlo = fminf(xlo, __shfl_xor(xlo, m, 64)); xhi = fmaxf(xhi, __shfl_xor(xhi, m, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, m, 64)); yhi = fmaxf(yhi, __shfl_xor(yhi, m, 64));
    }
    if (lane == 0) { red[0][wave] = xlo; red[1][wave] = xhi; red[2][wave] = ylo; red[3][wave] = yhi; }
    if (tid < cells) cnt[tid] = 0;                     // cells <= KC_BIN_THREADS
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < KC_BIN_THREADS / 64; ++w) {
            xlo = fminf(xlo, red[0][w]); xhi = fmaxf(xhi, red[1][w]); ylo = fminf(ylo, red[2][w]); yhi = fmaxf(yhi, red[3][w]);
        }
        if (!(xhi >= xlo)) xlo = xhi = 0.f;              // no finite coordinate at all
        if (!(yhi >= ylo)) ylo = yhi = 0.f;
        const float cw = (xhi - xlo) / (float)G, ch = (yhi - ylo) / (float)G;
        box[0] = xlo; box[1] = ylo;
        box[2] = cw > 0.f ? 1.f / cw : 0.f;              // a degenerate axis puts everything into its cell 0
        box[3] = ch > 0.f ? 1.f / ch : 0.f;
        float* fm = reinterpret_cast<float*>(meta + KC_FOFF);
        fm[0] = xlo; fm[1] = ylo; fm[2] = cw; fm[3] = ch; fm[4] = box[2]; fm[5] = box[3]; fm[6] = 0.f; fm[7] = 0.f;
    }
    __syncthreads();
    const float bx0 = box[0], by0 = box[1], icw = box[2], ich = box[3];
    // 2. histogram, 3. exclusive scan, 4. scatter (order inside a cell is arbitrary: the search orders candidates by (d2, index) keys)
    for (int p = tid; p < S; p += KC_BIN_THREADS) {
        const int c = cell_coord(sup[(long long)p * 3 + 1], by0, ich, G) * G + cell_coord(sup[(long long)p * 3], bx0, icw, G);
        atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    if (tid < 64) {                                      // wave 0: cells / 64 consecutive cells per lane, shuffle scan of the lane sums
        const int per = cells / 64;
        int sum = 0;
        for (int i = 0; i < per; ++i) sum += cnt[tid * per + i];
        int inc = sum;
        for (int m = 1; m < 64; m <<= 1) {
            const int o = __shfl_up(inc, m, 64);
            if (lane >= m) inc += o;
        }
        int run = inc - sum;
        for (int i = 0; i < per; ++i) {
            const int c = tid * per + i, v = cnt[c];
            cur[c] = run;
            meta[c] = run;
            run += v;
        }
        if (tid == 63) meta[cells] = run;                // == S
    }
    __syncthreads();
    for (int p = tid; p < S; p += KC_BIN_THREADS) {
        const float x = sup[(long long)p * 3], y = sup[(long long)p * 3 + 1], z = sup[(long long)p * 3 + 2];
        const int c = cell_coord(y, by0, ich, G) * G + cell_coord(x, bx0, icw, G);
        const int pos = atomicAdd(&cur[c], 1);
        sorted[pos] = make_float4(x, y, z, __int_as_float(p));
    }
}

#ifdef GDM_KNN_STATS
__device__ unsigned long long gdm_knn_stats_dev[8];      // queries, rings, row batches, extra steps, compactions, admitted, -, -
#define KSTAT(i, v) do { if (lane == 0) atomicAdd(&gdm_knn_stats_dev[i], (unsigned long long)(v)); } while (0)
#else
#define KSTAT(i, v) do { } while (0)
#endif

__global__ __launch_bounds__(KW_BLOCK) void knn_cells_kernel(const KnnTable tab)
{
    __shared__ int cstart[KC_CELLS_MAX + 1];
    __shared__ u64 cand[KW_WAVES][64];

    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.njobs && bid >= tab.jobs[j + 1].block_begin) ++j;
    const KnnJobDev& job = tab.jobs[j];
    const int local = bid - job.block_begin;
    const int b = local / job.blocks_per_b;
    const int qb = local - b * job.blocks_per_b;
    const int S = job.S, Q = job.Q, K = job.K;
    const int G = kc_grid(S);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* meta = reinterpret_cast<const int*>(job.ranges) + (long long)b * KC_META;
    const float4* sorted = job.packed + (long long)b * S;
    for (int i = tid; i <= G * G; i += KW_BLOCK) cstart[i] = meta[i];
    __syncthreads();
    const int q = qb * KW_WAVES + wave;
    if (q >= Q) return;                                  // wave-uniform; no barrier below
    const float* fm = reinterpret_cast<const float*>(meta + KC_FOFF);
    const float bx0 = fm[0], by0 = fm[1], cw = fm[2], ch = fm[3], icw = fm[4], ich = fm[5];
    const float* qry = job.query + (long long)b * job.query_bstride + (long long)q * 3;
    const float qx = qry[0], qy = qry[1], qz = qry[2];
    u64* buf = cand[wave];

    float td = INFINITY;
    int ti = IDX_EMPTY;
    int cnt = 0;
    auto compact = [&]() -> u64 {
        KSTAT(4, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        u64 key = lane < cnt ? buf[lane] : KEY_EMPTY;
        wave_sort64(key, lane);
        if (lane < K) buf[lane] = key;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt = min(cnt, K);
        td = __int_as_float(__shfl((int)(unsigned)(key >> 32), K - 1, 64));
        ti = __shfl((int)(unsigned)key, K - 1, 64);
        return key;
    };
    auto admit = [&](float d, int di) {
        if (__ballot(d <= td) == 0ull) return;
        bool pass = lex_less(d, di, td, ti);
        unsigned long long bal = __ballot(pass);
        while (bal) {
            const int n = __builtin_popcountll(bal);
            if (cnt + n > 64 && cnt > K) {
                compact();
                pass = pass && lex_less(d, di, td, ti);
                bal = __ballot(pass);
                continue;
            }
            const int room = 64 - cnt;
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            const bool now = pass && pos < room;
            if (now) buf[cnt + pos] = make_key(d, di);
            cnt += min(n, room);
            pass = pass && !now;
            bal = __ballot(pass);
        }
    };
    // Rows of the window, FOUR at a time: a row's new cells are sorted[l0, l1) and sorted[r0, r1) (its two flanks, or its whole width);
    // the first 64 points of each of the four rows are loaded together (four independent L2 round trips in flight instead of one
    // after the other -- the search is latency-bound), then admitted row by row; a row with more than 64 new points continues alone.
    auto rows4 = [&](int y0, int ny1, int nx0, int nx1, int ox0, int ox1, int oy0, int oy1) {
        int l0[4], nl[4], r0[4], tot[4];
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + i;
            l0[i] = nl[i] = r0[i] = tot[i] = 0;
            if (y <= ny1) {
                const int row = y * G;
                if (y < oy0 || y > oy1) {                             // a new row: whole width
                    l0[i] = cstart[row + nx0];
                    nl[i] = cstart[row + nx1 + 1] - l0[i];
                    tot[i] = nl[i];
                } else {                                              // an old row: the two new flanks
                    l0[i] = cstart[row + nx0];
                    nl[i] = cstart[row + ox0] - l0[i];
                    r0[i] = cstart[row + ox1 + 1];
                    tot[i] = nl[i] + cstart[row + nx1 + 1] - r0[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = lane < tot[i] ? (lane < nl[i] ? l0[i] + lane : r0[i] + (lane - nl[i])) : 0;
            v[i] = sorted[pos];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool live = lane < tot[i];
            admit(live ? dist2_ref(qx, qy, qz, v[i].x, v[i].y, v[i].z) : INFINITY, live ? __float_as_int(v[i].w) : IDX_EMPTY);
            for (int e0 = 64; e0 < tot[i]; e0 += 64) {
                KSTAT(3, 1);
                const int e = e0 + lane;
                const bool lv = e < tot[i];
                const float4 w = sorted[lv ? (e < nl[i] ? l0[i] + e : r0[i] + (e - nl[i])) : 0];
                admit(lv ? dist2_ref(qx, qy, qz, w.x, w.y, w.z) : INFINITY, lv ? __float_as_int(w.w) : IDX_EMPTY);
            }
        }
    };

    const int qcx = cell_coord(qx, bx0, icw, G), qcy = cell_coord(qy, by0, ich, G);
    int ox0 = qcx, ox1 = qcx - 1, oy0 = qcy, oy1 = qcy - 1;          // visited window: empty
    int r = 1;
    u64 key = KEY_EMPTY;
    KSTAT(0, 1);
    while (true) {
        KSTAT(1, 1);
        const int nx0 = max(qcx - r, 0), nx1 = min(qcx + r, G - 1), ny0 = max(qcy - r, 0), ny1 = min(qcy + r, G - 1);
        for (int y = ny0; y <= ny1; y += 4) { KSTAT(2, 1); rows4(y, ny1, nx0, nx1, ox0, ox1, oy0, oy1); }
        key = compact();
        ox0 = nx0; ox1 = nx1; oy0 = ny0; oy1 = ny1;
        if (nx0 == 0 && nx1 == G - 1 && ny0 == 0 && ny1 == G - 1) break;                 // the whole grid
        // distance from the query to the nearest edge of the window that still has cells beyond it
        const float exl = nx0 == 0 ? INFINITY : qx - (bx0 + (float)nx0 * cw);
        const float exh = nx1 == G - 1 ? INFINITY : (bx0 + (float)(nx1 + 1) * cw) - qx;
        const float eyl = ny0 == 0 ? INFINITY : qy - (by0 + (float)ny0 * ch);
        const float eyh = ny1 == G - 1 ? INFINITY : (by0 + (float)(ny1 + 1) * ch) - qy;
        const float m = fminf(fminf(exl, exh), fminf(eyl, eyh));
        if (m > 0.f && m * m * 0.999f > td) break;                                       // (td = inf while fewer than K are known)
        // next ring: far enough for the current K-th distance where it is known, else twice as far
        const float cmin = fminf(cw > 0.f ? cw : INFINITY, ch > 0.f ? ch : INFINITY);
        int want = r * 2;
        if (td < INFINITY && cmin < INFINITY) want = (int)fminf((float)G, ceilf(sqrtf(td) / cmin)) + 1;
        r = min(G, max(r + 1, want));
    }
    if (lane < K) {
        const long long o = ((long long)b * Q + q) * K + lane;
        const int si = (int)(unsigned)key;
        const float sd = __int_as_float((int)(unsigned)(key >> 32));
        job.idx[o] = si == IDX_EMPTY ? 0 : si;
        if (job.d2) job.d2[o] = isinf(sd) ? 3.402823466e+38f : sd;
    }
}

int pick_logT(int S)
{
    // every lane scans >= 128 support points where possible; T in {1,2,...,64}
    int logT = 0;
    while (logT < 6 && (S >> (logT + 1)) >= 128) ++logT;
    return logT;
}

void fill_job(KnnJobDev& d, const gdm_knn_job& j)
{
    d.support = j.support;
    d.query = j.query;
    d.idx = j.idx;
    d.d2 = j.d2;
    d.support_bstride = j.support_bstride;
    d.query_bstride = j.query_bstride;
    d.S = j.S;
    d.Q = j.Q;
    d.K = j.K;
    d.packed = nullptr;
    d.ranges = nullptr;
    d.grid_w = d.grid_h = 0;
}

// K == 1 jobs: per-lane best + shuffle merge
int launch_k1(const gdm_knn_job* jobs, int njobs, int B, hipStream_t stream)
{
    KnnTable tab;
    tab.njobs = 0;
    tab.B = B;
    int nblocks = 0;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].K != 1) continue;
        KnnJobDev& d = tab.jobs[tab.njobs++];
        fill_job(d, jobs[i]);
        d.logT = pick_logT(d.S);
        d.blocks_per_b = gdm_cdiv(d.Q, KNN_BLOCK >> d.logT);
        d.block_begin = nblocks;
        nblocks += d.blocks_per_b * B;
    }
    if (tab.njobs == 0) return 0;
    hipLaunchKernelGGL(knn_kernel<1>, dim3(nblocks), dim3(KNN_BLOCK), 0, stream, tab);
    return gdm_launch_status("knn_kernel<1>");
}

size_t packed_bytes(int S, int B)
{
    int ntiles, npad;
    tile_geometry(S, ntiles, npad);
    return (size_t)B * ntiles * npad * sizeof(float4);
}

bool is_grid_job(const gdm_knn_job& j)
{
    const int W = j.grid_w;
    return j.K > 1 && W > 0 && j.S % W == 0 && W <= KG_MAXDIM && j.S / W <= KG_MAXDIM && W >= 8 && j.S / W >= 8;
}

size_t ranges_bytes(const gdm_knn_job& j, int B) { return (size_t)B * range_floats(j.grid_w, j.S / j.grid_w) * sizeof(float); }

// a large unorganised support: cell lists (knn_cells_bin_kernel + knn_cells_kernel)
bool is_cell_job(const gdm_knn_job& j) { return j.K > 1 && !is_grid_job(j) && j.S >= KC_MIN_S; }
size_t cells_bytes(int S, int B) { return (size_t)B * S * sizeof(float4) + (((size_t)B * KC_META * sizeof(int) + 15) & ~(size_t)15); }

bool same_support(const gdm_knn_job& a, const gdm_knn_job& b)
{
    return a.support == b.support && a.S == b.S && a.support_bstride == b.support_bstride;
}

// K in [2, 32] jobs: one query per wave.  Large jobs first, so that the tail of the launch is made of short blocks.
// With a workspace every distinct unorganised support set is re-laid out once (knn_pack_kernel) and shared by the jobs that search
// it, and the searches against organised supports (grid_w > 0) run as window searches (knn_grid_ranges_kernel + knn_grid_kernel).
int launch_wave(const gdm_knn_job* jobs, int njobs, int B, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    KnnTable tab, gtab, ctab;                            // wave / organised / cell lists
    tab.njobs = gtab.njobs = ctab.njobs = 0;
    tab.B = gtab.B = ctab.B = B;
    CellTable ct;
    ct.n = 0;
    ct.B = B;
    const gdm_knn_job* ct_job[GDM_KNN_MAX_JOBS];
    int cblocks = 0;
    int order[GDM_KNN_MAX_JOBS], n = 0;
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].K > 1) order[n++] = i;
    if (n == 0) return 0;
    for (int a = 1; a < n; ++a)                          // insertion sort by support size, descending (stable)
        for (int c = a; c > 0 && jobs[order[c]].S > jobs[order[c - 1]].S; --c) {
            const int tmp = order[c];
            order[c] = order[c - 1];
            order[c - 1] = tmp;
        }
    PackTable pk;
    pk.n = 0;
    pk.B = B;
    RangeTable rt;
    rt.n = 0;
    rt.B = B;
    const gdm_knn_job* rt_job[GDM_KNN_MAX_JOBS];
    const gdm_knn_job* pk_job[GDM_KNN_MAX_JOBS];
    int pack_blocks = 0, range_blocks = 0;
    size_t used = 0;
    int nblocks = 0, gblocks = 0;
    for (int a = 0; a < n; ++a) {
        const gdm_knn_job& jb = jobs[order[a]];
        if (workspace && is_grid_job(jb)) {               // organised support: window search
            int e = 0;
            while (e < rt.n && !same_support(*rt_job[e], jb)) ++e;
            bool ok = e < rt.n;
            if (!ok) {
                const size_t need = (ranges_bytes(jb, B) + 15) & ~(size_t)15;
                if (used + need <= workspace_bytes) {
                    RangeEntry& re = rt.e[rt.n];
                    re.support = jb.support;
                    re.support_bstride = jb.support_bstride;
                    re.W = jb.grid_w;
                    re.H = jb.S / jb.grid_w;
                    re.ranges = (float*)((char*)workspace + used);
                    re.nblocks = B * gdm_cdiv(re.H, KG_ROWS);
                    range_blocks += re.nblocks;
                    rt_job[rt.n++] = &jb;
                    used += need;
                    ok = true;
                }
            }
            if (ok) {
                KnnJobDev& d = gtab.jobs[gtab.njobs++];
                fill_job(d, jb);
                d.logT = 0;
                d.grid_w = jb.grid_w;
                d.grid_h = jb.S / jb.grid_w;
                d.ranges = rt.e[e].ranges;
                d.blocks_per_b = gdm_cdiv(d.Q, KW_WAVES);
                d.block_begin = gblocks;
                gblocks += d.blocks_per_b * B;
                continue;
            }
        }
        if (workspace && is_cell_job(jb)) {               // large unorganised support: cell lists
            int e = 0;
            while (e < ct.n && !same_support(*ct_job[e], jb)) ++e;
            bool ok = e < ct.n;
            if (!ok) {
                const size_t need = cells_bytes(jb.S, B);
                if (used + need <= workspace_bytes) {
                    CellEntry& ce = ct.e[ct.n];
                    ce.support = jb.support;
                    ce.support_bstride = jb.support_bstride;
                    ce.S = jb.S;
                    ce.sorted = (float4*)((char*)workspace + used);
                    ce.meta = (int*)((char*)workspace + used + (size_t)B * jb.S * sizeof(float4));
                    ct_job[ct.n++] = &jb;
                    used += need;
                    ok = true;
                }
            }
            if (ok) {
                KnnTable& tt = ctab;
                int& nb = cblocks;
                KnnJobDev& d = tt.jobs[tt.njobs++];
                fill_job(d, jb);
                d.logT = 0;
                d.packed = ct.e[e].sorted;
                d.ranges = reinterpret_cast<const float*>(ct.e[e].meta);
                d.grid_w = kc_grid(jb.S);
                d.blocks_per_b = gdm_cdiv(d.Q, KW_WAVES);
                d.block_begin = nb;
                nb += d.blocks_per_b * B;
                continue;
            }
        }
        KnnJobDev& d = tab.jobs[tab.njobs++];
        fill_job(d, jb);
        d.logT = 0;
        d.blocks_per_b = gdm_cdiv(d.Q, KW_WAVES);
        d.block_begin = nblocks;
        nblocks += d.blocks_per_b * B;
        if (!workspace) continue;
        int e = 0;                                        // a support set already packed for an earlier job?
        while (e < pk.n && !same_support(*pk_job[e], jb)) ++e;
        if (e == pk.n) {
            const size_t need = packed_bytes(jb.S, B);
            if (used + need > workspace_bytes) continue;  // does not fit: this job fills its tiles from the [S][3] array
            PackEntry& pe = pk.e[pk.n];
            pe.support = jb.support;
            pe.support_bstride = jb.support_bstride;
            pe.S = jb.S;
            tile_geometry(jb.S, pe.ntiles, pe.npad);
            pe.packed = (float4*)((char*)workspace + used);
            pe.block_begin = pack_blocks;
            pack_blocks += (int)gdm_cdiv((long)B * pe.ntiles * pe.npad, 256);
            pk_job[pk.n++] = &jb;
            used += need;
        }
        d.packed = pk.e[e].packed;
    }
    int rc;
    if (rt.n) {
        // every word of the range tables is written by the kernel itself: no zero fill
        hipLaunchKernelGGL(knn_grid_ranges_kernel, dim3(range_blocks), dim3(256), 0, stream, rt);
        if ((rc = gdm_launch_status("knn_grid_ranges_kernel"))) return rc;
    }
    if (ct.n) {
        hipLaunchKernelGGL(knn_cells_bin_kernel, dim3(ct.n * B), dim3(KC_BIN_THREADS), 0, stream, ct);
        if ((rc = gdm_launch_status("knn_cells_bin_kernel"))) return rc;
    }
    if (pk.n) {
        hipLaunchKernelGGL(knn_pack_kernel, dim3(pack_blocks), dim3(256), 0, stream, pk);
        if ((rc = gdm_launch_status("knn_pack_kernel"))) return rc;
    }
    if (ctab.njobs) {
        hipLaunchKernelGGL(knn_cells_kernel, dim3(cblocks), dim3(KW_BLOCK), 0, stream, ctab);
        if ((rc = gdm_launch_status("knn_cells_kernel"))) return rc;
    }
    if (tab.njobs) {
        hipLaunchKernelGGL(knn_wave_kernel, dim3(nblocks), dim3(KW_BLOCK), 0, stream, tab);
        if ((rc = gdm_launch_status("knn_wave_kernel"))) return rc;
    }
    if (gtab.njobs) {
        hipLaunchKernelGGL(knn_grid_kernel, dim3(gblocks), dim3(KW_BLOCK), 0, stream, gtab);
        if ((rc = gdm_launch_status("knn_grid_kernel"))) return rc;
    }
    return 0;
}

} // namespace

static int check_jobs(const gdm_knn_job* jobs, int njobs, int B, const char* who)
{
    GDM_CHECK_ARG(jobs && njobs >= 0 && njobs <= GDM_KNN_MAX_JOBS, "%s: njobs=%d out of range", who, njobs);
    GDM_CHECK_ARG(B >= 1, "%s: B=%d", who, B);
    for (int i = 0; i < njobs; ++i) {
        const gdm_knn_job& j = jobs[i];
        GDM_CHECK_ARG(j.support && j.query && j.idx, "%s: job %d has a NULL pointer", who, i);
        GDM_CHECK_ARG(j.S >= 1 && j.Q >= 1, "%s: job %d S=%d Q=%d", who, i, j.S, j.Q);
        GDM_CHECK_ARG(j.K >= 1 && j.K <= 32, "%s: job %d K=%d not in [1,32]", who, i, j.K);
        GDM_CHECK_ARG(j.support_bstride >= (int64_t)j.S * 3 || B == 1, "%s: job %d support_bstride too small", who, i);
        GDM_CHECK_ARG(j.query_bstride >= (int64_t)j.Q * 3 || B == 1, "%s: job %d query_bstride too small", who, i);
    }
    return 0;
}

#ifdef GDM_KNN_STATS
extern "C" int gdm_knn_stats_read(unsigned long long* out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gdm_knn_stats_dev), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gdm_knn_stats_dev), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

extern "C" size_t gdm_knn_jobs_workspace_bytes(const gdm_knn_job* jobs, int njobs, int B)
{
    if (!jobs || njobs < 0 || njobs > GDM_KNN_MAX_JOBS || B < 1) return 0;
    size_t total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].K <= 1) continue;
        bool seen = false;
        for (int e = 0; e < i && !seen; ++e)
            seen = jobs[e].K > 1 && same_support(jobs[e], jobs[i]) && is_grid_job(jobs[e]) == is_grid_job(jobs[i]);
        if (seen) continue;
        // (a cell job is sized for both layouts: it falls back to the hashed tiles when the cell lists no longer fit)
        total += is_grid_job(jobs[i]) ? ((ranges_bytes(jobs[i], B) + 15) & ~(size_t)15)
                                      : (is_cell_job(jobs[i]) ? cells_bytes(jobs[i].S, B) : packed_bytes(jobs[i].S, B));
    }
    return total;
}

extern "C" int gdm_knn_jobs_ws_hip(const gdm_knn_job* jobs, int njobs, int B, void* workspace, size_t workspace_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_jobs(jobs, njobs, B, "gdm_knn_jobs_ws_hip"))) return rc;
    GDM_CHECK_ARG(workspace || workspace_bytes == 0, "gdm_knn_jobs_ws_hip: NULL workspace with workspace_bytes=%zu", workspace_bytes);
    GDM_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "gdm_knn_jobs_ws_hip: workspace must be 16-byte aligned");
    if ((rc = launch_k1(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_wave(jobs, njobs, B, workspace, workspace_bytes, stream))) return rc;
    return 0;
}

extern "C" int gdm_knn_jobs_hip(const gdm_knn_job* jobs, int njobs, int B, void* stream_)
{
    return gdm_knn_jobs_ws_hip(jobs, njobs, B, nullptr, 0, stream_);
}

extern "C" int gdm_knn_batch_hip(const float* support, const float* query, int B, int S, int Q, int K,
                                 int32_t* idx, float* d2, void* stream)
{
    gdm_knn_job j;
    j.support = support;
    j.query = query;
    j.idx = idx;
    j.d2 = d2;
    j.support_bstride = (int64_t)S * 3;
    j.query_bstride = (int64_t)Q * 3;
    j.S = S;
    j.Q = Q;
    j.K = K;
    j.grid_w = 0;
    return gdm_knn_jobs_hip(&j, 1, B, stream);
}

// Host-pointer drop-in with the reference's exact argument list (knn_.h:17-19).
extern "C" void gdm_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                              const float* queries, size_t nqueries, size_t K, long* batch_indices)
{
    if (dim != 3 || K < 1 || K > 32 || batch_size < 1 || npts < 1 || nqueries < 1) {
        gdm_set_error("gdm_knn_batch: unsupported shape (dim=%zu K=%zu B=%zu S=%zu Q=%zu)", dim, K, batch_size, npts, nqueries);
        return;
    }
    const size_t sb = batch_size * npts * 3 * sizeof(float);
    const size_t qb = batch_size * nqueries * 3 * sizeof(float);
    const size_t ib = batch_size * nqueries * K * sizeof(int32_t);
    float *d_s = nullptr, *d_q = nullptr;
    int32_t* d_i = nullptr;
    hipStream_t stream = nullptr;
    bool ok = hipStreamCreate(&stream) == hipSuccess;
    ok = ok && hipMalloc((void**)&d_s, sb) == hipSuccess && hipMalloc((void**)&d_q, qb) == hipSuccess &&
         hipMalloc((void**)&d_i, ib) == hipSuccess;
    std::vector<int32_t> h_i;
    if (ok) {
        h_i.resize(batch_size * nqueries * K);
        ok = hipMemcpyAsync(d_s, batch_data, sb, hipMemcpyHostToDevice, stream) == hipSuccess &&
             hipMemcpyAsync(d_q, queries, qb, hipMemcpyHostToDevice, stream) == hipSuccess &&
             gdm_knn_batch_hip(d_s, d_q, (int)batch_size, (int)npts, (int)nqueries, (int)K, d_i, nullptr, stream) == 0 &&
             hipMemcpyAsync(h_i.data(), d_i, ib, hipMemcpyDeviceToHost, stream) == hipSuccess &&
             hipStreamSynchronize(stream) == hipSuccess;
    }
    if (ok) {
        for (size_t i = 0; i < h_i.size(); ++i) batch_indices[i] = (long)h_i[i];
    } else {
        gdm_set_error("gdm_knn_batch: HIP runtime failure (no GPU, or out of memory); output untouched");
    }
    if (d_s) (void)hipFree(d_s);
    if (d_q) (void)hipFree(d_q);
    if (d_i) (void)hipFree(d_i);
    if (stream) (void)hipStreamDestroy(stream);
}
