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
};

struct KnnTable {
    KnnJobDev jobs[GDM_KNN_MAX_JOBS];
    int njobs;
    int B;
};

__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float px, float py, float pz)
{
    // nanoflann.hpp:343-346 for dim == 3: result = ((0 + d0*d0) + d1*d1) + d2*d2, diff = query - point
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

// K in [2, 32] jobs: one query per wave.  Large jobs first, so that the tail of the launch is made of short blocks.
// With a workspace every distinct support set is re-laid out once (knn_pack_kernel) and shared by the jobs that search it.
int launch_wave(const gdm_knn_job* jobs, int njobs, int B, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    KnnTable tab;
    tab.njobs = 0;
    tab.B = B;
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
    int pack_blocks = 0;
    size_t used = 0;
    int nblocks = 0;
    for (int a = 0; a < n; ++a) {
        const gdm_knn_job& jb = jobs[order[a]];
        KnnJobDev& d = tab.jobs[tab.njobs++];
        fill_job(d, jb);
        d.logT = 0;
        d.blocks_per_b = gdm_cdiv(d.Q, KW_WAVES);
        d.block_begin = nblocks;
        nblocks += d.blocks_per_b * B;
        if (!workspace) continue;
        int e = 0;                                        // a support set already packed for an earlier job?
        while (e < pk.n && !(pk.e[e].support == jb.support && pk.e[e].S == jb.S && pk.e[e].support_bstride == jb.support_bstride)) ++e;
        if (e == pk.n) {
            const size_t need = packed_bytes(jb.S, B);
            if (used + need > workspace_bytes) continue;  // does not fit: this job fills its tiles from the [S][3] array
            PackEntry& pe = pk.e[pk.n++];
            pe.support = jb.support;
            pe.support_bstride = jb.support_bstride;
            pe.S = jb.S;
            tile_geometry(jb.S, pe.ntiles, pe.npad);
            pe.packed = (float4*)((char*)workspace + used);
            pe.block_begin = pack_blocks;
            pack_blocks += (int)gdm_cdiv((long)B * pe.ntiles * pe.npad, 256);
            used += need;
        }
        d.packed = pk.e[e].packed;
    }
    if (pk.n) {
        hipLaunchKernelGGL(knn_pack_kernel, dim3(pack_blocks), dim3(256), 0, stream, pk);
        const int rc = gdm_launch_status("knn_pack_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(knn_wave_kernel, dim3(nblocks), dim3(KW_BLOCK), 0, stream, tab);
    return gdm_launch_status("knn_wave_kernel");
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

extern "C" size_t gdm_knn_jobs_workspace_bytes(const gdm_knn_job* jobs, int njobs, int B)
{
    if (!jobs || njobs < 0 || njobs > GDM_KNN_MAX_JOBS || B < 1) return 0;
    size_t total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].K <= 1) continue;
        bool seen = false;
        for (int e = 0; e < i && !seen; ++e)
            seen = jobs[e].K > 1 && jobs[e].support == jobs[i].support && jobs[e].S == jobs[i].S &&
                   jobs[e].support_bstride == jobs[i].support_bstride;
        if (!seen) total += packed_bytes(jobs[i].S, B);
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
    j._pad = 0;
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
