// Exact K-nearest-neighbour search on gfx950 (MI355X): brute force, fp32, canonical order.
//
// Replaces (reference, /root/reference):
//   models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135  cpp_knn_batch_omp
//   nanoflann.hpp:323-348 (distance arithmetic), :115-139 (result set)
//   22 calls per crop from datasets/lm/linemod_pbr.py:534-569
//
// Design (HBM/latency-bound integer-and-compare work, no MFMA):
//   * one launch serves a whole TABLE of independent searches (all pyramid calls x all crops);
//     the table travels as a by-value kernel argument, so a launch needs no device allocation
//     and can be captured in a hipGraph.
//   * a query is owned by T = 2^t lanes of one wave (T chosen per job from the support size):
//     lane t scans support points t, t+T, ... from an LDS tile (float4 per point, one
//     ds_read_b128 per pair; lanes of the same query read consecutive slots, lanes of different
//     queries read the same slot -> broadcast), keeps its own sorted top-KMAX in registers, and
//     the T lists are merged with wave shuffles (K rounds of lexicographic arg-min).
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

template <int G>
__device__ __forceinline__ void knn_group_body(const KnnJobDev& job, int local, float4* tile)
{
    const int b = local / job.blocks_per_b;
    const int qb = local - b * job.blocks_per_b;
    const int S = job.S, Q = job.Q, K = job.K;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = tid & (G - 1);                     // lane within the group
    const int gbase = lane & ~(G - 1);               // first wave-lane of the group
    const int q = qb * (KNN_BLOCK / G) + tid / G;
    const bool valid = q < Q;
    const int qc = valid ? q : Q - 1;
    const float* sup = job.support + (long long)b * job.support_bstride;
    const float* qry = job.query + (long long)b * job.query_bstride + (long long)qc * 3;
    const float qx = qry[0], qy = qry[1], qz = qry[2];
    const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    const unsigned long long kmask = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);

    float ld = INFINITY;                             // this lane's list element
    int li = IDX_EMPTY;
    float worst = INFINITY;                          // list[K-1], identical in all lanes of the group
    int worst_i = IDX_EMPTY;

    // Tile t holds the support points t, t + ntiles, t + 2 ntiles, ...: every tile is a uniform subsample of the
    // whole set.  Support sets here are often pixel grids in raster order; scanned in that order the distance to a
    // query falls monotonically for half the scan and nearly every point would be inserted.  With subsampled tiles
    // the K-th distance is tight after the first tile.  Order is then not index order, so ties are resolved by an
    // explicit (d2, index) comparison.
    const int ntiles = (S + KNN_TILE - 1) / KNN_TILE;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        for (int p = tid; p < KNN_TILE; p += KNN_BLOCK) {
            const int gi = p * ntiles + t;
            float4 v;
            if (gi < S) {
                const float* s3 = sup + (long long)gi * 3;
                v = make_float4(s3[0], s3[1], s3[2], __int_as_float(gi));
            } else {
                v = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(IDX_EMPTY));
            }
            tile[p] = v;
        }
        __syncthreads();
        const int npt = (S - t + ntiles - 1) / ntiles;               // valid slots in this tile
        const int steps = (npt + G - 1) / G;
        // four steps' candidates are fetched and their distances formed before any of them is considered: the read and the
        // dependent arithmetic chain of a step (~150 cycles of latency) would otherwise sit in front of every ballot
        for (int s = 0; s < steps; s += 4) {
            float4 v4[4];
            float d4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = tile[min(s + u, KNN_TILE / G - 1) * G + g];
#pragma unroll
            for (int u = 0; u < 4; ++u) d4[u] = dist2_ref(qx, qy, qz, v4[u].x, v4[u].y, v4[u].z);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = d4[u];
                const int di = __float_as_int(v4[u].w);
                bool pass = (s + u < steps) && (d < worst || (d == worst && di < worst_i));
                unsigned long long bal = __ballot(pass);
                while (bal) {                            // wave-uniform loop
                    const unsigned long long m = (bal >> gbase) & gmask;
                    const bool act = m != 0ull;          // group-uniform
                    const int src = act ? __builtin_ctzll(m) : 0;
                    const float cd = __shfl(d, gbase + src, 64);
                    const int ci = __shfl(di, gbase + src, 64);
                    const float ud = __shfl_up(ld, 1, G);
                    const int ui = __shfl_up(li, 1, G);
                    const bool before = ld < cd || (ld == cd && li < ci);
                    const unsigned long long le = (__ballot(before) >> gbase) & gmask & kmask;
                    const int pos = __builtin_popcountll(le);        // entries that stay in front of the candidate
                    if (act) {
                        if (g > pos) {
                            ld = ud;
                            li = ui;
                        } else if (g == pos) {
                            ld = cd;
                            li = ci;
                        }
                    }
                    worst = __shfl(ld, gbase + K - 1, 64);
                    worst_i = __shfl(li, gbase + K - 1, 64);
                    pass = pass && !(act && g == src) && (d < worst || (d == worst && di < worst_i));
                    bal = __ballot(pass);
                }
            }
        }
    }
    if (valid && g < K) {
        int32_t* out_i = job.idx + ((long long)b * Q + q) * K;
        out_i[g] = li == IDX_EMPTY ? 0 : li;
        if (job.d2) job.d2[((long long)b * Q + q) * K + g] = isinf(ld) ? 3.402823466e+38f : ld;
    }
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
    if (job.logT < 0) {                               // query-rich job: lane-distributed list, same launch
        if (KMAX == 16) knn_group_body<16>(job, local, tile);
        else if (KMAX == 32) knn_group_body<32>(job, local, tile);
        return;
    }
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

// ---- K > 1: one query per group of G lanes, the sorted top-K list DISTRIBUTED over the group's lanes ----
// Lane g of a group holds the g-th best (d2, index) so far.  Each step the G lanes evaluate G consecutive
// support points; candidates below the current K-th distance are inserted one at a time, lowest lane (= lowest
// index) first: position = popcount(ballot(list <= cand)), the tail shifts up by one lane (__shfl_up) and the
// K-th distance is re-broadcast.  An insertion costs ~15 wave instructions instead of the ~80 of a per-lane
// register list, and the admission threshold is the query's GLOBAL K-th distance, so insertions happen
// ~K(1+ln(S/K)) times per query instead of that many times per lane.
template <int G>
__global__ __launch_bounds__(KNN_BLOCK) void knn_group_kernel(const KnnTable tab)
{
    __shared__ float4 tile[KNN_TILE];
    const int bid = blockIdx.x;
    int j = 0;
    while (j + 1 < tab.njobs && bid >= tab.jobs[j + 1].block_begin) ++j;
    knn_group_body<G>(tab.jobs[j], bid - tab.jobs[j].block_begin, tile);
}

int kmax_class(int K)
{
    if (K <= 1) return 1;
    if (K <= 8) return 8;
    if (K <= 16) return 16;
    return 32;
}

int pick_logT(int S)
{
    // every lane scans >= 128 support points where possible; T in {1,2,...,64}
    int logT = 0;
    while (logT < 6 && (S >> (logT + 1)) >= 128) ++logT;
    return logT;
}

int knn_kernel_version();
int group_class(int K) { return K <= 16 ? 16 : 32; }

// Which kernel serves a K > 1 job: the lane-distributed list needs one group per query, so with few queries
// against a large support set (e.g. 16384 pixels -> 128 points) it leaves the chip idle; those jobs keep the
// per-lane-list kernel, which splits ONE query's support over up to 64 lanes.
bool use_group_kernel(const gdm_knn_job& j, int B)
{
    if (knn_kernel_version() != 2 || j.K < 2) return false;   // separate-launch form only in mode 2
    return (long)B * j.Q >= 4096 || j.S <= 512;
}

template <int G>
int launch_group(const gdm_knn_job* jobs, int njobs, int B, hipStream_t stream)
{
    KnnTable tab;
    tab.njobs = 0;
    tab.B = B;
    int nblocks = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!use_group_kernel(jobs[i], B) || group_class(jobs[i].K) != G) continue;
        KnnJobDev& d = tab.jobs[tab.njobs++];
        d.support = jobs[i].support;
        d.query = jobs[i].query;
        d.idx = jobs[i].idx;
        d.d2 = jobs[i].d2;
        d.support_bstride = jobs[i].support_bstride;
        d.query_bstride = jobs[i].query_bstride;
        d.S = jobs[i].S;
        d.Q = jobs[i].Q;
        d.K = jobs[i].K;
        d.logT = 0;
        d.blocks_per_b = gdm_cdiv(d.Q, KNN_BLOCK / G);
        d.block_begin = nblocks;
        nblocks += d.blocks_per_b * B;
    }
    if (tab.njobs == 0) return 0;
    hipLaunchKernelGGL(knn_group_kernel<G>, dim3(nblocks), dim3(KNN_BLOCK), 0, stream, tab);
    return gdm_launch_status("knn_group_kernel");
}

// 1 = per-lane register lists, every K > 1 job of a batch in ONE launch (default);
// 2 = lane-distributed list for the query-rich jobs (GDM_KNN_KERNEL=2).  Measured on MI355X (B=16 pyramid):
// v2 wins per job in isolation (2048x2048: 106 vs 182 us, 16384->512: 172 vs 343 us) but splitting the K=16 jobs over
// two serial launches loses the cross-job overlap: 1.18 ms vs 0.73 ms per pyramid.  Next step: one launch, both bodies.
int knn_kernel_version()
{
    const char* e = getenv("GDM_KNN_KERNEL");
    if (e && e[0] == '1') return 1;
    if (e && e[0] == '2') return 2;
    return 3;                                          // 3 = both bodies in one launch per K class
}

template <int KMAX>
int launch_class(const gdm_knn_job* jobs, int njobs, int B, hipStream_t stream)
{
    KnnTable tab;
    tab.njobs = 0;
    tab.B = B;
    int nblocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const bool mixed = knn_kernel_version() == 3 && (KMAX == 16 || KMAX == 32) && jobs[i].K >= 2 &&
                           ((long)B * jobs[i].Q >= 4096 || jobs[i].S <= 512) && group_class(jobs[i].K) == KMAX;
        if (mixed) {
            KnnJobDev& d = tab.jobs[tab.njobs++];
            d.support = jobs[i].support; d.query = jobs[i].query; d.idx = jobs[i].idx; d.d2 = jobs[i].d2;
            d.support_bstride = jobs[i].support_bstride; d.query_bstride = jobs[i].query_bstride;
            d.S = jobs[i].S; d.Q = jobs[i].Q; d.K = jobs[i].K;
            d.logT = -1;
            d.blocks_per_b = gdm_cdiv(d.Q, KNN_BLOCK / KMAX);
            d.block_begin = nblocks;
            nblocks += d.blocks_per_b * B;
            continue;
        }
        if (knn_kernel_version() == 3 && jobs[i].K >= 2 && kmax_class(jobs[i].K) != group_class(jobs[i].K) &&
            ((long)B * jobs[i].Q >= 4096 || jobs[i].S <= 512))
            continue;                                  // K <= 8 query-rich job: served by the KMAX=16 launch below
        if (kmax_class(jobs[i].K) != KMAX || use_group_kernel(jobs[i], B)) continue;
        KnnJobDev& d = tab.jobs[tab.njobs++];
        d.support = jobs[i].support;
        d.query = jobs[i].query;
        d.idx = jobs[i].idx;
        d.d2 = jobs[i].d2;
        d.support_bstride = jobs[i].support_bstride;
        d.query_bstride = jobs[i].query_bstride;
        d.S = jobs[i].S;
        d.Q = jobs[i].Q;
        d.K = jobs[i].K;
        d.logT = pick_logT(d.S);
        const int qpb = KNN_BLOCK >> d.logT;
        d.blocks_per_b = gdm_cdiv(d.Q, qpb);
        d.block_begin = nblocks;
        nblocks += d.blocks_per_b * B;
    }
    if (tab.njobs == 0) return 0;
    hipLaunchKernelGGL(knn_kernel<KMAX>, dim3(nblocks), dim3(KNN_BLOCK), 0, stream, tab);
    return gdm_launch_status("knn_kernel");
}

} // namespace

extern "C" int gdm_knn_jobs_hip(const gdm_knn_job* jobs, int njobs, int B, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    GDM_CHECK_ARG(jobs && njobs >= 0 && njobs <= GDM_KNN_MAX_JOBS, "gdm_knn_jobs_hip: njobs=%d out of range", njobs);
    GDM_CHECK_ARG(B >= 1, "gdm_knn_jobs_hip: B=%d", B);
    for (int i = 0; i < njobs; ++i) {
        const gdm_knn_job& j = jobs[i];
        GDM_CHECK_ARG(j.support && j.query && j.idx, "gdm_knn_jobs_hip: job %d has a NULL pointer", i);
        GDM_CHECK_ARG(j.S >= 1 && j.Q >= 1, "gdm_knn_jobs_hip: job %d S=%d Q=%d", i, j.S, j.Q);
        GDM_CHECK_ARG(j.K >= 1 && j.K <= 32, "gdm_knn_jobs_hip: job %d K=%d not in [1,32]", i, j.K);
        GDM_CHECK_ARG(j.support_bstride >= (int64_t)j.S * 3 || B == 1, "gdm_knn_jobs_hip: job %d support_bstride too small", i);
        GDM_CHECK_ARG(j.query_bstride >= (int64_t)j.Q * 3 || B == 1, "gdm_knn_jobs_hip: job %d query_bstride too small", i);
    }
    int rc;
    if ((rc = launch_class<1>(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_group<16>(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_group<32>(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_class<8>(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_class<16>(jobs, njobs, B, stream))) return rc;
    if ((rc = launch_class<32>(jobs, njobs, B, stream))) return rc;
    return 0;
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
