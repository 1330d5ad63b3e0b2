/*
 * ORACLE -- test infrastructure only.  Never imported, linked or executed by the
 * product path (geometric_aware_dense_matching_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the reference's exact K-nearest-neighbour search
 *   /root/reference/models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135
 *     (cpp_knn_batch_omp: per batch element, per query, K nearest support
 *      points, ascending squared L2 distance, int64 indices out)
 *   /root/reference/models/RandLA/utils/nearest_neighbors/nanoflann.hpp:323-348
 *     (L2_Adaptor::evalMetric: for dim=3 the "last 0-3 components" loop runs,
 *      result = ((0 + dx*dx) + dy*dy) + dz*dz in fp32, diff = query - point)
 *   /root/reference/models/RandLA/utils/nearest_neighbors/nanoflann.hpp:115-139
 *     (KNNResultSet::addPoint: sorted insertion, strict '>' shift, so equal
 *      distances keep arrival order; arrival order is KD-tree traversal order)
 *
 * The reference orders equal distances by tree traversal, which is not a
 * property of the data.  The oracle (and the HIP kernel) use the canonical
 * order (d2 ascending, then index ascending).  On tie-free inputs both orders
 * coincide bit for bit (pinned in tests/test_oracle_knn.py against the compiled
 * reference in oracle/_ref and against tests/golden/knn_pyramid_c1.npz).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is built by gcc -O2 for baseline
 * x86-64, which has no FMA, so every product and sum is rounded to fp32.
 */
#include <stddef.h>
#include <stdint.h>
#include <float.h>

static inline float d2_ref(const float *q, const float *p)
{
    /* nanoflann.hpp:343-346, three trips of the tail loop */
    float r = 0.0f;
    float d0 = q[0] - p[0];
    r += d0 * d0;
    float d1 = q[1] - p[1];
    r += d1 * d1;
    float d2 = q[2] - p[2];
    r += d2 * d2;
    return r;
}

/* one query against one support set; out_idx/out_d2 have K slots */
static void knn_one(const float *sup, size_t S, const float *q, size_t K,
                    int64_t *out_idx, float *out_d2)
{
    size_t count = 0;
    for (size_t s = 0; s < S; ++s) {
        float d = d2_ref(q, sup + 3 * s);
        /* canonical admission: (d, s) lexicographically below the current worst */
        if (count == K) {
            if (!(d < out_d2[K - 1])) continue; /* s is larger than any stored index */
        }
        size_t i = count < K ? count : K - 1;
        while (i > 0 && out_d2[i - 1] > d) { /* strict: equal d keeps lower index first */
            out_d2[i] = out_d2[i - 1];
            out_idx[i] = out_idx[i - 1];
            --i;
        }
        out_d2[i] = d;
        out_idx[i] = (int64_t)s;
        if (count < K) ++count;
    }
    for (size_t i = count; i < K; ++i) { /* K > S: reference leaves slots unwritten (knn.pyx zero-inits) */
        out_idx[i] = 0;
        out_d2[i] = FLT_MAX;
    }
}

/* Same argument list as cpp_knn_batch_omp (knn_.h:17-19), plus optional d2 out. */
void oracle_knn_batch(const float *batch_data, size_t batch_size, size_t npts, size_t dim,
                      const float *queries, size_t nqueries, size_t K,
                      int64_t *batch_indices, float *batch_d2 /* may be NULL */)
{
    if (dim != 3) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (size_t b = 0; b < batch_size; ++b) {
        for (size_t i = 0; i < nqueries; ++i) {
            float dtmp[64];
            int64_t itmp[64];
            if (K > 64) continue;
            knn_one(batch_data + b * npts * 3, npts, queries + (b * nqueries + i) * 3, K, itmp, dtmp);
            for (size_t k = 0; k < K; ++k) {
                batch_indices[(b * nqueries + i) * K + k] = itmp[k];
                if (batch_d2) batch_d2[(b * nqueries + i) * K + k] = dtmp[k];
            }
        }
    }
}

/* squared distances of given (query, index) pairs, for tie-group comparisons */
void oracle_knn_d2_of(const float *sup, const float *queries, size_t nqueries, size_t K,
                      const int64_t *idx, float *d2)
{
    for (size_t i = 0; i < nqueries; ++i)
        for (size_t k = 0; k < K; ++k)
            d2[i * K + k] = d2_ref(queries + 3 * i, sup + 3 * idx[i * K + k]);
}
