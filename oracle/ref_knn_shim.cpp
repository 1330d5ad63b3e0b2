// ORACLE -- test infrastructure only (see oracle/README.md).
//
// C-linkage doorway into the REAL reference kNN, compiled from the sources
// where they lie under /root/reference (never copied into this repo):
//   /root/reference/models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135
//   declared in .../knn_.h:17-19
// The shim only forwards; all arithmetic is the reference's own nanoflann.
// Built by oracle/Makefile into oracle/_ref/libknn_ref.so (git-ignored).
#include <cstddef>
#include "knn_.h"

extern "C" void ref_knn_batch_omp(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                                  const float* queries, size_t nqueries, size_t K, long* batch_indices)
{
    cpp_knn_batch_omp(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}

extern "C" void ref_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                              const float* queries, size_t nqueries, size_t K, long* batch_indices)
{
    cpp_knn_batch(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}
