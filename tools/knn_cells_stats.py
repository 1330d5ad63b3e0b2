"""Counters of the cell-list kNN search (build with -DGDM_KNN_STATS into tools/micro/variants/libgdm_knnstats.so): rings, row batches,
compactions and admitted candidates per query for the cloud's own K = 16 search of one synthetic batch.  Development aid (round 4)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geometric_aware_dense_matching_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "micro", "variants", "libgdm_knnstats.so")
import torch
from geometric_aware_dense_matching_amd import ops, pyramid, synthetic
L = _lib.lib()
L.gdm_knn_stats_read.restype = ctypes.c_int
L.gdm_knn_stats_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B, N = 16, 2048
batch = synthetic.make_batch(seed=100, batch=B, n_points=N)
cld = pyramid.cloud_from_inputs(torch.from_numpy(batch["cld_rgb_nrm"]).cuda())
buf = (ctypes.c_ulonglong * 8)()
L.gdm_knn_stats_read(buf, 1)
ops.knn_jobs([(cld, cld, 16)], B)
torch.cuda.synchronize()
L.gdm_knn_stats_read(buf, 1)
q = max(buf[0], 1)
print("queries %d: rings %.2f, row batches %.2f, extra steps %.2f, compactions %.2f, admitted %.1f per query" % (
    buf[0], buf[1] / q, buf[2] / q, buf[3] / q, buf[4] / q, buf[5] / q))
