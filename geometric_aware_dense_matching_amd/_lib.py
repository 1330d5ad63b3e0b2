"""ctypes binding of libgdm_hip.so (the C ABI declared in include/gdm.h).

The product path has no CPU fallback: if the shared library is missing or a call fails,
this module raises.  `build()` compiles it in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libgdm_hip.so")
CSRC = os.path.join(_PKG, "csrc")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t


class KnnJob(ctypes.Structure):
    """struct gdm_knn_job (include/gdm.h)."""
    _fields_ = [("support", _vp), ("query", _vp), ("idx", _vp), ("d2", _vp),
                ("support_bstride", ctypes.c_int64), ("query_bstride", ctypes.c_int64),
                ("S", ctypes.c_int32), ("Q", ctypes.c_int32), ("K", ctypes.c_int32), ("grid_w", ctypes.c_int32)]


# name -> (restype, argtypes); must list every symbol include/gdm.h declares
SIGNATURES = {
    "gdm_last_error": (ctypes.c_char_p, []),
    "gdm_version": (_i, []),
    "gdm_knn_batch": (None, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "gdm_knn_batch_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_knn_jobs_hip": (_i, [ctypes.POINTER(KnnJob), _i, _i, _vp]),
    "gdm_knn_jobs_workspace_bytes": (_sz, [ctypes.POINTER(KnnJob), _i, _i]),
    "gdm_knn_jobs_ws_hip": (_i, [ctypes.POINTER(KnnJob), _i, _i, _vp, _sz, _vp]),
    "gdm_ballquery_hip": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "gdm_furthestsampling_hip": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "gdm_interpolation_forward_hip": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gdm_interpolation_backward_hip": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gdm_labelstat_ballrange_hip": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "gdm_labelstat_idx_hip": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gdm_group_gather_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_group_gather_bwd_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_group_gather_bwd2_hip": (_i, [_vp, ctypes.c_long, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_gather_max_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_gather_max_bwd_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_gather_nn_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_gather_nn_bwd_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_rel_pos_enc_hip": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gdm_att_pool_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_att_pool_bwd_hip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_match_workspace_bytes": (_sz, [_i, _i, _i]),
    "gdm_match_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gdm_match_rows_bytes": (_sz, [_i]),
    "gdm_match_partial_bytes": (_sz, [_i, _i]),
    "gdm_match_pack_hip": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_match_pack2_hip": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "gdm_match_packed_hip": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gdm_seg_mask_hip": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "gdm_upsample_bilinear_hip": (_i, [_vp, ctypes.c_long, _i, _i, _i, _i, _vp, _vp]),
    "gdm_upsample_bilinear_bwd_hip": (_i, [_vp, ctypes.c_long, _i, _i, _i, _i, _vp, _vp]),
    "gdm_topk_rows_hip": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _vp]),
    "gdm_topk_negdist_hip": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gdm_affine_act_maxk_hip": (_i, [_vp, _vp, _vp, ctypes.c_long, _i, ctypes.c_long, _i, _i, _f, _vp, _vp]),
    "gdm_edge_feature_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_edge_feature_bwd_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_circle_rows_fwd_hip": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "gdm_circle_rows_bwd_hip": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "gdm_circle_match_rows_bytes": (_sz, [_i]),
    "gdm_circle_match_tp_bytes": (_sz, [_i]),
    "gdm_circle_match_pack_hip": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "gdm_circle_match_nbr_hip": (_i, [_vp, _i, _f, _vp, _vp]),
    "gdm_circle_match_visbits_hip": (_i, [_vp, _i, _i, _vp, _vp]),
    "gdm_circle_match_fwd_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "gdm_circle_match_bwd_parts": (_i, [_i, _i]),
    "gdm_circle_match_bwd_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gdm_kabsch_stats_hip": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "gdm_kabsch_solve_hip": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "gdm_lfa_stage_hip": (_i, [_vp] * 13 + [_i, _i, _i, _i, _i, _f, _vp, _vp]),
    "gdm_affine_act_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _i, ctypes.c_long, _i, _f, _vp, _vp]),
    "gdm_prelu1_hip": (_i, [_vp, _vp, ctypes.c_long, _vp, _vp]),
    "gdm_prelu1_bwd_hip": (_i, [_vp, _vp, _vp, ctypes.c_long, _vp, _vp, _vp]),
    "gdm_upconv3x3_gather_hip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "gdm_upconv3x3_gather2_hip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "gdm_upconv3x3_gather_bwd_hip": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_upconv_fused64_weight_bytes": (_sz, []),
    "gdm_upconv_fused64_pack_weight_hip": (_i, [_vp, _vp, _vp]),
    "gdm_upconv_fused64_hip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "gdm_psp_combine_hip": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_psp_combine2_hip": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_gather_add_affine_act_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "gdm_conv1x1_gather_add_act_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_long, _i, _f, _vp, _vp]),
    "gdm_conv1x1_gather_add_act2_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_long, _i, _f, _i, _vp, _vp]),
    "gdm_pack_rows64_hip": (_i, [_vp, _i, _vp, _vp]),
    "gdm_conv64_gather_add_act_mfma_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, ctypes.c_long, _i, _f, _i, _i, _vp, _vp]),
    "gdm_conv64_gather_add_act_mfma2_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, ctypes.c_long, _i, _f, _i, _i, _vp, _vp, _i, _vp]),
    "gdm_point_heads_hip": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gdm_point_heads2_hip": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gdm_upconv_final_points_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_conv3x3_act_bytes": (_sz, [_i, _i, _i, _i]),
    "gdm_conv3x3_weight_bytes": (_sz, [_i, _i]),
    "gdm_conv3x3_pack_weight_hip": (_i, [_vp, _i, _i, _vp, _vp]),
    "gdm_conv3x3_pack_act_hip": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_conv3x3_packed_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_conv3x3_packed2_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_conv3x3_strided_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gdm_conv1x1_strided_hip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_affine_relu_maxpool_hip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_conv1x1_logsoftmax_hip": (_i, [_vp, _vp, _vp, _i, _i, ctypes.c_long, _vp, _vp]),
    "gdm_psp_pools_hip": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gdm_bn_sums_len": (ctypes.c_long, [_i, _i, ctypes.c_long]),
    "gdm_bn_stats_hip": (_i, [_vp, _i, _i, ctypes.c_long, _vp, _vp]),
    "gdm_bn_fwd_apply_hip": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, ctypes.c_long, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "gdm_bn_bwd_reduce_hip": (_i, [_vp, _vp, _vp, _i, _i, ctypes.c_long, _i, _f, _vp, _vp]),
    "gdm_bn_bwd_apply_hip": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, ctypes.c_long, _i, _f, _vp, _vp, _vp, _vp]),
    "gdm_psp_pools_bwd_hip": (_i, [_vp, _vp, _vp, _vp, ctypes.c_long, _i, _i, _vp, _vp]),
    "gdm_conv1x1_weight_bytes": (_sz, [_i, _i]),
    "gdm_conv1x1_pack_weight_hip": (_i, [_vp, _i, _i, _vp, _vp]),
    "gdm_conv_pack_weight_dgrad_hip": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "gdm_conv1x1_packed_hip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_depth_to_xyz_hip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_spline_aggregate_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_spline_aggregate_bwd_hip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "gdm_spline_direct_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "gdm_gemm_grouped_hip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdm_spline_pairs_aggregate_hip": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
}



class PwSeg(ctypes.Structure):
    """gdm_pw_seg (include/gdm.h)."""
    _fields_ = [("x", _vp), ("idx", _vp), ("C", ctypes.c_int32), ("n_src", ctypes.c_int32)]


class PwJob(ctypes.Structure):
    """gdm_pw_job (include/gdm.h)."""
    _fields_ = [("x", _vp), ("wt", _vp), ("out", _vp), ("n", ctypes.c_int32)]


class CopyJob(ctypes.Structure):
    """gdm_copy_job (include/gdm.h)."""
    _fields_ = [("dst", _vp), ("src", _vp), ("sb", ctypes.c_int64), ("s1", ctypes.c_int64), ("s2", ctypes.c_int64),
                ("B", ctypes.c_int32), ("R1", ctypes.c_int32), ("R2", ctypes.c_int32), ("E", ctypes.c_int32)]


SIGNATURES["gdm_circle_match_nbr_items_hip"] = (_i, [_vp, _i, _vp, _i, _vp, _vp])
SIGNATURES["gdm_circle_match_fwd2_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp])
SIGNATURES["gdm_circle_match_bwd2_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["gdm_stem_weight_bytes"] = (_sz, [])
SIGNATURES["gdm_stem_pack_weight_hip"] = (_i, [_vp, _vp, _vp])
SIGNATURES["gdm_stem_hip"] = (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["gdm_wgrad_x_bytes"] = (_sz, [_i, _i, _i, _i])
SIGNATURES["gdm_wgrad_go_bytes"] = (_sz, [_i, _i, _i, _i])
SIGNATURES["gdm_wgrad_pack_x_hip"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp])
SIGNATURES["gdm_wgrad_pack_go_hip"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp])
SIGNATURES["gdm_conv1x1_packed_wb_hip"] = (_i, [_vp, _vp, ctypes.c_long, _i, _i, _i, _i, _i, _vp, _vp])
SIGNATURES["gdm_gather_add_affine_act2_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp])
SIGNATURES["gdm_spline_direct3_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
SIGNATURES["gdm_spline_pairs_aggregate3_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp])
SIGNATURES["gdm_spline_direct2_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["gdm_spline_pairs_aggregate2_hip"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["gdm_wgrad_x1_bytes"] = (_sz, [_i, _i, _i])
SIGNATURES["gdm_wgrad_pack_x1_hip"] = (_i, [_vp, _i, _i, _i, _vp, _vp])
SIGNATURES["gdm_wgrad_direct_hip"] = (_i, [_vp, ctypes.c_long, _vp, ctypes.c_long, _i, _i, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["gdm_mfma_probe_hip"] = (_i, [_i, _i, _i, _vp, _vp])
SIGNATURES["gdm_mfma_probe_lds_hip"] = (_i, [_i, _i, _i, _vp, _vp])
SIGNATURES["gdm_copy_jobs_hip"] = (_i, [ctypes.POINTER(CopyJob), _i, _vp])
SIGNATURES["gdm_pointwise_chain2_hip"] = (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["gdm_pointwise_jobs_hip"] = (_i, [ctypes.POINTER(PwJob), _i, _i, _i, _i, _vp])
SIGNATURES["gdm_pointwise2_hip"] = (_i, [ctypes.POINTER(PwSeg), _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp])
SIGNATURES["gdm_pointwise_hip"] = (_i, [ctypes.POINTER(PwSeg), _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp])

_lib = None


def _objects_state():
    st = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".o"):
            st[f] = os.stat(os.path.join(CSRC, f)).st_mtime_ns
    return st


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of libgdm_hip.so (cross-compiles without a GPU).  Writes csrc/build_record.json: which
    translation units this call recompiled (`make` decides by time stamps) and the library's size / SHA-256 -- the evidence of what
    was built, read back by `build_record()` (bench.py prints it with its line)."""
    import hashlib
    import json
    import time
    before = _objects_state()
    out = None if verbose else subprocess.DEVNULL
    t0 = time.time()
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    after = _objects_state()
    rebuilt = sorted(f for f, m in after.items() if before.get(f) != m)
    with open(LIB_PATH, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()[:16]
    rec = {"mode": "hipcc --offload-arch=gfx950 via csrc/Makefile (in-tree, ahead of time)", "recompiled": rebuilt,
           "up_to_date": sorted(set(after) - set(rebuilt)), "lib": os.path.basename(LIB_PATH), "lib_bytes": os.path.getsize(LIB_PATH),
           "lib_sha256_16": digest, "seconds": round(time.time() - t0, 1), "when": time.strftime("%Y-%m-%dT%H:%M:%S")}
    try:
        with open(os.path.join(CSRC, "build_record.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass
    print("[gdm build] recompiled %d of %d translation units%s; %s %d bytes sha256 %s" %
          (len(rebuilt), len(after), (": " + " ".join(rebuilt)) if rebuilt else "", rec["lib"], rec["lib_bytes"], digest))
    return LIB_PATH


def build_record():
    """What `build()` last recorded, plus the SHA-256 of the library that is actually loaded (they must agree)."""
    import hashlib
    import json
    rec = {}
    try:
        with open(os.path.join(CSRC, "build_record.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {"mode": "prebuilt library (no build record travelled)"}
    try:
        with open(LIB_PATH, "rb") as f:
            rec["loaded_lib_sha256_16"] = hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        rec["loaded_lib_sha256_16"] = None
    rec.pop("up_to_date", None)
    if rec.get("lib_sha256_16") is not None:                    # the record describes ANOTHER build (e.g. `make` run by hand afterwards)
        rec["record_matches_loaded_lib"] = rec["lib_sha256_16"] == rec["loaded_lib_sha256_16"]
    return rec


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libgdm_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C geometric_aware_dense_matching_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # torch FIRST: PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so) and the process must hold exactly one.  Loaded
        # before torch, this library pulls in /opt/rocm's runtime instead, torch then brings its own, and the first launch from here
        # fails with "no ROCm-capable device is detected" (build() followed by smoke() in one process did exactly that).
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class GdmError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib().gdm_last_error()
        raise GdmError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
