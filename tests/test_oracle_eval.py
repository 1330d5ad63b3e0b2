"""CPU: the oracle's restatement of the reference's pose errors (oracle/eval_ref.py) against the vectors tests/golden/make_golden.py
produced by executing pose_error.py's re / te / arp_2d and pose_utils.py's get_closest_rot from the reference's own text."""
import os

import numpy as np

from oracle import eval_ref
from tests.golden import inputs as gin

G = os.path.join(os.path.dirname(__file__), "golden")


def test_re_te_arp2d_and_closest_symmetric_rotation_vs_reference_golden():
    from geometric_aware_dense_matching_amd.synthetic import LM_K
    g = np.load(os.path.join(G, "pose.npz"))
    pi = gin.pose_inputs()
    sym = gin.sym_rotations()
    K = LM_K.astype(np.float64)
    pts = pi["model"].astype(np.float64)
    for b in range(pi["idx"].shape[0]):
        T = g["RT"][b]
        Rg, tg = pi["RT"][b, :, :3].astype(np.float64), pi["RT"][b, :, 3].astype(np.float64)
        assert abs(eval_ref.re(T[:, :3], Rg) - g["re"][b]) < 1e-9
        assert abs(eval_ref.te(T[:, 3], tg) - g["te"][b]) < 1e-12 * max(1.0, g["te"][b])
        assert abs(eval_ref.arp_2d(T[:, :3], T[:, 3], Rg, tg, pts, K) - g["proj"][b]) < 1e-9 * max(1.0, g["proj"][b])
        Rg2 = Rg.dot(sym[1 + b % (sym.shape[0] - 1)])
        Rc = eval_ref.get_closest_rot(T[:, :3], Rg2, sym)
        assert abs(eval_ref.re(T[:, :3], Rc) - g["re_sym"][b]) < 1e-9
        assert abs(eval_ref.arp_2d(T[:, :3], T[:, 3], Rc, tg, pts, K) - g["proj_sym"][b]) < 1e-9 * max(1.0, g["proj_sym"][b])


def test_recall_flags_and_table_shape():
    f = eval_ref.recall_flags(0.004, 1.5, 0.01, 1.0, 0.1)
    assert list(f) == eval_ref.METRICS and f["ad_5"] == 1.0 and f["ad_2"] == 0.0 and f["rete_2"] == 1.0 and f["proj_2"] == 1.0
    rec = {"ape": {m: [1.0, 0.0] for m in eval_ref.METRICS}, "cat": {m: [] for m in eval_ref.METRICS}}
    err = {"ape": {"re": [1.0, 3.0], "te": [0.01, 0.03]}, "cat": {"re": [], "te": []}}
    tab = eval_ref.table(rec, err)
    assert tab[0] == ["objects", "ape", "cat", "Avg(2)"] and tab[1] == ["ad_2", "50.00", 0.0, "25.00"] and len(tab) == 1 + 16 + 2
    assert tab[-2][:2] == ["re", "2.00"] and np.isnan(tab[-2][2])


def test_product_recall_table_on_host_arrays_equals_the_oracle_table():
    """evaluation.RecallTable is host-side bookkeeping: fed with numpy errors it must reproduce oracle/eval_ref.table (the restatement of
    /root/reference/evaluator.py:408-463) string for string, ground truths without a prediction included."""
    from geometric_aware_dense_matching_amd import evaluation
    assert evaluation.METRICS == eval_ref.METRICS
    rs = np.random.RandomState(2)
    tab = evaluation.RecallTable()
    rec, err = {}, {}
    for name, diam, n in (("ape", 0.102, 23), ("can", 0.201, 11), ("eggbox", 0.165, 1)):
        e = dict(ad=np.abs(rs.randn(n)) * 0.02, re=np.abs(rs.randn(n)) * 6, te=np.abs(rs.randn(n)) * 0.06, proj=np.abs(rs.randn(n)) * 6)
        tab.update(name, e, diam)
        rec[name] = {m: [] for m in eval_ref.METRICS}
        err[name] = {"re": e["re"].tolist(), "te": e["te"].tolist()}
        for i in range(n):
            for m, v in eval_ref.recall_flags(e["ad"][i], e["re"][i], e["te"][i], e["proj"][i], diam).items():
                rec[name][m].append(v)
    tab.missing("can", 3)
    for m in eval_ref.METRICS:
        rec["can"][m] += [0.0] * 3
    assert tab.table() == eval_ref.table(rec, err)
    assert len(tab.format().splitlines()) == 19


def test_settings_switch_lists_are_consistent():
    from geometric_aware_dense_matching_amd import settings
    for name in settings.ALL_SWITCHES + settings.SPLIT_BF16_SWITCHES:
        assert isinstance(getattr(settings, name), bool), name
    assert set(settings.SPLIT_BF16_SWITCHES) <= set(settings.ALL_SWITCHES)
    doc = settings.__doc__
    for name in settings.ALL_SWITCHES:
        assert name in doc, "settings.py's table does not describe %s" % name


def test_precision_variant_ignores_undetected_and_has_no_absolute_ad_line():
    """evaluator.py:466-660 (`_eval_predictions_precision`): same flags, ground truths without a prediction skipped (:549-551), no
    "ad_0.1" line (:513-529)."""
    from geometric_aware_dense_matching_amd import evaluation
    rs = np.random.RandomState(5)
    e = dict(ad=np.abs(rs.randn(7)) * 0.02, re=np.abs(rs.randn(7)) * 6, te=np.abs(rs.randn(7)) * 0.06, proj=np.abs(rs.randn(7)) * 6)
    rec, prec = evaluation.RecallTable(), evaluation.RecallTable(precision=True)
    for t in (rec, prec):
        t.update("ape", e, 0.102)
        t.missing("ape", 7)
    assert [r[0] for r in prec.table()[1:]] == [m for m in eval_ref.METRICS if m != "ad_0.1"] + ["re", "te"]
    for m in evaluation.PRECISION_METRICS:
        assert len(prec.recalls["ape"][m]) == 7 and len(rec.recalls["ape"][m]) == 14
        assert abs(np.mean(prec.recalls["ape"][m]) - 2 * np.mean(rec.recalls["ape"][m])) < 1e-12      # half of the recall's entries are misses
        want = [eval_ref.recall_flags(e["ad"][i], e["re"][i], e["te"][i], e["proj"][i], 0.102)[m] for i in range(7)]
        assert prec.recalls["ape"][m] == want


def test_bop_csv_lines_and_result_dumps(tmp_path):
    """evaluator.py:341,365-373,429-431 (csv) and :449-455 / :647-660 (pickles + table text)."""
    import pickle
    from geometric_aware_dense_matching_amd import evaluation
    rs = np.random.RandomState(7)
    R = np.linalg.qr(rs.randn(3, 3))[0]
    t = rs.randn(3) * 0.3
    csv = evaluation.BopCsv()
    csv.add("000002/rgb/000431", 5, R, t)
    RT = np.concatenate([R, t[:, None]], 1)[None].repeat(2, 0)
    csv.add_batch(["000048/000007", "000048/000012"], 9, RT)
    assert csv.lines[0] == "scene_id,im_id,obj_id,score,R,t,time" and len(csv.lines) == 4
    want = "{},{},{},{},{},{},{}".format(2, "000431", 5, -1, " ".join(map(str, R.flatten().tolist())),
                                       " ".join(map(str, (t * 1000).flatten().tolist())), -1)           # the reference's format call
    assert csv.lines[1] == want
    f = csv.lines[3].split(",")
    assert f[:4] == ["48", "000012", "9", "-1"] and len(f[4].split(" ")) == 9 and len(f[5].split(" ")) == 3 and f[6] == "-1"
    assert np.allclose(np.array(f[5].split(" "), dtype=np.float64), t * 1000, rtol=0, atol=0)
    path = csv.write(str(tmp_path / "bop" / "gt_ycbv-test.csv"))
    text = open(path).read()
    assert text.split("\n") == csv.lines and not text.endswith("\n")
    tab = evaluation.RecallTable()
    tab.update("ape", dict(ad=np.array([0.001]), re=np.array([1.0]), te=np.array([0.01]), proj=np.array([1.0])), 0.102)
    pe, pr, pt = tab.dump(str(tmp_path), "lmo_test")
    assert [os.path.basename(p) for p in (pe, pr, pt)] == ["_lmo_test_errors.pkl", "_lmo_test_recalls.pkl", "_lmo_test_tab.txt"]
    assert pickle.load(open(pr, "rb"))["ape"]["ad_2"] == [1.0] and pickle.load(open(pe, "rb"))["ape"]["re"] == [1.0]
    assert open(pt).read() == tab.format() + "\n"
    pp = evaluation.RecallTable(precision=True)
    pp.update("ape", dict(ad=np.array([0.001]), re=np.array([1.0]), te=np.array([0.01]), proj=np.array([1.0])), 0.102)
    names = [os.path.basename(p) for p in pp.dump(str(tmp_path), "lmo_test", method_name="geomatch")]
    assert names == ["geomatch_lmo_test_errors.pkl", "geomatch_lmo_test_precisions.pkl", "geomatch_lmo_test_tab_precisions.txt"]
