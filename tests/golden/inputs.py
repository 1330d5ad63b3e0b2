"""Deterministic input recipes shared by make_golden.py (run against the reference, here) and
the tests (run anywhere).  Only numpy RandomState streams: the golden .npz files then need to
hold outputs only."""
import numpy as np


def matching_inputs(Ne=1024, Me=4096, seed=13):
    rs = np.random.RandomState(seed)
    return dict(
        seg_features=rs.randn(2, Ne).astype(np.float32),
        rgbd_features=(rs.randn(128, Ne) * (0.5 + rs.rand(1, Ne))).astype(np.float32),
        mesh_features=rs.randn(128, Me).astype(np.float32),
        cld=rs.rand(9, Ne).astype(np.float32),
    )


def loss_inputs(M=512, Bl=2, Nl=256, seed=11):
    rs = np.random.RandomState(seed)
    d = dict(
        rgbd_f=rs.randn(Bl, 128, Nl).astype(np.float32),
        mesh_f=rs.randn(1, 128, M).astype(np.float32),
        labels=(rs.rand(Bl, Nl) < 0.6).astype(np.int64),
        match_idx=rs.randint(0, M + 1, size=(Bl, Nl)).astype(np.int64),
        vis=(rs.rand(Bl, M) < 0.5).astype(np.float32),
        seg=rs.randn(Bl, 2, Nl).astype(np.float32),
        sim=(rs.rand(64, 100).astype(np.float32) * 2 - 1),
    )
    msk = rs.rand(64, 100) < 0.1
    msk[:, 0] = True
    d["mask"] = msk
    return d


def dgcnn_loss_inputs(M=384, Bl=3, Nl=256, seed=31):
    """Training matching of the geoMatch_DGCNN variant (geoMatch_DGCNN.py:52-135): three items -- the last with fewer than three rows
    of origin_labels == 1 (skipped by the reference) -- and a pose per item that puts the model 0.6 .. 1.0 m in front of the camera."""
    rs = np.random.RandomState(seed)
    labels = (rs.rand(Bl, Nl) < 0.6).astype(np.int64)
    labels[Bl - 1] = 0
    labels[Bl - 1, :2] = 1
    RT = np.zeros((Bl, 3, 4), np.float32)
    for b in range(Bl):
        q, _ = np.linalg.qr(rs.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        RT[b, :, :3] = q
        RT[b, :, 3] = (0.05 * rs.randn(), 0.05 * rs.randn(), 0.6 + 0.4 * rs.rand())
    return dict(
        rgbd_f=rs.randn(Bl, 128, Nl).astype(np.float32),
        mesh_f=rs.randn(1, 128, M).astype(np.float32),
        origin_labels=labels,
        match_idx=rs.randint(0, M + 1, size=(Bl, Nl)).astype(np.int64),
        vis=(rs.rand(Bl, M) < 0.5).astype(np.float32),
        RT=RT,
    )


def sym_loss_inputs(M=512, Bl=2, seed=23):
    """Symmetric-object training matching (geoMatch.py:86-100): the reference indexes the per-vertex symmetry table with POINT
    indices and the per-point match table with its values, so the fixture keeps N == M (as the reference's default 4096/4096)."""
    rs = np.random.RandomState(seed)
    Nl = M
    perm = rs.permutation(M)
    sys_idx = np.arange(M)
    for a, b in zip(perm[0::2], perm[1::2]):          # an involution, like a 180-degree symmetry
        sys_idx[a], sys_idx[b] = b, a
    labels = (rs.rand(Bl, Nl) < 0.55).astype(np.int64)
    return dict(
        rgbd_f=rs.randn(Bl, 128, Nl).astype(np.float32),
        mesh_f=rs.randn(1, 128, M).astype(np.float32),
        labels=labels,
        match_idx=rs.randint(0, M + 1, size=(Bl, Nl)).astype(np.int64),
        vis=(rs.rand(Bl, M) < 0.5).astype(np.float32),
        sys_idx=sys_idx.astype(np.int64),
    )


def block_inputs(B=2, n=128, K=16, seed=7):
    rs = np.random.RandomState(seed)
    return dict(
        xyz=rs.rand(B, n, 3).astype(np.float32),
        feat8=rs.randn(B, 8, n, 1).astype(np.float32),
        fset=rs.randn(B, 32, n, K).astype(np.float32),
    )


def pose_inputs(B=3, N=600, M=500, seed=17):
    """Correspondences with a known rigid motion + noise + outliers; one crop with too few points."""
    rs = np.random.RandomState(seed)
    model = (rs.rand(M, 3).astype(np.float32) - 0.5) * 0.2
    idx = rs.randint(0, M, size=(B, N)).astype(np.int32)
    mask = (rs.rand(B, N) < 0.6).astype(np.uint8)
    mask[2] = 0
    mask[2, :3] = 1
    cld = np.zeros((B, 9, N), np.float32)
    RT = np.zeros((B, 3, 4), np.float32)
    for b in range(B):
        q, _ = np.linalg.qr(rs.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        t = np.array([0.05 * b, -0.02, 0.9], np.float32)
        RT[b, :, :3], RT[b, :, 3] = q, t
        pts = model[idx[b]] @ q.T.astype(np.float32) + t + 0.002 * rs.randn(N, 3).astype(np.float32)
        out = rs.rand(N) < 0.1
        pts[out] += 0.05 * rs.randn(int(out.sum()), 3).astype(np.float32)
        cld[b, :3] = pts.T
        cld[b, 3:] = rs.rand(6, N)
    return dict(model=model, idx=idx, mask=mask, cld=cld, RT=RT)


def sym_rotations():
    """A discrete symmetry group as evaluator.py's sym_infos hold it (K x 3 x 3, model to model): identity and the three half turns."""
    out = [np.eye(3)]
    for ax in range(3):
        d = -np.ones(3)
        d[ax] = 1.0
        out.append(np.diag(d))
    return np.stack(out)
