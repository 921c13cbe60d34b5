"""distCUDA2 replacement (csrc/knn.hip) against the brute-force / kd-tree restatement in oracle/knn.py."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn as oknn  # noqa: E402

torch = pytest.importorskip("torch")
PKG = "editable-gaussian-reflections_amd"


def test_restatements_agree():
    rng = np.random.default_rng(0)
    p = rng.normal(size=(700, 3)).astype(np.float32)
    p[10] = p[11]  # a duplicate pair: distance 0 counts
    a, b = oknn.dist2_bruteforce(p), oknn.dist2_kdtree(p)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-14)
    # hand-checked: 4 points on a line at 0, 1, 3, 7 -> for x=0 the neighbours are at 1, 3, 7: (1 + 9 + 49) / 3
    q = np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0], [7, 0, 0]], np.float32)
    np.testing.assert_allclose(oknn.dist2_bruteforce(q), [59 / 3, (1 + 4 + 36) / 3, (4 + 9 + 16) / 3, (16 + 36 + 49) / 3])


@pytest.fixture(scope="module")
def knn():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    return importlib.import_module(PKG + ".simple_knn")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 5, 63, 1025, 5000])
def test_matches_brute_force(knn, n):
    rng = np.random.default_rng(n)
    p = (rng.normal(size=(n, 3)) * np.array([1.0, 3.0, 0.2])).astype(np.float32)
    if n > 100:
        p[7] = p[50]  # duplicates
        p[n // 2:n // 2 + 40, 2] = 0.0  # a coplanar cluster
    out = knn.distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy()
    ref = oknn.dist2_bruteforce(p)
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-9)


@pytest.mark.gpu
def test_tiny_and_degenerate_inputs(knn):
    assert knn.distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
    assert float(knn.distCUDA2(torch.zeros(1, 3).cuda())[0]) == 0.0
    two = knn.distCUDA2(torch.tensor([[0.0, 0, 0], [2.0, 0, 0]]).cuda()).cpu().numpy()
    np.testing.assert_allclose(two, [4.0, 4.0])
    same = knn.distCUDA2(torch.ones(100, 3).cuda()).cpu().numpy()  # all coincident
    assert np.all(same == 0.0)
    with pytest.raises(RuntimeError):
        knn.distCUDA2(torch.zeros(8, 3))  # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        knn.distCUDA2(torch.zeros(8, 2).cuda())


@pytest.mark.gpu
def test_scene_scale_init_1m_points(knn):
    """The reference's use (gaussian_model.py:197-201) at full size: 1M surface points, checked against the kd-tree."""
    syn = importlib.import_module(PKG + ".synthetic")
    g = syn.make_scene(1_000_000, "init", seed=3)
    p = g["mean"].astype(np.float32)
    t = torch.from_numpy(p).cuda()
    out = knn.distCUDA2(t)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    out = knn.distCUDA2(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sub = np.random.default_rng(0).choice(len(p), 20000, replace=False)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(p.astype(np.float64)).query(p[sub].astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    np.testing.assert_allclose(out.cpu().numpy()[sub], ref, rtol=5e-5, atol=1e-10)
    scales = torch.log(torch.sqrt(torch.clamp_min(out, 1e-7)))  # what the caller derives
    assert bool(torch.isfinite(scales).all())
    print(f"distCUDA2 1M points: {dt * 1e3:.1f} ms")
    assert dt < 2.0
