"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 6): CPU restatement of `simple_knn._C.distCUDA2`.

simple-knn is an un-vendored submodule of the reference (`/root/reference/submodules/simple-knn` is empty), so there is no
source or golden vector to pin against: PARITY UNPINNED. Its published contract - used by
editable_gauss_refl/scene/gaussian_model.py:197-201,246-250 - is "mean of the squared distances to the 3 nearest neighbours";
this brute-force restatement is the checker for csrc/knn.hip. Only tests/ may import it.
"""
import numpy as np


def dist2_bruteforce(points, chunk=1024):
    """O(N^2) exact 3-NN mean squared distance in float64 (self excluded by index, duplicates count as distance 0)."""
    p = np.asarray(points, np.float64)
    n = len(p)
    out = np.zeros(n, np.float64)
    for s in range(0, n, chunk):
        d = ((p[s:s + chunk, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(min(chunk, n - s)), np.arange(s, min(s + chunk, n))] = np.inf
        k = min(3, n - 1)
        if k <= 0:
            continue
        part = np.partition(d, k - 1, axis=1)[:, :k]
        out[s:s + chunk] = part.mean(1)
    return out


def dist2_kdtree(points):
    """Same quantity through scipy's exact kd-tree (for sizes where O(N^2) is too slow)."""
    from scipy.spatial import cKDTree

    p = np.asarray(points, np.float64)
    d, _ = cKDTree(p).query(p, k=min(4, len(p)))
    return (d[:, 1:] ** 2).mean(1)
