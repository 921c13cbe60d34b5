"""Generates tests/golden/reference_primary_rays.npz by IMPORTING the reference's pure-torch helper
editable_gauss_refl/utils/depth_utils.py:28-63 (compute_primary_ray_directions). Run in the build container
only (needs /root/reference); the .npz it writes is data (inputs + expected outputs), committed next to it.

    python tests/golden/make_reference_vectors.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from editable_gauss_refl.utils.depth_utils import compute_primary_ray_directions  # noqa: E402


def rot(axis, ang):
    axis = np.asarray(axis, np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


cases = [(4, 6, 0.8, np.eye(3)), (9, 16, 0.6911, rot((0.3, -0.5, 0.8), 1.1)), (32, 24, 1.2, rot((1, 2, 3), -2.0)),
         (27, 48, 0.6911, np.array([[0.0, 0.0, -1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]))]
out = {}
for i, (h, w, fov, c2w) in enumerate(cases):
    d = compute_primary_ray_directions(h, w, fov, torch.tensor(c2w, dtype=torch.float64))
    out[f"case{i}_hw"] = np.array([h, w])
    out[f"case{i}_fov"] = np.array(fov)
    out[f"case{i}_c2w"] = c2w
    out[f"case{i}_dirs"] = d.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_primary_rays.npz"), **out)
print("wrote", len(cases), "cases; probe dir[0,0] of case0 =", out["case0_dirs"][0, 0])
