"""Generates tests/golden/psnr_vectors.npz by IMPORTING the reference's pure-torch metric
editable_gauss_refl/utils/image_utils.py:19-21 (psnr: clamp to [0,1], per-image mean, 20*log10(1/sqrt(mse))). Run in the build
container only (needs /root/reference); the .npz it writes is data (inputs + expected outputs), committed next to it.

    python tests/golden/make_psnr_vectors.py
"""
import importlib.util
import os

import numpy as np
import torch

spec = importlib.util.spec_from_file_location("ref_image_utils", "/root/reference/editable_gauss_refl/utils/image_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(7)
out = {}
cases = [
    (rng.random((2, 3, 8, 9)), 1e-3),   # inside [0,1], small error
    (rng.random((1, 3, 16, 16)) * 1.6 - 0.2, 5e-2),  # values outside [0,1]: the reference clamps first
    (rng.random((3, 3, 5, 7)), 1e-6),
    (rng.random((1, 1, 32, 32)) * 4.0, 1e-2),  # HDR-like
]
for i, (a, eps) in enumerate(cases):
    a = a.astype(np.float32)
    b = (a + eps * rng.standard_normal(a.shape)).astype(np.float32)
    p = ref.psnr(torch.tensor(a), torch.tensor(b)).numpy()
    out[f"case{i}_a"], out[f"case{i}_b"], out[f"case{i}_psnr"] = a, b, p
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "psnr_vectors.npz"), **out)
print("wrote", len(cases), "cases;", {k: v.ravel().tolist() for k, v in out.items() if k.endswith("psnr")})
