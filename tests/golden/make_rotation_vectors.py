"""Generates tests/golden/reference_scaling_rotation.npz by RUNNING the reference's `build_rotation` / `build_scaling_rotation`
(editable_gauss_refl/utils/general_utils.py:79-113) - the reference's own Python statement of the quaternion -> R . diag(s) convention
((r, x, y, z) order, normalised inside) that `create_transform_matrix` (cuda/csrc/optix/bvh_wrapper.cu:9-31) implements on the device.
Build container only (needs /root/reference); the .npz is data: inputs (activated scales, RAW quaternions) and the reference's outputs.
The two helpers allocate with device="cuda"; there is no GPU here, so torch.zeros is wrapped to drop that argument for this script.

    python tests/golden/make_rotation_vectors.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
from editable_gauss_refl.utils.general_utils import build_rotation, build_scaling_rotation  # noqa: E402

rng = np.random.default_rng(20260929)
n = 96
quat = rng.normal(size=(n, 4)) * rng.uniform(0.2, 5.0, size=(n, 1))  # raw, un-normalised (the activation normalises)
quat[0] = [1, 0, 0, 0]
quat[1] = [0, 1, 0, 0]
quat[2] = [0, 0, 2, 0]
quat[3] = [0, 0, 0, -3]
quat[4] = [1, 1, 0, 0]
scale = np.exp(rng.uniform(-5.0, 0.5, size=(n, 3)))  # activated scales exp(scale_raw), anisotropic
scale[5] = [1, 1, 1]
q32, s32 = torch.tensor(quat, dtype=torch.float32), torch.tensor(scale, dtype=torch.float32)
R = build_rotation(q32).numpy()
L = build_scaling_rotation(s32, q32).numpy()
R64 = build_rotation(torch.tensor(quat, dtype=torch.float64)).numpy()  # (the helper allocates float32 output: values are fp64 inside, stored fp32)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_scaling_rotation.npz")
np.savez_compressed(out, rotation_raw=q32.numpy(), scaling=s32.numpy(), R=R, L=L)
print(out, R.shape, L.shape, float(np.abs(R - R64).max()))
