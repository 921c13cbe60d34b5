"""`simple_knn._C.distCUDA2` for MI355X.

The reference does `from simple_knn._C import distCUDA2` (editable_gauss_refl/scene/gaussian_model.py:17) and calls it on an
[N,3] float GPU tensor to get, per point, the mean squared distance to its 3 nearest neighbours (:197-201, :246-250). This
module exposes the same name over `egr_knn_mean_dist2` (csrc/knn.hip) through `torch.ops.simple_knn.distCUDA2`, which
libraytracer.so registers. GPU only: a CPU tensor raises.
"""
import importlib

import torch

importlib.import_module(__package__).load_library()


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    return torch.ops.simple_knn.distCUDA2(points)
