"""Generates tests/golden/scene_2k_64.npz: a seeded 2k-Gaussian synthetic scene (inputs) and the fp32 CPU oracle's
outputs for it (all per-step output buffers, gradients, total_weight, both stats) = SURVEY.md 8c item 4.
The oracle itself is pinned by tests/test_oracle_*.py. Data only (no reference source text).

    python tests/golden/make_golden_scene.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
from oracle import oracle as orc  # noqa: E402

W, H, N = 64, 64, 2000
g = syn.make_scene(N, "trained", seed=7)
cam = syn.default_camera()
tg = syn.make_targets(W, H)
out = {"W": W, "H": H}
for k, v in g.items():
    out["g_" + k] = v
for k, v in cam.items():
    out["cam_" + k] = np.asarray(v)
for k, v in tg.items():
    out["tg_" + k] = v
o = orc.Oracle(W, H)
o.set_camera(cam["origin"], cam["c2w"], cam["fov"])
o.set_config(jitter_primary_rays=1, num_bounces=2, **syn.TRAIN_LOSS_WEIGHTS)
o.set_gaussians(g)
o.update_bvh()
ref = o.raytrace(False)  # total_num_calls = 1
for k in ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance",
          "output_total_transmittance", "output_final"):
    out["ref_" + k] = ref[k].astype(np.float32)
out["ref_num_traversed"] = ref["num_traversed"]
out["ref_num_accumulated"] = ref["num_accumulated"]
out["ref_random_seeds"] = ref["random_seeds"]
refg = o.raytrace(True, targets=tg)  # total_num_calls = 2
for k in ("dL_drgb", "dL_dnormal", "dL_df0", "dL_droughness", "dL_dopacity", "dL_dscale", "dL_dmean", "dL_drotation", "total_weight"):
    out["ref_" + k] = refg[k].astype(np.float32)
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_2k_64.npz")
np.savez_compressed(p, **out)
print("wrote", p, os.path.getsize(p) // 1024, "KiB")
