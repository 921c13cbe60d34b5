"""Generates tests/golden/config_a/: BASELINE.json config 1 with the synthetic substitute SURVEY.md 8d prescribes (the pretrained
shiny_kitchen assets are not available offline) - a model directory in the reference's on-disk layout plus the CPU oracle's image:

    point_cloud.ply        binary little-endian, 21 float properties in the order of scene/gaussian_model.py:366-407 (raw values)
    transforms_test.json   one NeRF / Blender style frame (dataset/blender_dataset.py:30-33,57-68)
    cfg.json               the config values pushed into the native raytracer (renderer/gaussian_raytracer.py:16-25)
    golden_256.npz         oracle render at 256x256 of exactly those files, jitter off, 2 bounces (float16 images, int16 hit counts)

Data only (inputs + expected outputs; no reference source text). The oracle itself is pinned by tests/test_oracle_*.py.

    python tests/golden/make_config_a.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
fmt = importlib.import_module("editable-gaussian-reflections_amd.formats")
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_a")
W = H = 256
N = 3000


def camera_from_frame(fr):
    """What the reference's Camera holds for a transforms frame (scene/cameras.py:22, scene/dataset_readers.py): R = c2w rotation in
    the COLMAP convention, camera_center = c2w translation; the caller-side flip (renderer/gaussian_raytracer.py:95-97) turns R back
    into the Blender-convention c2w the native camera wants."""
    R = np.asarray(fr["R"], np.float32)
    Rb = -R
    Rb[:, 0] = -Rb[:, 0]
    return dict(origin=np.asarray(fr["c2w"][:3, 3], np.float32), c2w=Rb.astype(np.float32), fov=np.float32(fr["FovY"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g = syn.make_scene(N, "trained", seed=21)
    fmt.save_gaussians_ply(os.path.join(OUT, "point_cloud.ply"), g)
    cam = syn.default_camera()
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = cam["c2w"], cam["origin"]
    fovx = fmt.focal2fov(fmt.fov2focal(float(cam["fov"]), H), W)
    with open(os.path.join(OUT, "transforms_test.json"), "w") as f:
        json.dump({"camera_angle_x": fovx, "frames": [{"file_path": "./test/r_0", "transform_matrix": c2w.tolist()}]}, f, indent=1)
    cfg = dict(loss_weight_diffuse=5.0, loss_weight_specular=3.0, loss_weight_normal=2.5, loss_weight_depth=2.5, loss_weight_f0=1.0,
               loss_weight_roughness=1.0, transmittance_threshold=0.01, alpha_threshold=0.005, exp_power=3, znear=0.01, zfar=999.9)
    fmt.save_cfg(os.path.join(OUT, "cfg.json"), cfg)
    # ---- the golden image: rendered from the FILES (what a user of the model directory gets)
    g2 = fmt.load_gaussians_ply(os.path.join(OUT, "point_cloud.ply"))
    fr = fmt.read_transforms(os.path.join(OUT, "transforms_test.json"), W, H)[0]
    c = camera_from_frame(fr)
    o = orc.Oracle(W, H)
    o.set_camera(c["origin"], c["c2w"], c["fov"], cfg["znear"], cfg["zfar"])
    o.set_config(jitter_primary_rays=0, num_bounces=2, **{k: v for k, v in cfg.items() if k not in ("znear", "zfar")})
    o.set_gaussians(g2)
    o.update_bvh()
    ref = o.raytrace(False)
    # float16 storage bounds the comparison at ~75 dB (the bar is 50 dB); the integer image is exact
    np.savez_compressed(os.path.join(OUT, "golden_256.npz"), output_final=ref["output_final"].astype(np.float16),
                        output_rgb0=ref["output_rgb"][0].astype(np.float16), output_depth0=ref["output_depth"][0].astype(np.float16),
                        num_accumulated=ref["num_accumulated"].astype(np.int16))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")
