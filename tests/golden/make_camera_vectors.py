"""Generates tests/golden/reference_cameras.npz by RUNNING the reference's own camera math (build container only: needs
/root/reference; the .npz is data - inputs and expected outputs - committed next to this script):

  * editable_gauss_refl/scene/cameras.py `Camera` (loaded by file path: the package's scene/__init__ needs kornia), whose
    `update()` derives `camera_center` from `getWorld2View2(R, T)` (utils/graphics_utils.py:46-57) - the two attributes
    renderer/gaussian_raytracer.py:94-104 reads;
  * utils/graphics_utils.py `focal2fov` / `fov2focal` for FoVy (dataset/blender_dataset.py:58);
  * utils/depth_utils.py `compute_primary_ray_directions` on R_blender = -R with column 0 re-negated - the call
    prepare_initial_ply.py:58-66 makes, i.e. the reference's own statement of which rays a camera (R, T, FoVy) shoots.

Per case: a NeRF/Blender `transform_matrix` + `camera_angle_x` (inputs), the (R, T) dataset/blender_dataset.py:62-69 derives from it
(four numpy lines restated below; that module imports cv2 / tifffile / torchvision, absent here), then the reference outputs.
There is no GPU in the build container: `Tensor.cuda()` is patched to the identity for the duration of this script.

    python tests/golden/make_camera_vectors.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self  # no GPU here; Camera.update() only moves small matrices
from editable_gauss_refl.utils.depth_utils import compute_primary_ray_directions  # noqa: E402
from editable_gauss_refl.utils.graphics_utils import focal2fov, fov2focal, getWorld2View2  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_cameras", "/root/reference/editable_gauss_refl/scene/cameras.py")
ref_cameras = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_cameras)


def random_pose(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = rng.uniform(-1.0, 1.0, 3) * np.array([1.2, 1.2, 0.8])  # inside the synthetic room of synthetic.py
    return m


rng = np.random.default_rng(2024)
sizes = [(64, 40), (40, 64), (48, 48), (72, 40), (33, 17), (96, 54)]
out = {"num_cases": np.array(len(sizes))}
for i, (W, H) in enumerate(sizes):
    tm = random_pose(rng) if i else np.array([[0.0, 0.0, -1.0, 0.0], [-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    fovx = float(rng.uniform(0.5, 1.5))
    fovy = focal2fov(fov2focal(fovx, W), H)  # blender_dataset.py:58
    c2w = tm.copy()
    c2w[:3, 1:3] *= -1  # blender_dataset.py:62-69: OpenGL/Blender (y up, z back) -> COLMAP (y down, z forward)
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    img = torch.zeros(3, H, W)
    cam = ref_cameras.Camera(colmap_id=i, R=R, T=T, FoVx=fovx, FoVy=fovy, image=img, gt_alpha_mask=None, image_name=str(i), uid=i,
                             specular_image=img, diffuse_image=img, depth_image=torch.zeros(1, H, W), normal_image=img,
                             roughness_image=torch.zeros(1, H, W), f0_image=img)
    assert cam.image_width == W and cam.image_height == H
    R_blender = -torch.from_numpy(np.asarray(cam.R)).clone()  # prepare_initial_ply.py:58-59 == gaussian_raytracer.py:95-97
    R_blender[:, 0] = -R_blender[:, 0]
    dirs = compute_primary_ray_directions(H, W, cam.FoVy, R_blender[:3, :3])
    origin_alt = -R @ T  # prepare_initial_ply.py:66
    cc = cam.camera_center.numpy()
    assert np.abs(cc - origin_alt).max() < 1e-5 and np.abs(cc - tm[:3, 3]).max() < 1e-5
    w2v = getWorld2View2(R, T)
    out[f"c{i}_wh"] = np.array([W, H])
    out[f"c{i}_transform_matrix"] = tm
    out[f"c{i}_camera_angle_x"] = np.array(fovx)
    out[f"c{i}_R"] = R
    out[f"c{i}_T"] = T
    out[f"c{i}_FoVy"] = np.array(float(cam.FoVy))
    out[f"c{i}_camera_center"] = cc.astype(np.float32)
    out[f"c{i}_world_view"] = w2v
    out[f"c{i}_R_blender"] = R_blender.numpy()
    out[f"c{i}_dirs"] = dirs.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_cameras.npz"), **out)
print("wrote", len(sizes), "cameras; case 0 centre", out["c0_camera_center"], "dir[0,0]", out["c0_dirs"][0, 0])
