"""Synthetic "dense-init" Gaussian cloud + camera + targets.

Mirrors what the reference's dense initialisation produces (prepare_initial_ply.py:82-104 voxel-averaged
surface points; scene/gaussian_model.py:197-230 isotropic log-scales from point spacing, init_opa,
init_roughness, init_f0 from config.py:44-49): a closed room whose six walls carry 85 % of the Gaussians on a
jittered grid, plus three reflective spheres carrying 15 %. All arrays are the reference's *raw*
(pre-activation) parameters with the reference's shapes (cuda/csrc/core/gaussians.h:6-13).

Pure numpy; shared by bench.py, __graft_entry__.smoke() and the tests so that the HIP path and the CPU
oracle are always fed bit-identical inputs.
"""
import math

import numpy as np

ROOM_HALF = np.array([2.0, 2.0, 1.5])
SPHERES = [((1.0, 0.8, -1.0), 0.5), ((1.0, -0.8, -1.0), 0.5), ((-1.0, 0.0, -1.0), 0.5)]

# reference training loss weights (config.py:53-58)
TRAIN_LOSS_WEIGHTS = dict(loss_weight_diffuse=5.0, loss_weight_specular=3.0, loss_weight_depth=2.5,
                          loss_weight_normal=2.5, loss_weight_f0=1.0, loss_weight_roughness=1.0)


def _logit(p):
    return math.log(p / (1.0 - p))


def make_scene(n, variant="trained", seed=0, init_scale=1.0, dtype=np.float32):
    """Returns dict(rgb, normal, f0, roughness, opacity, scale, mean, rotation) of raw parameters.

    variant: "init"    -> opacity sigmoid^-1(0.1) (config.py:44 init_opa)
             "trained" -> opacity sigmoid^-1(0.8) (surfaces opaque enough for reflection bounces)
    """
    rng = np.random.default_rng(seed)
    n_sph_total = int(round(0.15 * n))
    n_wall = n - n_sph_total
    hx, hy, hz = ROOM_HALF
    faces = [  # (axis, sign, extents of the two in-plane axes)
        (0, +1, (hy, hz)), (0, -1, (hy, hz)), (1, +1, (hx, hz)), (1, -1, (hx, hz)), (2, +1, (hx, hy)), (2, -1, (hx, hy))]
    areas = np.array([4.0 * a * b for _, _, (a, b) in faces])
    spacing = math.sqrt(areas.sum() / max(n_wall, 1))
    means, normals, is_sphere = [], [], []
    remaining = n_wall
    for fi, (axis, sign, (ea, eb)) in enumerate(faces):
        cnt = remaining if fi == len(faces) - 1 else int(round(n_wall * areas[fi] / areas.sum()))
        cnt = max(min(cnt, remaining), 0)
        remaining -= cnt
        if cnt == 0:
            continue
        na = max(int(round(math.sqrt(cnt * ea / eb))), 1)
        nb = (cnt + na - 1) // na
        k = np.arange(cnt)
        ia, ib = k % na, k // na
        u = (ia + 0.5 + rng.uniform(-0.25, 0.25, cnt)) / na * 2 * ea - ea
        v = (ib + 0.5 + rng.uniform(-0.25, 0.25, cnt)) / nb * 2 * eb - eb
        p = np.zeros((cnt, 3))
        others = [a for a in range(3) if a != axis]
        p[:, axis] = sign * ROOM_HALF[axis]
        p[:, others[0]] = u
        p[:, others[1]] = v
        nrm = np.zeros((cnt, 3))
        nrm[:, axis] = -sign  # inward
        means.append(p), normals.append(nrm), is_sphere.append(np.zeros(cnt, bool))
    for si, (c, r) in enumerate(SPHERES):
        cnt = n_sph_total // len(SPHERES) + (1 if si < n_sph_total % len(SPHERES) else 0)
        if cnt == 0:
            continue
        k = np.arange(cnt) + 0.5  # Fibonacci lattice: near-uniform surface samples
        z = 1.0 - 2.0 * k / cnt
        phi = k * math.pi * (3.0 - math.sqrt(5.0))
        rad = np.sqrt(np.maximum(1.0 - z * z, 0.0))
        d = np.stack([rad * np.cos(phi), rad * np.sin(phi), z], 1)
        means.append(np.asarray(c)[None, :] + r * d), normals.append(d), is_sphere.append(np.ones(cnt, bool))
    mean = np.concatenate(means)[:n]
    normal = np.concatenate(normals)[:n]
    sph = np.concatenate(is_sphere)[:n]
    m = mean.shape[0]
    scale = np.log(spacing * init_scale * rng.uniform(0.8, 1.2, (m, 3)))
    q = rng.normal(size=(m, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opa = np.full((m, 1), _logit(0.1 if variant == "init" else 0.8))
    checker = (np.floor(mean[:, 0] * 2.0) + np.floor(mean[:, 1] * 2.0) + np.floor(mean[:, 2] * 2.0)).astype(np.int64) & 1
    rgb = np.where(checker[:, None] == 1, 0.8, 0.2) * np.array([[1.0, 0.9, 0.8]])
    roughness = np.where(sph, 0.02, 0.1)[:, None]
    f0 = np.where(sph[:, None], 0.9, 0.04) * np.ones((m, 3))
    # estimated normals are never axis-exact in practice; exact (0,0,-1) also trips the reference's
    # sample_cook_torrance tangent-frame NaN (ggx_brdf.h:163 tests N.z, not |N.z|)
    normal = normal + rng.normal(scale=0.02, size=(m, 3))
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    perm = rng.permutation(m)  # the reference's clouds are not spatially sorted
    g = dict(rgb=rgb, normal=normal, f0=f0, roughness=roughness, opacity=opa, scale=scale, mean=mean, rotation=q)
    return {k: np.ascontiguousarray(v[perm].astype(dtype)) for k, v in g.items()}


def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """c2w rotation in the convention of cuda/csrc/core/camera.h:17-36: dir = c2w @ (x, y, -1)."""
    eye, target, up = map(lambda a: np.asarray(a, np.float64), (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    return np.stack([right, upv, -fwd], axis=1)


def default_camera():
    """Camera inside the room, seeing two spheres, the floor and two walls. FoVy from SURVEY §8d."""
    eye = np.array([-1.7, -1.2, 0.4])
    return dict(origin=eye.astype(np.float32), c2w=look_at(eye, (1.2, 0.5, -0.9)).astype(np.float32),
                fov=np.float32(0.6911), znear=np.float32(0.01), zfar=np.float32(999.9))


def make_targets(width, height, dtype=np.float32):
    """Procedural targets in the reference's framebuffer layout (core/framebuffer.h:181-186, HWC)."""
    yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    chk = (((xx // 32) + (yy // 32)) & 1).astype(dtype)
    diffuse = np.stack([0.35 + 0.3 * chk, 0.4 + 0.2 * chk, 0.45 + 0.1 * chk], -1).astype(dtype)
    specular = np.full((height, width, 3), 0.1, dtype)
    depth = np.full((height, width, 1), 2.0, dtype)
    normal = np.zeros((height, width, 3), dtype)
    normal[..., 0] = -1.0
    f0 = np.full((height, width, 3), 0.04, dtype)
    roughness = np.full((height, width, 1), 0.1, dtype)
    return dict(diffuse=diffuse, specular=specular, depth=depth, normal=normal, f0=f0, roughness=roughness)


def random_blob_scene(n, seed=0, extent=1.0, depth_range=(1.5, 4.0), scale_range=(0.05, 0.25), dtype=np.float32):
    """Small anisotropic random cloud in front of a +x looking camera: used by parity / finite-difference tests."""
    rng = np.random.default_rng(seed)
    mean = np.stack([rng.uniform(*depth_range, n), rng.uniform(-extent, extent, n), rng.uniform(-extent, extent, n)], 1)
    scale = np.log(rng.uniform(scale_range[0], scale_range[1], (n, 3)))
    q = rng.normal(size=(n, 4))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    g = dict(rgb=rng.uniform(0.05, 0.95, (n, 3)), normal=nrm, f0=rng.uniform(0.05, 0.95, (n, 3)),
             roughness=rng.uniform(0.05, 0.95, (n, 1)), opacity=rng.uniform(-1.5, 2.0, (n, 1)), scale=scale, mean=mean, rotation=q)
    return {k: np.ascontiguousarray(v.astype(dtype)) for k, v in g.items()}


def plus_x_camera(fov=0.6911):
    """Origin at 0 looking along +x with z up."""
    return dict(origin=np.zeros(3, np.float32), c2w=look_at((0, 0, 0), (1, 0, 0)).astype(np.float32), fov=np.float32(fov),
                znear=np.float32(0.01), zfar=np.float32(999.9))
