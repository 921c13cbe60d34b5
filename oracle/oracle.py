"""ctypes front-end for the CPU oracle (oracle/egr_oracle.cpp).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The shipped HIP path never imports this module.
Parity status: unpinned by the reference (no golden numbers exist upstream for this path);
pinned by the known-answer / finite-difference / golden-fixture tests in tests/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libegr_oracle.so")
_lib = None

NSTEPS = 3

CONFIG_FIELDS = [  # order of core/config.h:5-26, defaults :32-51
    ("exp_power", 3.0),
    ("alpha_threshold", 0.005),
    ("transmittance_threshold", 0.01),
    ("accumulate_samples", 0),
    ("jitter_primary_rays", 1),
    ("num_bounces", 2),
    ("global_scale_factor", 1.0),
    ("loss_weight_diffuse", 1.0),
    ("loss_weight_specular", 1.0),
    ("loss_weight_depth", 1.0),
    ("loss_weight_normal", 1.0),
    ("loss_weight_f0", 1.0),
    ("loss_weight_roughness", 1.0),
    ("eps_forward_normalization", 1e-12),
    ("eps_scale_grad", 1e-12),
    ("eps_ray_surface_offset", 0.01),
    ("eps_min_roughness", 0.01),
    ("reflection_invalid_normal_threshold", 0.7),
    ("backfacing_invalid_normal_threshold", 0.9),
    ("backfacing_max_dist", 0.1),
]

GAUSSIAN_FIELDS = [("rgb", 3), ("normal", 3), ("f0", 3), ("roughness", 1), ("opacity", 1), ("scale", 3), ("mean", 3), ("rotation", 4)]


def build(force=False):
    src = os.path.join(_HERE, "egr_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libegr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Targets(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("diffuse", "specular", "depth", "normal", "f0", "roughness")]


_OUT_F64 = ["output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance",
            "output_total_transmittance", "output_ray_origin", "output_ray_direction", "output_final"]
_OUT_INT = ["random_seeds", "num_traversed", "num_accumulated", "num_composited_all_steps", "effective_steps", "num_composited_per_step", "num_depth_ties"]
_OUT_GRAD = ["dL_drgb", "dL_dnormal", "dL_df0", "dL_droughness", "dL_dopacity", "dL_dscale", "dL_dmean", "dL_drotation",
             "total_weight"]


_OUT_DIAG = ["num_bounce_near_ties", "hit_sequence_hash", "decision_margin"]  # (after num_depth_ties in the C struct)


class _Outputs(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in _OUT_F64 + _OUT_INT + _OUT_DIAG + _OUT_GRAD]


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_set_config.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_set_camera.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.orc_set_gaussians.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8
        L.orc_set_use_bvh.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_set_partition.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_set_pixel_mask.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_update_bvh.argtypes = [ctypes.c_void_p]
        L.orc_reset_accumulators.argtypes = [ctypes.c_void_p]
        L.orc_raytrace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_get_instances.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
        L.orc_primary_ray.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_tea4.restype = ctypes.c_uint32
        L.orc_tea4.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.orc_lcg.restype = ctypes.c_uint32
        L.orc_lcg.argtypes = [ctypes.c_void_p]
        L.orc_sample_cook_torrance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        L.orc_cook_torrance_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    """Stateful mirror of the reference `Raytracer` semantics (raytracer.cpp:24-120) on the CPU.

    `set_gaussians` = the 8 `copy_` of GaussianRaytracer._export_param_values,
    `update_bvh` = snapshot of the instance transforms, `raytrace` = one launch.
    """

    def __init__(self, width, height, double=False, threads=None, use_bvh=True):
        self.L = lib()
        self.W, self.H = int(width), int(height)
        self.double = bool(double)
        self.h = self.L.orc_create(self.W, self.H, int(self.double))
        self.config = {k: v for k, v in CONFIG_FIELDS}
        self.total_num_calls = 0
        self.n = 0
        self.L.orc_set_use_bvh(self.h, int(use_bvh))
        if threads is not None:
            self.L.orc_set_threads(int(threads))
        self.set_camera(np.zeros(3), np.eye(3), 1.0, 0.01, 999.9)
        self._push_config()

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_config(self, **kw):
        for k, v in kw.items():
            if k not in self.config:
                raise KeyError(k)
            self.config[k] = v
        self._push_config()

    def _push_config(self):
        c = _f64([float(self.config[k]) for k, _ in CONFIG_FIELDS])
        self.L.orc_set_config(self.h, _ptr(c))

    def set_camera(self, origin, c2w, fov, znear=0.01, zfar=999.9):
        o, r = _f64(origin, (3,)), _f64(c2w, (3, 3))
        if not self.double:  # device tensors are fp32 in the reference
            o, r = _f64(o.astype(np.float32)), _f64(r.astype(np.float32))
            fov, znear, zfar = float(np.float32(fov)), float(np.float32(znear)), float(np.float32(zfar))
        self.L.orc_set_camera(self.h, _ptr(o), _ptr(r), float(fov), float(znear), float(zfar))

    def set_partition(self, rank, world, tile=16):
        self.L.orc_set_partition(self.h, int(rank), int(world), int(tile))

    def set_pixel_mask(self, mask):
        """Test hook: only pixels with mask != 0 are traced (None = all); the others write nothing, like another rank's tiles."""
        if mask is None:
            self.L.orc_set_pixel_mask(self.h, None)
        else:
            m = np.ascontiguousarray(np.asarray(mask).reshape(self.H * self.W) != 0, dtype=np.uint8)
            self.L.orc_set_pixel_mask(self.h, m.ctypes.data_as(ctypes.c_void_p))

    def set_gaussians(self, g):
        """g: dict with raw (pre-activation) arrays rgb[N,3] normal[N,3] f0[N,3] roughness[N,1] opacity[N,1]
        scale[N,3] mean[N,3] rotation[N,4] (core/gaussians.h:6-13)."""
        arrs = []
        n = None
        for k, c in GAUSSIAN_FIELDS:
            a = np.asarray(g[k])
            if not self.double:
                a = a.astype(np.float32)
            a = _f64(a).reshape(-1, c)
            n = a.shape[0] if n is None else n
            assert a.shape[0] == n, k
            arrs.append(a)
        self.n = n
        self.L.orc_set_gaussians(self.h, n, *[_ptr(a) for a in arrs])

    def update_bvh(self):
        self.L.orc_update_bvh(self.h)

    def reset_accumulators(self):
        self.L.orc_reset_accumulators(self.h)

    def instances(self):
        M = np.zeros((self.n, 3, 4)); Wm = np.zeros((self.n, 3, 4)); aabb = np.zeros((self.n, 6)); vis = np.zeros(self.n, np.int32)
        self.L.orc_get_instances(self.h, _ptr(M), _ptr(Wm), _ptr(aabb), _ptr(vis))
        return M, Wm, aabb, vis

    def primary_rays(self, jitter=False, total_num_calls=1):
        """[H,W,3] primary directions (camera.h:17-36), seeded like shaders.cu:88 when jitter is on."""
        d = np.zeros((self.H, self.W, 3))
        v = np.zeros(3)
        for iy in range(self.H):
            for ix in range(self.W):
                st = ctypes.c_uint32(tea4(iy * self.W + ix, total_num_calls))
                self.L.orc_primary_ray(self.h, ix, iy, int(jitter), ctypes.byref(st), _ptr(v))
                d[iy, ix] = v
        return d

    def raytrace(self, grads_enabled=False, targets=None, grads_into=None):
        """One launch. Returns dict of numpy arrays with the reference's tensor shapes
        (core/framebuffer.h:161-187). total_num_calls is incremented first (metadata.h:28-31)."""
        self.total_num_calls += 1
        P = self.W * self.H
        H, W = self.H, self.W
        out = {}
        shp = {"output_rgb": (NSTEPS, H, W, 3), "output_depth": (NSTEPS, H, W, 1), "output_normal": (NSTEPS, H, W, 3),
               "output_f0": (NSTEPS, H, W, 3), "output_roughness": (NSTEPS, H, W, 1), "output_transmittance": (NSTEPS, H, W, 1),
               "output_total_transmittance": (NSTEPS, H, W, 1), "output_ray_origin": (NSTEPS, H, W, 3),
               "output_ray_direction": (NSTEPS, H, W, 3), "output_final": (1, H, W, 3)}
        for k in _OUT_F64:
            out[k] = np.zeros(shp[k], np.float64)
        out["random_seeds"] = np.zeros((H, W, 1), np.uint32)
        for k in _OUT_INT[1:5]:
            out[k] = np.zeros((H, W), np.int32)
        out["num_composited_per_step"] = np.zeros((NSTEPS, H, W), np.int32)
        out["num_depth_ties"] = np.zeros((H, W), np.int32)
        out["num_bounce_near_ties"] = np.zeros((H, W), np.int32)
        out["decision_margin"] = np.full((H, W), 1e30, np.float64)
        out["hit_sequence_hash"] = np.zeros((NSTEPS, H, W), np.uint64)  # ordered composited gaussian ids per step, hashed (0: no hit)
        n = self.n
        gshape = {"dL_drgb": (n, 3), "dL_dnormal": (n, 3), "dL_df0": (n, 3), "dL_droughness": (n, 1), "dL_dopacity": (n, 1),
                  "dL_dscale": (n, 3), "dL_dmean": (n, 3), "dL_drotation": (n, 4), "total_weight": (n, 1)}
        if grads_enabled:
            for k in _OUT_GRAD:
                if grads_into is not None and k in grads_into:
                    out[k] = grads_into[k]
                    assert out[k].dtype == np.float64 and out[k].flags.c_contiguous
                else:
                    out[k] = np.zeros(gshape[k], np.float64)
        tg = _Targets()
        keep = []
        if targets:
            for k in ("diffuse", "specular", "depth", "normal", "f0", "roughness"):
                v = targets.get(k)
                if v is not None:
                    a = np.asarray(v)
                    if not self.double:
                        a = a.astype(np.float32)
                    a = _f64(a)
                    keep.append(a)
                    setattr(tg, k, a.ctypes.data)
        o = _Outputs()
        for k in _OUT_F64 + _OUT_INT + _OUT_DIAG + _OUT_GRAD:
            if k in out:
                setattr(o, k, out[k].ctypes.data)
        self.L.orc_raytrace(self.h, int(bool(grads_enabled)), ctypes.c_uint32(self.total_num_calls), ctypes.byref(tg), ctypes.byref(o))
        return out


def tea4(a, b):
    return int(lib().orc_tea4(a & 0xFFFFFFFF, b & 0xFFFFFFFF))


def lcg_sequence(seed, count):
    st = ctypes.c_uint32(seed & 0xFFFFFFFF)
    outv = []
    for _ in range(count):
        outv.append(int(lib().orc_lcg(ctypes.byref(st))))
    return outv, int(st.value)


def sample_cook_torrance(N, V, roughness, u1, u2):
    L = np.zeros(3)
    lib().orc_sample_cook_torrance(_ptr(_f64(N)), _ptr(_f64(V)), float(roughness), float(u1), float(u2), _ptr(L))
    return L


def cook_torrance_weight(N, V, L, roughness, f0):
    w = np.zeros(3)
    lib().orc_cook_torrance_weight(_ptr(_f64(N)), _ptr(_f64(V)), _ptr(_f64(L)), float(roughness), _ptr(_f64(f0)), _ptr(w))
    return w


def l1_loss(out, targets, cfg, num_bounces):
    """The loss whose gradient backward_pass.cu:80-108 hard-codes (sum over pixels, no mean):
    w_d/3*|rgb0-diffuse| + w_depth*|depth0-t| + w_n/3*|normal0-t| + w_f0/3*|f00-t| + w_r*|rough0-t|
    (+ w_s/3*|sum_{j>=1} rgb_j - specular| for bounce steps; throughput/downweight are constants there)."""
    z3 = 0.0
    def t(k, shape):
        v = targets.get(k) if targets else None
        return np.zeros(shape) if v is None else np.asarray(v, np.float64).reshape(shape)
    H, W = out["output_final"].shape[1:3]
    L = cfg["loss_weight_diffuse"] / 3.0 * np.abs(out["output_rgb"][0] - t("diffuse", (H, W, 3))).sum()
    L += cfg["loss_weight_depth"] * np.abs(out["output_depth"][0] - t("depth", (H, W, 1))).sum()
    L += cfg["loss_weight_normal"] / 3.0 * np.abs(out["output_normal"][0] - t("normal", (H, W, 3))).sum()
    L += cfg["loss_weight_f0"] / 3.0 * np.abs(out["output_f0"][0] - t("f0", (H, W, 3))).sum()
    L += cfg["loss_weight_roughness"] * np.abs(out["output_roughness"][0] - t("roughness", (H, W, 1))).sum()
    return float(L)
