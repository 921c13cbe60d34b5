"""ctypes view of include/egr_raytracer.h for hosts WITHOUT torch: the structs mirror the header field for
field, every argument is a raw device pointer or an integer. `RawRaytracer` runs the sequence of the reference's `Raytracer`
constructor and methods (cuda/csrc/raytracer.cpp:45-120) on buffers the caller owns - any allocator that yields HIP device
pointers will do (tests/test_c_abi_direct.py uses torch tensors purely as that allocator and checks the result bit for bit against
the TORCH_LIBRARY shim)."""
import ctypes as C
import os

from . import HIP_LIB_PATH

_F, _U8, _I32 = C.c_void_p, C.c_void_p, C.c_void_p  # all fields are device pointers; the aliases only document the element type


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


GAUSSIAN_PARAMS = ("rgb", "normal", "f0", "roughness", "opacity", "scale", "mean", "rotation")
GAUSSIAN_GRADS = ("dL_drgb", "dL_dnormal", "dL_df0", "dL_droughness", "dL_dopacity", "dL_dscale", "dL_dmean", "dL_drotation", "total_weight")
CONFIG_FIELDS = ("exp_power", "alpha_threshold", "transmittance_threshold", "accumulate_samples", "jitter_primary_rays", "num_bounces", "global_scale_factor",
                 "loss_weight_diffuse", "loss_weight_specular", "loss_weight_depth", "loss_weight_normal", "loss_weight_f0", "loss_weight_roughness",
                 "eps_forward_normalization", "eps_scale_grad", "eps_ray_surface_offset", "eps_min_roughness", "reflection_invalid_normal_threshold",
                 "backfacing_invalid_normal_threshold", "backfacing_max_dist")
CAMERA_FIELDS = ("origin", "vertical_fov_radians", "rotation_c2w", "rotation_w2c", "znear", "zfar")
FRAMEBUFFER_FIELDS = ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance", "output_total_transmittance",
                      "output_ray_origin", "output_ray_direction", "output_final", "output_denoised", "accumulated_rgb", "accumulated_transmittance",
                      "accumulated_total_transmittance", "accumulated_depth", "accumulated_normal", "accumulated_f0", "accumulated_roughness",
                      "accumulated_sample_count", "target_diffuse", "target_specular", "target_depth", "target_normal", "target_f0", "target_roughness")
METADATA_FIELDS = ("grads_enabled", "total_num_calls", "random_seeds")
STATS_FIELDS = ("num_accumulated_per_pixel", "num_traversed_per_pixel")

egr_gaussians = _struct("egr_gaussians", [("count", C.c_uint32)] + [(k, _F) for k in GAUSSIAN_PARAMS + GAUSSIAN_GRADS])
egr_config = _struct("egr_config", [(k, _F) for k in CONFIG_FIELDS])
egr_camera = _struct("egr_camera", [(k, _F) for k in CAMERA_FIELDS])
egr_framebuffer = _struct("egr_framebuffer", [(k, _F) for k in FRAMEBUFFER_FIELDS])
egr_metadata = _struct("egr_metadata", [(k, _F) for k in METADATA_FIELDS])
egr_stats = _struct("egr_stats", [(k, _F) for k in STATS_FIELDS])
egr_counters = _struct("egr_counters", [("rays", C.c_uint64 * 3), ("candidates", C.c_uint64 * 3), ("composited", C.c_uint64 * 3), ("lifetime_rays", C.c_uint64),
                                        ("lifetime_launches", C.c_uint32), ("status", C.c_uint32), ("bvh_depth", C.c_uint32), ("bucket_records", C.c_uint32),
                                        ("device_bytes", C.c_uint64), ("arena_blocks_used", C.c_uint32), ("arena_blocks_cap", C.c_uint32),
                                        ("ext_blocks_used", C.c_uint32), ("ext_blocks_cap", C.c_uint32), ("accepted", C.c_uint64 * 3)])

_lib = None


def lib(path=None):
    """dlopen libegr_hip.so and declare the prototypes of include/egr_raytracer.h."""
    global _lib
    if _lib is None:
        path = path or HIP_LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it first (there is no CPU fallback)")
        L = C.CDLL(path)
        P = C.c_void_p
        L.egr_create.argtypes = [C.POINTER(P), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64]
        L.egr_destroy.argtypes = [P]
        L.egr_destroy.restype = None
        L.egr_bind.argtypes = [P, C.POINTER(egr_camera), C.POINTER(egr_config), C.POINTER(egr_framebuffer), C.POINTER(egr_metadata), C.POINTER(egr_stats)]
        L.egr_set_gaussians.argtypes = [P, C.POINTER(egr_gaussians)]
        for name in ("egr_rebuild_bvh", "egr_update_bvh", "egr_denoise", "egr_reset_lifetime_counters", "egr_debug_check_bvh"):
            getattr(L, name).argtypes = [P, P]
        L.egr_raytrace.argtypes = [P, C.c_int, P]
        L.egr_update_bvh_ex.argtypes = [P, C.c_uint, P]
        L.egr_set_partition.argtypes = [P, C.c_int, C.c_int]
        L.egr_set_exact_stats.argtypes = [P, C.c_int]
        L.egr_set_grad_overwrite.argtypes = [P, C.c_int]
        L.egr_grad_delta_consumed.argtypes = [P]
        L.egr_debug_set_pixel_mask.argtypes = [P, P]
        L.egr_set_targets_chw.argtypes = [P, P, P, P, P, P, P, P]
        L.egr_set_camera_from_dataset.argtypes = [P, P, P, C.c_float, C.c_float, C.c_float, P]
        L.egr_tile_owner.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.egr_tile_owner.restype = C.c_int
        L.egr_set_strands.argtypes = [P, C.c_int]
        L.egr_set_rays_per_task.argtypes = [P, C.c_int]
        L.egr_set_team_help.argtypes = [P, C.c_int]
        L.egr_get_counters.argtypes = [P, C.POINTER(egr_counters), P]
        L.egr_get_counters_ex.argtypes = [P, P, C.c_size_t, P]
        L.egr_last_error.argtypes = [P]
        L.egr_last_error.restype = C.c_char_p
        L.egr_version.restype = C.c_char_p
        _lib = L
    return _lib


class RawRaytracer:
    """The reference's `Raytracer` life cycle on caller-owned device buffers.

    `pointers`: dict name -> integer device address for every field of the six structs above (see the *_FIELDS tuples);
    `count`: number of gaussians; `stream`: a hipStream_t as an integer (0 = the NULL stream)."""

    def __init__(self, width, height, count, pointers, ppll_forward_size=180_000_000, ppll_backward_size=120_000_000, device=0, stream=0):
        self.L = lib()
        self.stream = C.c_void_p(stream)
        self.ctx = C.c_void_p()
        if self.L.egr_create(C.byref(self.ctx), device, width, height, ppll_forward_size, ppll_backward_size) != 0:  # raytracer.cpp:45-60
            raise RuntimeError("egr_create failed: no usable HIP device (there is no CPU fallback)")
        fill = lambda st, names: st(**{k: pointers[k] for k in names})
        self.cam, self.cfg = fill(egr_camera, CAMERA_FIELDS), fill(egr_config, CONFIG_FIELDS)
        self.fb, self.meta, self.stats = fill(egr_framebuffer, FRAMEBUFFER_FIELDS), fill(egr_metadata, METADATA_FIELDS), fill(egr_stats, STATS_FIELDS)
        self._check(self.L.egr_bind(self.ctx, C.byref(self.cam), C.byref(self.cfg), C.byref(self.fb), C.byref(self.meta), C.byref(self.stats)))  # :61-68
        self.set_gaussians(count, pointers)
        self.rebuild_bvh()  # :76-78

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.egr_last_error(self.ctx).decode())  # C++ exceptions upstream -> RuntimeError in Python

    def set_gaussians(self, count, pointers):  # Raytracer::resize (:112-120): the holder re-reifies, the struct is uploaded again
        self.g = egr_gaussians(count=count, **{k: pointers[k] for k in GAUSSIAN_PARAMS + GAUSSIAN_GRADS})
        self._check(self.L.egr_set_gaussians(self.ctx, C.byref(self.g)))

    def rebuild_bvh(self):  # :102-110
        self._check(self.L.egr_rebuild_bvh(self.ctx, self.stream))

    def update_bvh(self, fuse_live=False):  # :100 (fuse_live: egr_update_bvh_ex with EGR_UPDATE_FUSE_LIVE)
        if fuse_live:
            self._check(self.L.egr_update_bvh_ex(self.ctx, 1, self.stream))
        else:
            self._check(self.L.egr_update_bvh(self.ctx, self.stream))

    def raytrace(self, grads_enabled):  # :81-94
        self._check(self.L.egr_raytrace(self.ctx, 1 if grads_enabled else 0, self.stream))

    def denoise(self):  # :96
        self._check(self.L.egr_denoise(self.ctx, self.stream))

    def counters(self):  # synchronises the stream; the sized call: the library never writes more than THIS mirror of the struct holds
        c = egr_counters()
        self._check(self.L.egr_get_counters_ex(self.ctx, C.byref(c), C.sizeof(c), self.stream))
        return c

    def close(self):
        if self.ctx:
            self.L.egr_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
