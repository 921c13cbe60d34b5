"""The C ABI of include/egr_raytracer.h driven WITHOUT the TORCH_LIBRARY shim: ctypes.CDLL(libegr_hip.so), the egr_* structs filled
from raw device addresses (torch is used as nothing but the allocator), the reference's call sequence
(cuda/csrc/raytracer.cpp:45-120: construct = create + bind + set_gaussians + rebuild, update_bvh, raytrace) - and the results
compared bit for bit with the shim path on the same inputs. This is INTEGRATION.md 2's ctypes binding, executable."""
import importlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
PKG = "editable-gaussian-reflections_amd"


def test_ctypes_structs_match_the_header_layout():
    """CPU: field counts / sizes of the ctypes mirrors against the header text (a reordered or added field must fail here)."""
    import os
    import re

    cabi = importlib.import_module(PKG + ".c_abi")
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "egr_raytracer.h")).read()

    def header_fields(struct):
        body = re.search(r"typedef struct " + struct + r" \{(.*?)\} " + struct + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
        return names

    for st in (cabi.egr_gaussians, cabi.egr_config, cabi.egr_camera, cabi.egr_framebuffer, cabi.egr_metadata, cabi.egr_stats, cabi.egr_counters):
        assert [f[0] for f in st._fields_] == header_fields(st.__name__), st.__name__
    L = cabi.lib()
    assert b"gfx950" in L.egr_version()


@pytest.mark.gpu
def test_c_abi_without_the_torch_shim_matches_the_shim_bit_for_bit():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    cabi = importlib.import_module(PKG + ".c_abi")
    ren = importlib.import_module(PKG + ".renderer")
    syn = importlib.import_module(PKG + ".synthetic")
    W, H, N = 96, 64, 4000
    g = syn.make_scene(N, "trained", seed=9)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    dev = "cuda"
    f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    # ---- the caller's buffers (shapes of core/*.h), plain device memory
    T = {}
    for k, c in (("rgb", 3), ("normal", 3), ("f0", 3), ("roughness", 1), ("opacity", 1), ("scale", 3), ("mean", 3), ("rotation", 4)):
        T[k] = torch.tensor(g[k], dtype=torch.float32, device=dev).reshape(N, c).contiguous()
    for k, c in (("dL_drgb", 3), ("dL_dnormal", 3), ("dL_df0", 3), ("dL_droughness", 1), ("dL_dopacity", 1), ("dL_dscale", 3), ("dL_dmean", 3), ("dL_drotation", 4), ("total_weight", 1)):
        T[k] = f32(N, c)
    defaults = dict(exp_power=3.0, alpha_threshold=0.005, transmittance_threshold=0.01, global_scale_factor=1.0, loss_weight_diffuse=5.0, loss_weight_specular=3.0,
                    loss_weight_depth=2.5, loss_weight_normal=2.5, loss_weight_f0=1.0, loss_weight_roughness=1.0, eps_forward_normalization=1e-12, eps_scale_grad=1e-12,
                    eps_ray_surface_offset=0.01, eps_min_roughness=0.01, reflection_invalid_normal_threshold=0.7, backfacing_invalid_normal_threshold=0.9,
                    backfacing_max_dist=0.1)
    for k, v in defaults.items():
        T[k] = torch.tensor([v], dtype=torch.float32, device=dev)
    T["accumulate_samples"] = torch.zeros(1, dtype=torch.bool, device=dev)
    T["jitter_primary_rays"] = torch.ones(1, dtype=torch.bool, device=dev)
    T["num_bounces"] = torch.full((1,), 2, dtype=torch.int32, device=dev)
    c2w = torch.tensor(cam["c2w"], dtype=torch.float32, device=dev)
    T["origin"], T["rotation_c2w"], T["rotation_w2c"] = torch.tensor(cam["origin"], device=dev), c2w.contiguous(), c2w.t().contiguous()  # camera.h:62-68
    T["vertical_fov_radians"], T["znear"], T["zfar"] = torch.tensor([float(cam["fov"])], device=dev), torch.tensor([0.01], device=dev), torch.tensor([999.9], device=dev)
    for k, c in (("output_rgb", 3), ("output_depth", 1), ("output_normal", 3), ("output_f0", 3), ("output_roughness", 1), ("output_transmittance", 1),
                 ("output_total_transmittance", 1), ("output_ray_origin", 3), ("output_ray_direction", 3), ("accumulated_rgb", 3), ("accumulated_transmittance", 1),
                 ("accumulated_total_transmittance", 1), ("accumulated_depth", 1), ("accumulated_normal", 3), ("accumulated_f0", 3), ("accumulated_roughness", 1)):
        T[k] = f32(3, H, W, c)
    T["output_final"], T["output_denoised"] = f32(1, H, W, 3), f32(1, H, W, 3)
    T["accumulated_sample_count"] = torch.zeros(1, dtype=torch.int32, device=dev)
    for k, c in (("diffuse", 3), ("specular", 3), ("depth", 1), ("normal", 3), ("f0", 3), ("roughness", 1)):
        T["target_" + k] = torch.tensor(tg[k], dtype=torch.float32, device=dev).reshape(H, W, c).contiguous()
    T["grads_enabled"] = torch.ones(1, dtype=torch.bool, device=dev)
    T["total_num_calls"] = torch.zeros(1, dtype=torch.int32, device=dev)
    T["random_seeds"] = torch.zeros(H, W, 1, dtype=torch.int32, device=dev)
    T["num_accumulated_per_pixel"], T["num_traversed_per_pixel"] = torch.zeros(H, W, dtype=torch.int32, device=dev), torch.zeros(H, W, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    raw = cabi.RawRaytracer(W, H, N, {k: v.data_ptr() for k, v in T.items()}, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000,
                            device=torch.cuda.current_device(), stream=torch.cuda.current_stream().cuda_stream)
    assert raw.L.egr_debug_check_bvh(raw.ctx, raw.stream) == 0
    raw.update_bvh()
    raw.raytrace(False)  # call 1: images
    c1 = raw.counters()
    images = {k: T[k].clone() for k in ("output_rgb", "output_final", "output_depth", "output_normal", "output_total_transmittance", "random_seeds",
                                       "num_accumulated_per_pixel", "num_traversed_per_pixel")}
    raw.update_bvh()
    raw.raytrace(True)  # call 2: gradients
    c2 = raw.counters()
    assert c1.status == 0 and c2.status == 0 and c1.rays[0] == W * H and int(T["total_num_calls"]) == 2 and bool(T["grads_enabled"])
    assert c2.device_bytes > 0 and c2.arena_blocks_used > 0
    # egr_update_bvh_ex: unknown flags are refused with a message; EGR_UPDATE_FUSE_LIVE is the sequence the shim path below runs
    assert raw.L.egr_update_bvh_ex(raw.ctx, 2, raw.stream) != 0 and b"unknown flag" in raw.L.egr_last_error(raw.ctx)
    raw.update_bvh(fuse_live=True)
    # ---- the same through torch.classes.raytracer (the shim), reference call sequence of GaussianRaytracer.__call__
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000)
    camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **{k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()})
    with torch.no_grad():
        rt(camera, force_update_bvh=True)
    m = rt.cuda_module
    fb, st = m.get_framebuffer(), m.get_stats()
    for k in ("output_rgb", "output_final", "output_depth", "output_normal", "output_total_transmittance"):
        assert torch.equal(getattr(fb, k), images[k]), k  # bit for bit: the forward path has no unordered arithmetic
    assert torch.equal(m.get_metadata().random_seeds, images["random_seeds"])
    assert torch.equal(st.num_accumulated_per_pixel, images["num_accumulated_per_pixel"]) and torch.equal(st.num_traversed_per_pixel, images["num_traversed_per_pixel"])
    rt.zero_grad()
    ren.render(camera, rt)
    torch.cuda.synchronize()
    gs = m.get_gaussians()
    for k in cabi.GAUSSIAN_GRADS:  # float atomics: equal up to the order of the additions
        a, b = T[k], getattr(gs, k)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30, k
    raw.close()
