"""the denoiser stand-in (csrc/denoise.hip, an edge-avoiding a-trous wavelet filter on output_final guided by
output_normal) against a plain PyTorch fp32 implementation of the same filter. Parity with the OptiX AI denoiser is unpinned."""
import importlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
PKG = "editable-gaussian-reflections_amd"


def atrous_reference(img, normal, sigma_c=0.6, sigma_n=0.3):
    """img, normal: [H,W,3] float32 GPU tensors. Same 5 passes / 25 taps / weights as k_atrous, written with torch ops."""
    h = torch.tensor([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16], dtype=torch.float32, device=img.device)
    H, W, _ = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, device=img.device), torch.arange(W, device=img.device), indexing="ij")
    cur = img
    for p in range(5):
        hole = 1 << p
        acc = torch.zeros_like(cur)
        wsum = torch.zeros(H, W, 1, device=img.device)
        for j in range(-2, 3):
            for i in range(-2, 3):
                yy, xx = ys + j * hole, xs + i * hole
                ok = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)).unsqueeze(-1).float()
                yc, xc = yy.clamp(0, H - 1), xx.clamp(0, W - 1)
                tap, ntap = cur[yc, xc], normal[yc, xc]
                wc = torch.exp(-((tap - cur) ** 2).sum(-1, keepdim=True) / sigma_c ** 2)
                wn = torch.exp(-((ntap - normal) ** 2).sum(-1, keepdim=True) / sigma_n ** 2)
                w = h[i + 2] * h[j + 2] * wc * wn * ok
                acc += w * tap
                wsum += w
        cur = acc / wsum
        sigma_c *= 0.5
    return cur


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    ren = importlib.import_module(PKG + ".renderer")
    syn = importlib.import_module(PKG + ".synthetic")
    pc = ren.GaussianParams(syn.make_scene(3000, "trained", seed=2))
    return ren, syn, ren.GaussianRaytracer(pc, 160, 96)


def test_matches_torch_reference_on_a_render(rt):
    ren, syn, r = rt
    cam = syn.default_camera()
    camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
    with torch.no_grad():
        r(camera, denoise=True)
    fb = r.cuda_module.get_framebuffer()
    assert fb.output_denoised.shape == fb.output_final.shape
    H, W = r.image_height, r.image_width
    final, out = fb.output_final.reshape(H, W, 3), fb.output_denoised.reshape(H, W, 3)
    ref = atrous_reference(final.clone(), fb.output_normal[0].clone())
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((out - final).abs().max()) > 1e-4  # it is not the old copy


def test_properties_constant_noise_and_edges(rt):
    ren, syn, r = rt
    m = r.cuda_module
    fb = m.get_framebuffer()
    H, W = r.image_height, r.image_width
    final_view, den_view = fb.output_final.reshape(H, W, 3), fb.output_denoised.reshape(H, W, 3)
    g = torch.Generator(device="cuda").manual_seed(0)
    # constant image + constant normals -> unchanged
    fb.output_final.fill_(0.37)
    fb.output_normal.zero_()
    fb.output_normal[0, :, :, 2] = 1.0
    m.denoise()
    assert float((den_view - 0.37).abs().max()) < 1e-6
    # flat region with noise: variance drops a lot; a normal discontinuity in the middle keeps the two halves apart
    base = torch.zeros(H, W, 3, device="cuda")
    base[:, W // 2:] = 1.0
    noisy = base + 0.05 * torch.randn(H, W, 3, device="cuda", generator=g)
    final_view.copy_(noisy)
    fb.output_normal[0, :, W // 2:, 2] = 0.0
    fb.output_normal[0, :, W // 2:, 0] = 1.0
    m.denoise()
    out = den_view
    left, right = out[:, : W // 2], out[:, W // 2:]
    assert float(left.std()) < 0.012 and float(right.std()) < 0.012  # 0.05 -> ~0.005
    assert abs(float(left.mean())) < 0.01 and abs(float(right.mean()) - 1.0) < 0.01  # no bleeding across the edge
    ref = atrous_reference(noisy, fb.output_normal[0].clone())
    assert float((out - ref).abs().max()) < 2e-5
