"""MI355X-native drop-in for the hot path of graphdeco-inria/editable-gaussian-reflections.

Mirrors editable_gauss_refl/__init__.py:15-27: `make_raytracer(...)` loads `libraytracer.so` with
`torch.classes.load_library` and returns `torch.classes.raytracer.Raytracer(...)`. The library is the
TORCH_LIBRARY shim over the HIP C ABI (include/egr_raytracer.h); there is no CPU fallback and no second
backend: if the native libraries are missing or no GPU is present this fails loudly.
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "build")
GAUSS_TRACER_PATH = os.path.join(BUILD_DIR, "libraytracer.so")
HIP_LIB_PATH = os.path.join(BUILD_DIR, "libegr_hip.so")
LOADED = False


def load_library():
    global LOADED
    if not LOADED:
        import torch

        if not (os.path.exists(GAUSS_TRACER_PATH) and os.path.exists(HIP_LIB_PATH)):
            raise RuntimeError(
                f"native libraries not built: expected {GAUSS_TRACER_PATH} and {HIP_LIB_PATH}; "
                "run `python -c 'import __graft_entry__ as g; g.build()'` (there is no Python/CPU fallback)")
        torch.classes.load_library(GAUSS_TRACER_PATH)
        LOADED = True


def make_raytracer(image_width: int, image_height: int, num_gaussians: int, ppll_forward_size: int = 180_000_000,
                   ppll_backward_size: int = 120_000_000):
    """Same signature and defaults as the reference's make_raytracer (editable_gauss_refl/__init__.py:15-27)."""
    import torch

    load_library()
    return torch.classes.raytracer.Raytracer(image_width, image_height, num_gaussians, ppll_forward_size, ppll_backward_size)
