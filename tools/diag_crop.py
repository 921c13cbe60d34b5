"""Diagnostic (GPU box): the config-C crop gradient check of tests/test_hip_configs.py with everything printed - per-tensor errors of
HIP vs the fp32 oracle, vs the fp64 oracle, fp32 oracle vs fp64 oracle (the noise floor of evaluating the reference's formulas in fp32
at this scene scale), the pixels whose hit counts differ, the errors without their tiles, and the gaussian that carries the worst error.
    python tools/diag_crop.py [init|trained]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
from hip_common import GRAD_KEYS, OUT_KEYS, cam_obj, generic_targets, hip_outputs, make_pair
from test_hip_configs import CROP_TILES
PKG = "editable-gaussian-reflections_amd"
syn = importlib.import_module(PKG + ".synthetic"); ren = importlib.import_module(PKG + ".renderer"); par = importlib.import_module(PKG + ".parallel")
variant = sys.argv[1] if len(sys.argv) > 1 else "init"
W, H, N = 1920, 1080, int(os.environ.get("DIAG_N", 1_000_000))
g = syn.make_scene(N, variant, seed=0); cam = syn.default_camera(); tg = generic_targets(syn, W, H)
rt, o = make_pair(ren, orc, g, cam, W, H, fwd=400_000_000, bwd=300_000_000)
m = rt.cuda_module; camt = cam_obj(ren, cam, tg)
mtx, mty = par.macro_tiles(W, H); M = mtx * mty
owner = par.tile_owner(W, H, M).reshape(-1); K = 5
def tile_mask(tiles):
    mask = np.zeros((H, W), bool)
    for mx, my in tiles: mask[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] = True
    return mask
hits_h = np.zeros((3, H, W), np.int32); per_tile = {}
m.set_rays_per_task(64)
rt.zero_grad(); m.get_gaussians().total_weight.zero_()
for mx, my in CROP_TILES:
    m.set_partition(int(owner[my * mtx + mx]), M); m.get_metadata().total_num_calls.fill_(K - 1)
    before = m.get_gaussians().grad_flat.clone()
    ren.render(camt, rt)
    d = m.get_gaussians().grad_flat - before; nz = d.nonzero().reshape(-1)
    per_tile[(mx, my)] = (nz.cpu().numpy(), d[nz].cpu().numpy().astype(np.float64))
    hits_h += m.debug_step_hits().numpy() * tile_mask([(mx, my)])[None]
m.set_partition(0, 1); m.set_rays_per_task(0)
def hip_sum(tiles):
    got = np.zeros(22 * N)
    for t in tiles: np.add.at(got, per_tile[t][0], per_tile[t][1])
    return {k: v.numpy() for k, v in par.split_flat(torch.from_numpy(got), N).items()}
def oracle_on(oo, tiles):
    oo.set_pixel_mask(tile_mask(tiles)); oo.total_num_calls = K - 1
    r = oo.raytrace(True, targets=tg); oo.set_pixel_mask(None); return r
o64 = orc.Oracle(W, H, double=True); o64.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o64.set_gaussians(g)
o64.set_config(**{k: o.config[k] for k in o.config}); o64.update_bvh()
def show(tag, a, b, scale):
    print(tag, {k: f"{np.abs(a[k] - b[k]).max() / np.abs(scale[k]).max():.1e}" for k in GRAD_KEYS}, flush=True)
ref32, ref64, hip = oracle_on(o, CROP_TILES), oracle_on(o64, CROP_TILES), hip_sum(CROP_TILES)
show("HIP   vs orc32 :", hip, ref32, ref32)
show("HIP   vs orc64 :", hip, ref64, ref32)
show("orc32 vs orc64 :", ref32, ref64, ref32)
mask = tile_mask(CROP_TILES)
for name, other in (("orc32", ref32), ("orc64", ref64)):
    diff = np.any(hits_h != other["num_composited_per_step"], axis=0) & mask
    ys, xs = np.nonzero(diff)
    print(f"pixels with other hit counts than {name}:", [(int(x), int(y), hits_h[:, y, x].tolist(), other["num_composited_per_step"][:, y, x].tolist()) for x, y in zip(xs, ys)])
d3264 = np.any(ref32["num_composited_per_step"] != ref64["num_composited_per_step"], axis=0)
ys, xs = np.nonzero(d3264)
print("pixels where orc32 and orc64 differ in hit counts:", [(int(x), int(y), ref32["num_composited_per_step"][:, y, x].tolist(), ref64["num_composited_per_step"][:, y, x].tolist()) for x, y in zip(xs, ys)])
diff = np.any(hits_h != ref32["num_composited_per_step"], axis=0) & mask
ys, xs = np.nonzero(diff)
listed = sorted({(int(x) // 16, int(y) // 16) for x, y in zip(xs, ys)})
kept = [t for t in CROP_TILES if t not in listed]
r32k, r64k, hk = oracle_on(o, kept), oracle_on(o64, kept), hip_sum(kept)
print("without the", len(listed), "tiles of the listed pixels:")
show("HIP   vs orc32 :", hk, r32k, ref32)
show("HIP   vs orc64 :", hk, r64k, ref32)
show("orc32 vs orc64 :", r32k, r64k, ref32)
for key in ("dL_drotation", "dL_dmean", "dL_dscale"):
    e = np.abs(hk[key] - r32k[key]).max(axis=1); gid = int(e.argmax())
    tiles = [t for t in kept if (gid * hk[key].shape[1] + {"dL_drotation": 17 * N, "dL_dmean": 14 * N, "dL_dscale": 11 * N}[key]) in set(per_tile[t][0].tolist()[:0])]
    print(key, "worst gaussian", gid, "hip", hk[key][gid], "orc32", r32k[key][gid], "orc64", r64k[key][gid], "scale", np.exp(g["scale"][gid]), "max of tensor", np.abs(ref32[key]).max())
# per-tile view: one tile at a time, which tiles are above the bar
for t in kept:
    a, b = hip_sum([t]), oracle_on(o, [t])
    e = {k: np.abs(a[k] - b[k]).max() / np.abs(ref32[k]).max() for k in ("dL_dmean", "dL_dscale", "dL_drotation", "dL_drgb")}
    if max(e.values()) > 5e-4: print("tile", t, {k: f"{v:.1e}" for k, v in e.items()})
