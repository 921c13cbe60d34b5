import importlib, sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W,H,N=1920,1080,1_000_000
VARIANT=os.environ.get("VARIANT","trained")
g=syn.make_scene(N,VARIANT,seed=0); cam=syn.default_camera(); pc=ren.GaussianParams(g)
rt=ren.GaussianRaytracer(pc,W,H,ppll_forward_size=400_000_000,ppll_backward_size=300_000_000); m=rt.cuda_module
camera=ren.camera_from_c2w(cam["origin"],cam["c2w"],cam["fov"])
if int(os.environ.get("EMU_WORLD","1"))>1: m.set_strands(1); m.set_partition(int(os.environ.get("EMU_RANK","0")),int(os.environ["EMU_WORLD"]))
if os.environ.get("GRADS"):
    tg=syn.make_targets(W,H)
    camera=ren.camera_from_c2w(cam["origin"],cam["c2w"],cam["fov"],**{k+"_image": torch.tensor(v).cuda().moveaxis(-1,0).contiguous() for k,v in tg.items()})
    for _ in range(3):
        rt.zero_grad(); ren.render(camera, rt)
else:
    with torch.no_grad(): rt(camera)
c=m.get_counters(); print("rays",c[0:3],"Hc",[c[3+i]/max(c[i],1) for i in range(3)],"Kc",[c[6+i]/max(c[i],1) for i in range(3)])
st = m.get_stats().num_traversed_per_pixel.float().flatten()
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], device=st.device)
idx = (q * (st.numel() - 1)).long()
ss = st.sort().values
print("Hc per pixel (all steps) quantiles 50/90/99/99.9/99.99%:", ss[idx].tolist(), "max", float(ss[-1]), "mean", float(st.mean()))
t = st.view(H // 8, 8, W // 8, 8).amax(dim=(1, 3)).flatten().sort().values
print("per-tile max-lane Hc quantiles:", t[(q * (t.numel() - 1)).long()].tolist(), "max", float(t[-1]), "mean of tile max", float(t.mean()))
