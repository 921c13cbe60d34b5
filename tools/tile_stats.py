import importlib, sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo")
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W,H,N=1920,1080,1_000_000
for VARIANT in ("init","trained"):
    g=syn.make_scene(N,VARIANT,seed=0); cam=syn.default_camera(); pc=ren.GaussianParams(g)
    rt=ren.GaussianRaytracer(pc,W,H,ppll_forward_size=400_000_000,ppll_backward_size=300_000_000); m=rt.cuda_module
    m.get_config().num_bounces.fill_(0)
    camera=ren.camera_from_c2w(cam["origin"],cam["c2w"],cam["fov"])
    with torch.no_grad(): rt(camera)
    st=m.get_stats()
    for name,img in (("composited hits",st.num_accumulated_per_pixel),("accepted candidates (inside ellipsoid)",st.num_traversed_per_pixel)):
        a=img.float().view(H//8,8,W//8,8)
        tmax=a.amax(dim=(1,3)).flatten(); tmean=a.mean(dim=(1,3)).flatten()
        print(VARIANT,name,"per-ray mean",float(a.mean()),"mean over tiles of the tile MAX",float(tmax.mean()),"ratio",float(tmax.mean()/a.mean()),"| batches of 8: mean per ray",float(torch.ceil(a/8).mean()),"mean over tiles of max",float(torch.ceil(tmax/8).mean()))
    # cost model of the selection: per lane sum over batches of list length = ceil(hits/8) * cnt ; tile cost = max over lanes vs mean over lanes
    hits=st.num_accumulated_per_pixel.float().view(H//8,8,W//8,8); cnt=st.num_traversed_per_pixel.float().view(H//8,8,W//8,8)
    work=(torch.ceil(hits/8)+1)*cnt
    print(VARIANT,"selection work (batches x list length): mean per lane",float(work.mean()),"mean over tiles of the max lane",float(work.amax(dim=(1,3)).mean()))
    del rt
