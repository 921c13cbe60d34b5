#!/usr/bin/env python
"""Headline benchmark: Mrays/s forward+backward @1080p, 1M Gaussians (BASELINE.json metric) on MI355X.

One "step" = one training-iteration pass of the hot path behind the reference's own caller
(GaussianRaytracer.__call__ in grad mode: parameter export, target upload, update_bvh (instance snapshot +
LBVH refit), raytrace (forward + in-kernel loss gradient + backward), gradient all-reduce over ranks, gradient
import). A "ray" = one (pixel, bounce-step) traversal; rays/step come from the kernels' own counters.

  python bench.py                       # config C, 1 GPU: the literal dense-init cloud (`value`) and the trained-like cloud (`value_trained_like`)
  python bench.py --config B            # BASELINE config 2: 100k Gaussians, 1080p, forward only (measure_fps.py protocol)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Multi-GPU: the image is split into 16x16-pixel tiles dealt round-robin along a Z-curve to the ranks (the scene and the BVH are
replicated), THIS launch's gradients are summed with ONE RCCL all-reduce of a flat [22N] buffer -> "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ROUND = "r6"  # profiles/<ROUND>/pmc_summary.json holds this round's rocprofv3 --pmc passes of this command (tools/profile.sh), one entry per workload


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=50)  # ~1.5 s of load: the core clock needs about that long to ramp (measured)
    p.add_argument("--config", default="C", choices=["B", "C"], help="BASELINE.json config: C = 1M Gaussians forward+backward (metric), B = 100k forward only")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--gaussians", type=int, default=None)
    p.add_argument("--variant", default=None, choices=["trained", "init"], help="opacity of the synthetic cloud: trained-like 0.8 (bounces happen) or the literal dense-init 0.1 (config.py:44)")
    p.add_argument("--no-second-variant", action="store_true", help="config C at N=1 also times the other opacity variant (reported in `other_variant`); skip it")
    p.add_argument("--bounces", type=int, default=2)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", default="480x270")
    p.add_argument("--primary-steps", type=int, default=30, help="timed iterations of the primary-only (num_bounces = 0) leg; 0 skips it")
    p.add_argument("--profile-steps", type=int, default=9, help="extra untimed launches with per-kernel HIP events")
    p.add_argument("--strands", type=int, default=0, help="tile slices traced on separate HIP streams (0 = library default: 1); the per-kernel profile pass always uses 1")
    p.add_argument("--emulate-world", type=int, default=0, help="diagnostic: trace only rank 0's tiles of an N-rank partition on this one GPU (no collective)")
    p.add_argument("--emulate-rank", type=int, default=0, help="which rank's tiles --emulate-world traces")
    p.add_argument("--prewarm-seconds", type=float, default=0.0, help="keep the GPU busy with a torch matmul loop this long before the first launch (clock ramp; used under rocprofv3 so that the per-kernel averages are not carried by cold launches)")
    p.add_argument("--team-help", type=int, default=-1, help="egr_set_team_help: waves without tiles help their team mates' walks (1 / 0); -1 = the product's default (on: the list order reaches no output except the order of exact depth ties of bounce rays)")
    p.add_argument("--ppll-forward", type=int, default=400_000_000, help="forward capacity in the reference's 36-B entries (its own test uses 300M at 1536x1024; its default is 180M)")
    p.add_argument("--ppll-backward", type=int, default=300_000_000, help="backward capacity (reference default 120M)")
    p.add_argument("--forward-only", action="store_true", help="no-grad render instead of a training iteration (implied by --config B)")
    a = p.parse_args()
    if a.config == "B":
        a.forward_only = True
    if a.gaussians is None:
        a.gaussians = 100_000 if a.config == "B" else 1_000_000
    if a.variant is None:
        a.variant = "init"  # the north star's "synthetic dense-init Gaussian cloud" (init_opa 0.1, config.py:44); the trained-like cloud runs second
    return a


def algorithmic_bytes(kind, step, rays, hc, kc, pixels):
    """SURVEY.md 8d per-ray algorithmic bytes: forward 44*Hc (mean 12 + rotation 16 + scale 12 + opacity 4) + 40*Kc
    (rgb, normal, f0, roughness) [+76 B outputs in no-grad mode only]; backward step 0: 260*Kc, step>0: 176*Kc
    (parameter re-reads + 22/15 float atomics at 8 B each); +56 B/pixel targets in grad mode (charged to backward 0)."""
    if kind == "forward":
        return 44.0 * hc + 40.0 * kc
    if kind == "forward_nograd":
        return 44.0 * hc + 40.0 * kc + 76.0 * rays
    if kind == "backward":
        return (260.0 if step == 0 else 176.0) * kc + (56.0 * pixels if step == 0 else 0.0)
    raise ValueError(kind)


def main():
    a = parse()
    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU fallback)"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    local_rank = local_rank % torch.cuda.device_count()
    dist = None
    backend = "none"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm. EGR_DIST_BACKEND=gloo only exists to exercise this code path with 2 ranks on ONE GPU (tests/).
        backend = os.environ.get("EGR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
    ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
    W, H, N = a.width, a.height, a.gaussians
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
    camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
    dbg = (lambda *x: print(f"[rank {rank}]", *x, file=sys.stderr, flush=True)) if os.environ.get("EGR_BENCH_VERBOSE") else (lambda *x: None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_pair(dt, rays):  # max time / summed rays over ranks (host tensors with gloo, device tensors with RCCL)
        if world == 1:
            return dt, rays
        dev = "cuda" if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        tsum = torch.tensor([rays], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[0])

    if a.prewarm_seconds > 0:
        xw = torch.randn(4096, 4096, device="cuda")
        t_end = time.perf_counter() + a.prewarm_seconds
        while time.perf_counter() < t_end:
            for _ in range(20):
                xw = torch.nn.functional.normalize(xw @ xw, dim=1)
            torch.cuda.synchronize()
        del xw

    def run_variant(variant, with_profile, with_cpu):
        """Times `steps` passes of the hot path on the `variant` cloud; returns the pieces of the JSON line."""
        g = syn.make_scene(N, variant, seed=0)
        pc = ren.GaussianParams(g)
        # capacities: same meaning as the reference's ppll sizes (entries of 36 B); its own test uses 300M/200M at 1536x1024
        rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=a.ppll_forward, ppll_backward_size=a.ppll_backward, rank=rank, world_size=world)
        m = rt.cuda_module
        m.get_config().num_bounces.fill_(a.bounces)
        if a.strands > 0:
            m.set_strands(a.strands)
        if a.emulate_world > 1:
            assert world == 1
            m.set_partition(a.emulate_rank, a.emulate_world)
        if a.team_help >= 0:
            m.set_team_help(a.team_help == 1)

        def one_step():
            if a.forward_only:
                with torch.no_grad():
                    rt(camera)  # measure_fps.py:27-52: render under no_grad, no BVH update between frames
            else:
                rt.zero_grad()
                rt(camera, target_diffuse=images["diffuse_image"], target_specular=images["specular_image"], target_depth=images["depth_image"],
                   target_normal=images["normal_image"], target_roughness=images["roughness_image"], target_f0=images["f0_image"])

        dbg("setup done", variant)
        for _ in range(a.warmup):
            one_step()
        barrier()
        m.reset_lifetime_counters()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one_step()
        barrier()
        dt = time.perf_counter() - t0
        c = m.get_counters()
        res = {"status": int(c[11])}
        dt, total_rays = reduce_pair(dt, float(c[9]))
        res["ms_per_step"] = dt / a.steps * 1e3
        res["value"] = total_rays / dt / 1e6
        # "report also primary-only (num_bounces = 0: exactly P rays), because bounce termination is scene dependent"
        if a.bounces > 0 and a.primary_steps > 0:
            m.get_config().num_bounces.fill_(0)
            for _ in range(10):
                one_step()
            barrier()
            m.reset_lifetime_counters()
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.primary_steps):
                one_step()
            barrier()
            dtp = time.perf_counter() - t0
            dtp, rays_p = reduce_pair(dtp, float(m.get_counters()[9]))
            res["primary_only"] = {"value": round(rays_p / dtp / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dtp / a.primary_steps * 1e3, 4), "steps": a.primary_steps,
                                   "rays_per_step": int(rays_p / a.primary_steps), "note": "num_bounces = 0: one ray per pixel, same iteration otherwise"}
            m.get_config().num_bounces.fill_(a.bounces)
            for _ in range(5):
                one_step()

        # the same iteration with team help OFF (egr_set_team_help(0): single-wave workgroups, every launch reproducible bit for bit - with help, the
        # product's default, the order of exact depth ties of bounce rays is timing): what the default is worth on the same box in the same process
        if world == 1 and a.emulate_world <= 1 and a.team_help < 0 and a.primary_steps > 0 and not a.forward_only:
            m.set_team_help(False)
            for _ in range(15):
                one_step()
            barrier()
            m.reset_lifetime_counters()
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.primary_steps):
                one_step()
            barrier()
            dth = time.perf_counter() - t0
            res["team_help_off"] = {"value": round(float(m.get_counters()[9]) / dth / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dth / a.primary_steps * 1e3, 4), "steps": a.primary_steps,
                                    "note": "egr_set_team_help(0): no wave helps another; launches reproducible bit for bit"}
            m.set_team_help(True)
            for _ in range(5):
                one_step()

        cpu = None
        if with_cpu and rank == 0 and world == 1:  # reported at N=1 only (the other ranks would sit in the next collective meanwhile)
            from oracle import oracle as orc

            cw, ch = (int(x) for x in a.cpu_sample.split("x"))
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            o = orc.Oracle(cw, ch, threads=cores)
            o.set_camera(cam["origin"], cam["c2w"], cam["fov"])
            o.set_config(num_bounces=a.bounces, **syn.TRAIN_LOSS_WEIGHTS)
            o.set_gaussians(g)
            o.update_bvh()  # the build (rebuild_bvh); the timed reps refit like the reference's per-iteration update
            ctg = syn.make_targets(cw, ch)
            gshape = {"dL_drgb": 3, "dL_dnormal": 3, "dL_df0": 3, "dL_droughness": 1, "dL_dopacity": 1, "dL_dscale": 3, "dL_dmean": 3, "dL_drotation": 4, "total_weight": 1}
            gbuf = {k: np.zeros((N, w), np.float64) for k, w in gshape.items()}  # the caller's gradient tensors, allocated once
            o.raytrace(not a.forward_only, targets=ctg, grads_into=gbuf)  # warm-up
            rates, t_start = [], time.perf_counter()
            while len(rates) < 5 or (time.perf_counter() - t_start < 10.0 and len(rates) < 25):
                t1 = time.perf_counter()
                if not a.forward_only:
                    o.update_bvh()
                out = o.raytrace(not a.forward_only, targets=ctg, grads_into=gbuf)
                rates.append(int(out["effective_steps"].sum()) / (time.perf_counter() - t1) / 1e6)
            cpu = {"value": round(float(np.median(rates)), 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"CPU restatement of the reference algorithm (oracle/), same scene+camera+config at {cw}x{ch}, "
                             f"{'forward' if a.forward_only else 'update_bvh (refit) + forward+backward'}, median of {len(rates)} reps after 1 warm-up, "
                             f"OpenMP over 64-pixel blocks, per-thread gradient tables"}
        res["cpu_baseline"] = cpu

        # ---- per-kernel timing pass (HIP events on the launch stream, untimed region) -> roofline of the dominant kernel
        roof, kern, acc = None, {}, {}
        if with_profile:
            m.set_strands(1)  # one kernel at a time: a kernel's duration is then its exclusive time on the GPU
            for _ in range(min(a.warmup, 30) if a.profile_steps > 0 else 0):  # the CPU baseline above left the GPU idle: ramp the clocks again
                one_step()
            m.enable_timing(rank == 0)
            for _ in range(a.profile_steps):  # every rank runs these steps (they contain the all-reduce); rank 0 reads the stamps
                one_step()
                torch.cuda.synchronize()
                if rank == 0:
                    for name, ms in m.last_kernel_ms():
                        acc.setdefault(name, []).append(ms)
                    acc.setdefault("update_bvh", []).append(m.last_update_bvh_ms())
                    acc.setdefault("raytrace_total", []).append(m.last_raytrace_ms())
            m.enable_timing(False)
            cc = m.get_counters()
            # Hc of SURVEY 8d = gaussians whose CUBE the segment overlaps (what the reference's intersection program is invoked for),
            # measured on the GPU by ONE exact-statistics launch of the same frame (cube boxes, egr_set_exact_stats); the default
            # tree bounds ellipsoids; a default launch counts the candidates inside their ellipsoid (`inside_ellipsoid_per_ray`). Every rank does this (same call sequence).
            m.set_exact_stats(True)
            m.update_bvh()
            m.get_metadata().total_num_calls.sub_(1)  # the same jitter / bounce random stream as the last profiled launch
            with torch.no_grad():
                m.raytrace()
            ce = m.get_counters()
            m.set_exact_stats(False)
            m.update_bvh()
            barrier()
        if rank == 0 and with_profile and a.profile_steps > 0:
            kern = {k: float(np.median(v)) for k, v in acc.items() if v and v[0] >= 0}
            rays, cand_eval, comp = cc[0:3], list(cc[3:6]), cc[6:9]
            cand = [ce[3 + s] * (rays[s] / max(ce[s], 1)) for s in range(3)]  # (grad / no-grad launches trace the same rays; guard anyway)
            pixels_rank = rays[0]
            fk = "forward_nograd" if a.forward_only else "forward"
            fwd_ref = sum(algorithmic_bytes(fk, s, rays[s], cand[s], comp[s], pixels_rank) for s in range(3))
            cands = {"forward_chain": fwd_ref}
            if not a.forward_only:
                cands["backward_chain"] = sum(algorithmic_bytes("backward", s, rays[s], cand[s], comp[s], pixels_rank) for s in range(3))
            dom = max((k for k in cands if k in kern), key=lambda k: kern[k])  # the longest kernel of the launch IS the reported one
            # HBM traffic per kernel: rocprofv3 --pmc passes of THIS workload collected in THIS round by tools/profile.sh (separate
            # FETCH_SIZE / WRITE_SIZE passes; units of KiB; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md) and committed
            # as profiles/<round>/pmc_summary.json, keyed by workload. bench.py cannot run the profiler around itself.
            pmc, pmc_round = {}, ROUND
            try:
                pmc_round = next((r for r in (ROUND, "r5") if os.path.exists(os.path.join(ROOT, "profiles", r, "pmc_summary.json"))), ROUND)  # (this round's passes once they are committed; the traffic source names the round)
                pj = os.path.join(ROOT, "profiles", pmc_round, "pmc_summary.json")
                if os.path.exists(pj) and world == 1 and a.emulate_world <= 1 and N == (100_000 if a.config == "B" else 1_000_000) and (W, H) == (1920, 1080):
                    pmc = json.load(open(pj)).get(f"{a.config}_{variant}", {})
            except Exception:
                pmc = {}

            def roof_of(k):
                achieved = cands[k] / (kern[k] * 1e-3) / 1e9
                pm, traffic, src = pmc.get(k), None, None
                if pm and "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                    traffic = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
                    src = f"profiles/{pmc_round}/pmc_summary.json[{a.config}_{variant}] (rocprofv3 --pmc passes of round {pmc_round}, same workload; 2 x FETCH_SIZE + WRITE_SIZE per launch)"
                r = {"bound": "hbm", "kernel": k, "longest_kernel_of_the_launch": k == dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                     "traffic": traffic, "traffic_over_algorithmic": round(traffic / cands[k], 3) if traffic else None, "traffic_source": src,
                     "avg_kernel_ms": round(kern[k], 4), "algorithmic_bytes_per_launch": cands[k]}
                if k == "backward_chain":
                    # what actually bounds this kernel: it sends its gradients as 64-B records of 16 non-returning fp32 atomic adds, and the GPU
                    # retires 21.1 G such records / s whatever their footprint (tools/atomic_rate.hip, profiles/r3/atomic_rate.txt)
                    rec = float(cc[13])
                    r.update({"atomic_records_per_launch": int(rec), "atomic_record_rate_G_per_s": round(rec / (kern[k] * 1e-3) / 1e9, 2),
                              "atomic_record_peak_G_per_s": 21.1, "atomic_record_frac": round(rec / (kern[k] * 1e-3) / 21.1e9, 4),
                              "atomic_record_peak_source": "tools/atomic_rate.hip on MI355X: 16-lane (64-B) fp32 atomic-add records to pseudo-random gradient rows, 0.1 MB .. 268 MB footprint"})
                return r

            roof = roof_of(dom)
            roof.update({"rays_per_step": [int(x) for x in rays], "Hc_per_ray": [round(cand[s] / max(rays[s], 1), 2) for s in range(3)],
                         "Hc_source": "one exact-statistics launch of the same frame on the GPU (cube boxes; reference-defined count)",
                         "inside_ellipsoid_per_ray": [round(cand_eval[s] / max(rays[s], 1), 2) for s in range(3)],
                         "Kc_per_ray": [round(comp[s] / max(rays[s], 1), 2) for s in range(3)],
                         "strands_timed_region": a.strands if a.strands > 0 else 1, "strands_profile_pass": 1,
                         "whole_launch_GBps": round(sum(cands.values()) / (kern.get("raytrace_total", 1e9) * 1e-3) / 1e9, 2),
                         "other_kernels": [roof_of(k) for k in cands if k != dom and k in kern]})
            res["device_bytes"] = int(cc[14])
        # what a rank's iteration is made of (ranks of a partition, real or emulated): the two chains shrink with the rank's share of the image, the
        # exchange step is the all-reduce + fold of the [22N] buffer (timed alone, HIP events, every rank takes part), everything else - refit, live
        # records, gradient gather, the caller's copies, launch gaps - is replicated on every rank
        if with_profile and a.profile_steps > 0 and (world > 1 or a.emulate_world > 1):
            ar_ms = None
            if world > 1:
                par = importlib.import_module("editable-gaussian-reflections_amd.parallel")
                gg = m.get_gaussians()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    par.all_reduce_launch_delta(gg.grad_flat, gg.grad_delta)
                barrier()
                e0.record()
                for _ in range(10):
                    par.all_reduce_launch_delta(gg.grad_flat, gg.grad_delta)
                e1.record()
                torch.cuda.synchronize()
                ar_ms = e0.elapsed_time(e1) / 10.0
                rt.zero_grad()
            if rank == 0:
                chains = float(np.median(acc.get("forward_chain", [0.0]))) + float(np.median(acc.get("backward_chain", [0.0])))
                res["multi_gpu"] = {"iteration_ms": round(res["ms_per_step"], 4), "chains_ms": round(chains, 4), "allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
                                    "replicated_ms": round(res["ms_per_step"] - chains - (ar_ms or 0.0), 4),
                                    "note": "chains = forward + backward chain of this rank's tiles (profile pass, one strand); allreduce = all-reduce + fold of the launch's [22N] fp32 buffer, timed alone; "
                                            "replicated = the rest of the iteration: refit + live records + gradient gather + the caller's copies + launch gaps, the same on every rank"}
        res["roofline"], res["kernel_ms"] = roof, {k: round(v, 4) for k, v in kern.items()}
        del rt, pc
        torch.cuda.empty_cache()
        return res

    what = "forward only (no_grad, no BVH update between frames: measure_fps.py protocol)" if a.forward_only else \
        "one training iteration: export + update_bvh + forward + backward + grad import"
    label = {"trained": "trained-like opacity 0.8: reflection bounces happen, three steps per pixel", "init": "literal dense-init opacity 0.1 (config.py:44): Kc ~ 22 per ray, bounces mostly die"}
    vkey = {"trained": "value_trained_like", "init": "value_dense_init"}
    main_res = run_variant(a.variant, True, not a.no_cpu_baseline)
    other = None
    if a.config == "C" and world == 1 and not a.no_second_variant and a.emulate_world <= 1 and not a.forward_only:
        ov = "init" if a.variant == "trained" else "trained"
        r2 = run_variant(ov, True, False)
        other = {"variant": ov, "value": round(r2["value"], 3), "unit": "Mrays/s", "ms_per_step": round(r2["ms_per_step"], 4), "status": r2["status"],
                 "roofline": r2["roofline"], "kernel_ms": r2["kernel_ms"], "device_bytes": r2.get("device_bytes"), "primary_only": r2.get("primary_only"), "team_help_off": r2.get("team_help_off")}
    if rank == 0:
        line = {
            "metric": "Mrays/s fwd+bwd @1080p, 1M Gaussians" if a.config == "C" and not a.forward_only else "Mrays/s fwd-only @1080p",
            "value": round(main_res["value"], 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(main_res["ms_per_step"], 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {a.config}: synthetic dense-init room+spheres cloud, HEADLINE variant = {a.variant} ({label[a.variant]}; "
                                   f"the other variant is in `other_variant`), N={N}, {W}x{H}, {what}, num_bounces={a.bounces}, jitter on, reference default config",
                       "variant": a.variant, "gaussians": N, "width": W, "height": H, "parallelism": f"image tiles x{world} + 1 all-reduce of the launch's [22N] gradients", "team_help": "product default (on)" if a.team_help < 0 else bool(a.team_help == 1)},
            "roofline": main_res["roofline"], "cpu_baseline": main_res["cpu_baseline"], "kernel_ms": main_res["kernel_ms"],
            "value_primary_only": (main_res.get("primary_only") or {}).get("value"), "primary_only": main_res.get("primary_only"),
            "value_team_help_off": (main_res.get("team_help_off") or {}).get("value"), "team_help_off": main_res.get("team_help_off"),
            "multi_gpu": main_res.get("multi_gpu"), "other_variant": other, "status": main_res["status"], "psnr_vs_optix": None, "device_bytes": main_res.get("device_bytes"),
            vkey[a.variant]: round(main_res["value"], 3), vkey["init" if a.variant == "trained" else "trained"]: (other or {}).get("value"),
            "note": "vs_baseline null: the reference publishes no throughput number; PSNR vs OptiX is unmeasurable here (no NVIDIA "
                    "hardware), parity is against the CPU oracle (tests/).",
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
