"""The N > 1 path executed with HIP gradients: two processes share the one GPU of the test box, rendezvous over
gloo (RCCL needs one device per rank; 8-GPU runs belong to the driver), each traces its tile partition and the product's own
exchange step sums the gradients."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, port, env_extra=None, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_sum_to_the_single_rank_gradients():
    r = _launch([os.path.join(ROOT, "tests", "multirank_worker.py")], 29531)
    assert r.returncode == 0 and "MULTIRANK_OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.timeout(900)
def test_team_help_does_not_drift_training():
    """egr_set_team_help makes a ray's list order timing-dependent (deviation (a)): 20 training iterations on two ranks with help on
    stay a bounded perturbation of the help-off run: measured 18x the run-to-run noise of the float atomics, final renders 50 dB (tests/teamhelp_worker.py)."""
    r = _launch([os.path.join(ROOT, "tests", "teamhelp_worker.py")], 29537)
    assert r.returncode == 0 and "TEAMHELP_OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.timeout(600)
def test_exchange_step_through_rccl_at_world_size_1():
    """RCCL itself (backend "nccl"): communicator initialisation, the all-reduce on the device-resident per-launch buffer behind the
    launch on torch's stream, the fold, the import, and the evaluation all-gather, on a one-rank group (the box has one GPU)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                       timeout=540, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_line():
    """bench.py under torch.distributed.run exactly as the driver launches it (reduced sizes, gloo because both ranks sit on cuda:0)."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--width", "480", "--height", "272",
                 "--gaussians", "30000", "--profile-steps", "1", "--no-cpu-baseline"], 29533, env_extra={"EGR_DIST_BACKEND": "gloo", "EGR_TEAM_HELP": "1"})  # (the product's default for such ranks: help on)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["status"] == 0 and d["config"]["width"] == 480
