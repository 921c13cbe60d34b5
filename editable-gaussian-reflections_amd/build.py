"""In-tree build of the two native artefacts (no cmake, no JIT cache):

  build/libegr_hip.so    hipcc --offload-arch=gfx950   csrc/{trace,bvh,api}.hip         the C-ABI product (include/egr_raytracer.h)
  build/libraytracer.so  g++ against the installed torch  csrc/torch_binding.cpp      TORCH_LIBRARY(raytracer) shim

hipcc cross-compiles gfx950 without a GPU; both .so files travel to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "build")
ROOT = os.path.dirname(HERE)
HIP_LIB = os.path.join(OUT, "libegr_hip.so")
TORCH_LIB = os.path.join(OUT, "libraytracer.so")

HIPCC = os.environ.get("HIPCC", shutil.which("hipcc") or "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: on gfx950 a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 takes longer than the two scalar instructions it replaces (tools/ubench/valu_rate.hip:
# 6.9 vs 2 x 3.1 cycles per SIMD), and the SLP vectoriser packs every adjacent pair of fp32 operations it finds (both chains -5 ... -7 % without it)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wno-unused-result"]
# (tuning knobs - EGR_GPOP, EGR_FPOP, EGR_PSTK, EGR_TEAM, EGR_BOX, EGR_DONATE_MIN, EGR_FWD_WAVES, EGR_BWD_WAVES, EGR_BWD_TEAM, EGR_GT_SLOTS, EGR_ORDER_BUCKETS, EGR_ORDER_SHIFT:
# numeric constants with an #ifndef default in csrc/ - are set for a sweep through EGR_EXTRA_FLAGS="-DEGR_GPOP=4"; the alternative code paths that rounds 1-5 switched
# between at build time are settled and gone: profiles/HISTORY.md has their measurements)
if os.environ.get("EGR_EXTRA_FLAGS"):  # compiler-flag experiments, e.g. "-mllvm -amdgpu-sched-strategy=iterative-minreg"
    HIP_FLAGS += os.environ["EGR_EXTRA_FLAGS"].replace(",", " ").split()
if os.environ.get("EGR_TASK_TIMES"):  # diagnostic: per-task walk / composite time of one step in the stats images (tools/task_times.py)
    HIP_FLAGS.append("-DEGR_TASK_TIMES=" + os.environ["EGR_TASK_TIMES"])
if os.environ.get("EGR_DEBUG_PIXEL"):
    HIP_FLAGS.append("-DEGR_DEBUG_PIXEL=" + os.environ["EGR_DEBUG_PIXEL"])
if os.environ.get("EGR_DEBUG_LIST"):
    HIP_FLAGS.append("-DEGR_DEBUG_LIST=1")
if os.environ.get("EGR_TRAVERSAL_STATS"):
    HIP_FLAGS.append("-DEGR_TRAVERSAL_STATS=1")
HIP_SOURCES = ["trace.hip", "bvh.hip", "api.hip", "knn.hip", "step.hip", "denoise.hip"]
EXTRA_FLAGS = {}
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))) + [os.path.join(ROOT, "include", "egr_raytracer.h")]  # every object depends on all of them


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("native build failed: " + os.path.basename(cmd[0]))
    return r.stdout


def build_hip(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    relink = force
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + HEADERS):
            if verbose:
                print("hipcc", src, flush=True)
            _run([HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])
            relink = True
        objs.append(o)
    if relink or not os.path.exists(HIP_LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", HIP_LIB])
    return HIP_LIB


def build_torch(force=False, verbose=False):
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(CSRC, "torch_binding.cpp")
    if not (force or _newer(TORCH_LIB, [src, HIP_LIB] + HEADERS)):
        return TORCH_LIB
    tdir = os.path.dirname(torch.__file__)
    inc = cpp_extension.include_paths()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += ["-I" + p for p in inc] + ["-I" + os.path.join(rocm, "include")]
    cmd += [src, "-o", TORCH_LIB, "-L" + os.path.join(tdir, "lib"), "-L" + OUT, "-legr_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-Wl,-rpath,$ORIGIN", "-Wl,--no-as-needed"]
    if verbose:
        print("g++ torch_binding.cpp", flush=True)
    _run(cmd)
    return TORCH_LIB


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_torch(force, verbose)
    return HIP_LIB, TORCH_LIB


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
