"""Builds libsecp256k1_zkp_amd.so: the engine's translation units (csrc/engine_*.hip, one per kernel family -- see csrc/engine_internal.h)
are compiled for gfx950 side by side and linked into one shared object.  hipcc cross-compiles without a GPU.

    python -m secp256k1_zkp_amd.build_lib [-o OUT.so] [--force] [-DS2K_DIAG ...]      (extra -D / -m flags go to every compile)

Diagnostic variants (-DS2K_DIAG: environment overrides of the MSM launcher and kernels with parts switched off; -DS2K_PROF: region
counters) get their own object directory, so they never disturb the product library's objects."""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
UNITS = ["engine_core", "engine_rangeproof", "engine_msm", "engine_msm_many", "engine_bppp", "engine_halfagg"]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fvisibility=hidden"]
DEFAULT_LIB = os.path.join(HERE, "libsecp256k1_zkp_amd.so")


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "secp256k1_zkp_amd.h")]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build(out=DEFAULT_LIB, extra=(), force=False, verbose=True):
    extra = list(extra)
    tag = "product" if not extra else hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]
    objdir = os.path.join(CSRC, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    hdrs = headers()
    jobs = []
    for u in UNITS:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if force or _newer(obj, hdrs + [src]):
            jobs.append([HIPCC] + BASE_FLAGS + extra + ["-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    if jobs or force or _newer(out, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    out = DEFAULT_LIB
    if "-o" in a:
        i = a.index("-o"); out = os.path.abspath(a[i + 1]); del a[i:i + 2]
    force = "--force" in a
    a = [x for x in a if x != "--force"]
    print(build(out, a, force))
