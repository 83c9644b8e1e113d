"""Build libpdfops.so (hand-written HIP for gfx950) in-tree with hipcc.

One object per translation unit under csrc/, linked into ``pointcloudpdf_amd/lib/libpdfops.so``.
The geometry TUs (kNN, FPS) are compiled with ``-ffp-contract=off``: their distances must be the
as-written IEEE fp32 expression of the reference kernels ("Arithmetic").
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libpdfops.so")
OBJDIR = os.path.join(HERE, "build")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
# per-TU extra flags
EXTRA = {
    "knn_query.hip": ["-ffp-contract=off"],
    "sampling.hip": ["-ffp-contract=off"],
    "sampling_bucketed.hip": ["-ffp-contract=off"] + (["-DFPS_PROFILE"] if os.environ.get("PDFOPS_FPS_PROFILE") else []) + ([f"-DPDF_FPS_MW_UNROLL={os.environ['PDFOPS_FPS_MW_UNROLL']}"] if os.environ.get("PDFOPS_FPS_MW_UNROLL") else []),
    "knn_grid.hip": ["-ffp-contract=off"],
    "ball_query.hip": ["-ffp-contract=off"],
}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libpdfops.so")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libpdfops.so. Returns the library path."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "pdfops.h"))
    srcs = sources()
    objs = []
    jobs = []
    for src in srcs:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [hipcc] + COMMON + EXTRA.get(src, []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIBPATH, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", LIBPATH])
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
