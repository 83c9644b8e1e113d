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
    "window_edges.hip": ["-ffp-contract=off"],   # the quantised relative positions must round like the torch expression they replace
    "window_attention.hip": ([f"-DPDF_WA_UE={os.environ['PDFOPS_WA_UE']}"] if os.environ.get("PDFOPS_WA_UE") else []),   # tuning knob (A/B builds)
}


# compiler-flag A/B runs: PDFOPS_EXTRA_FLAGS="fused_layer.hip,fused_layer_mfma.hip:-mllvm -some-flag" (files : flags; ';' between groups)
for _grp in filter(None, os.environ.get("PDFOPS_EXTRA_FLAGS", "").split(";")):
    _files, _, _flags = _grp.partition(":")
    for _f in _files.split(","):
        EXTRA[_f.strip()] = EXTRA.get(_f.strip(), []) + _flags.split()


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


# The geometry TUs again with the squared distance as an explicit FMA chain (csrc/pdfops_common.h: pdf_sqdist3): libpdfops_fma1.so /
# libpdfops_fma2.so = the same library with those five objects swapped.  `PDFOPS_DIST_FMA=1|2` at import selects them (_native.py).
GEOMETRY_TUS = ["knn_query.hip", "knn_grid.hip", "sampling.hip", "sampling_bucketed.hip", "ball_query.hip"]
FMA_VARIANTS = (1, 2)


def variant_path(variant):
    return LIBPATH if not variant else os.path.join(LIBDIR, f"libpdfops_fma{int(variant)}.so")


def build_library(force=False, verbose=False, variants=FMA_VARIANTS):
    """Compile every csrc/*.hip for gfx950 and link libpdfops.so (+ the FMA-distance variants). Returns the default library's path."""
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

    vobjs = {}
    for v in variants:
        vdir = os.path.join(OBJDIR, f"fma{v}")
        os.makedirs(vdir, exist_ok=True)
        vobjs[v] = list(objs)
        for src in GEOMETRY_TUS:
            sp = os.path.join(CSRC, src)
            op = os.path.join(vdir, src.replace(".hip", ".o"))
            vobjs[v][srcs.index(src)] = op
            if force or _stale(op, [sp] + headers):
                jobs.append([hipcc] + COMMON + EXTRA.get(src, []) + [f"-DPDF_DIST_FMA={v}", "-c", sp, "-o", op])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIBPATH, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", LIBPATH])
    for v in variants:
        if force or jobs or _stale(variant_path(v), vobjs[v]):
            run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + vobjs[v] + ["-o", variant_path(v)])
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
