#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the three rocprofv3 runs of tools/prof_round.sh (kernel trace, --pmc FETCH_SIZE, --pmc
WRITE_SIZE; rocpd sqlite databases).  bench.py loads the file for `roofline.traffic` and refuses numbers measured on other
kernel sources (`kernel_source_hash`).

    python tools/traffic_json.py <kt.db> <fetch.db> <write.db> <steps in the pmc runs> <out.json>

Per kernel: calls, average duration, HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE is counted in
KB and has to be doubled on gfx950 -- /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section; separate passes per counter).
`host_calls`: HBM bytes per call of the composite host entry points bench.py times with HIP events (one Bottleneck backward =
the kernels that only run in backward passes of the 18 blocks, summed per step / 18; FPS = the sampling kernels per launch sequence).
`dominant_gpu_kernel`: the single kernel with the largest share of GPU time in the kernel trace, with its own roofline figures."""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# composite host calls -> the kernels they launch (regular expressions on demangled names) and calls per step
HOST_CALLS = {
    "bottleneck_backward": (r"fl[ms]?::k_b\d|sg::k_seg_rows|sg::k_seg_weighted<4, true>|rl2::k_wg|pw::k_bn_bwd|fl::k_colsum", 18),   # (seg_weighted<4, false> = interpolation backward, outside the blocks)
    "bottleneck_forward": (r"fl[ms]?::k_p\d|rl2::k_fwd|fl::k_bn_finalize|pw::k_bn_apply|pw::k_bn_stats", 18),
    "farthest_point_sampling": (r"k_fps|fps_plain", None),   # per launch (the grouped pre-pass launches it once per level)
}


def per_kernel(db, col):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
    val = "value" if "value" in cols else "counter_value"
    agg = {}
    for k, v in c.execute(f"select {name_col}, {val} from counters_collection"):
        k = re.sub(r"\s+", " ", str(k))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(v)
    return agg


def main():
    kt, fdb, wdb, steps, out = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
    import bench
    dur = {}
    for name, d in sqlite3.connect(kt).execute("select name, (end - start) from kernels"):
        name = re.sub(r"\s+", " ", name)
        a = dur.setdefault(name, [0, 0])
        a[0] += 1; a[1] += d
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels = {}
    for name in set(fetch) | set(write):
        f, w = fetch.get(name, [0, 0.0]), write.get(name, [0, 0.0])
        calls = max(f[0], w[0])
        kernels[name] = dict(calls_per_step=calls / steps,
                             fetch_bytes_per_launch=2.0 * 1024 * f[1] / max(f[0], 1), write_bytes_per_launch=1024 * w[1] / max(w[0], 1),
                             avg_us=(dur[name][1] / dur[name][0] / 1e3) if name in dur else None)
        kernels[name]["hbm_bytes_per_launch"] = kernels[name]["fetch_bytes_per_launch"] + kernels[name]["write_bytes_per_launch"]
    host = {}
    for call, (pat, per_step) in HOST_CALLS.items():
        sel = {k: v for k, v in kernels.items() if re.search(pat, k)}
        if not sel:
            continue
        if per_step:
            host[call] = sum(v["hbm_bytes_per_launch"] * v["calls_per_step"] for v in sel.values()) / per_step
        else:   # bytes per launch, averaged over the launches of the run
            tot_calls = sum(v["calls_per_step"] for v in sel.values())
            host[call] = sum(v["hbm_bytes_per_launch"] * v["calls_per_step"] for v in sel.values()) / max(tot_calls, 1e-9)
    total = sum(a[1] for a in dur.values()) or 1
    top = max(dur, key=lambda n: dur[n][1])
    dom = dict(kernel=top[:160], share_of_gpu_time=dur[top][1] / total, avg_us=dur[top][1] / dur[top][0] / 1e3, calls=dur[top][0])
    if top in kernels:
        k = kernels[top]
        dom.update(hbm_bytes_per_launch=k["hbm_bytes_per_launch"], hbm_GBps=k["hbm_bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9,
                   frac_of_8TBps=k["hbm_bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 8e12)
    res = dict(kernel_source_hash=bench.kernel_source_hash(), steps_in_pmc_runs=steps, host_calls=host, dominant_gpu_kernel=dom,
               kernels={k[:160]: v for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["calls_per_step"])[:60]})
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(dict(host_calls=host, dominant_gpu_kernel=dom), indent=1))


if __name__ == "__main__":
    main()
