#!/usr/bin/env python3
"""Per-kernel SQ counter table from rocprofv3 --pmc passes (rocpd sqlite), for tools/prof_sq.sh.

    python tools/rocpd_sq.py results.db             -> one line per (kernel, counter): calls, average per launch, average duration
    python tools/rocpd_sq.py --merge a.txt b.txt .. -> one block per kernel with all counters and the derived fractions
    python tools/rocpd_sq.py --json out.json kt.db steps a.txt b.txt ..  -> profiles/<tag>_sq.json for bench.py's `mfma` entry: per kernel
        the un-profiled average duration and launches per step over the last `steps` optimizer steps of the kernel trace kt.db, the limiter fractions and the
        matrix-pipe busy share, + the per-step totals of the dense-product kernel families (rl2:: rl:: flm:: td::)

Derived (guide: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles; MFMA_BUSY in cycles):
  parked   = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (s_waitcnt / barrier)
  stalled  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     (issue stall: dependency / pipe)
  issuing  = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  occupancy (waves per SIMD while busy) = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES-equivalent): reported as WAVE_CYCLES / BUSY_CYCLES
  mfma busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs: 32 per v_mfma_f32_16x16x4_f32) / (1024 SIMDs x kernel cycles);
              kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
"""
import re
import sqlite3
import sys

HOT = re.compile(r"fl::k_[bp]\d|fl[ms]::k_[bp]\d|rl2::k_|rl::k_|td::k_|sg::k_seg|pw::k_bn|fl::k_colsum|fl::k_bn_finalize|kg::k_grid_query|k_fps_mw|"
                 r"k_dot3|k_step|gather|grouping|interp|agg_|sub_|wb::k_|ln::k_|k_seg_softmax|rg::k_|gp::k_|k_grid_radius")


def short(name):
    name = re.sub(r"\s+", " ", str(name))
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def dump(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
    cname = "counter_name" if "counter_name" in cols else "name"
    val = "value" if "value" in cols else "counter_value"
    dur = {}
    try:
        for name, d in c.execute("select name, (end - start) from kernels"):
            a = dur.setdefault(short(name), [0, 0])
            a[0] += 1; a[1] += d
    except sqlite3.Error:
        pass
    agg = {}
    for k, cn, v in c.execute(f"select {name_col}, {cname}, {val} from counters_collection"):
        k = short(k)
        if not HOT.search(k):
            continue
        a = agg.setdefault((k, cn), [0, 0.0])
        a[0] += 1; a[1] += float(v)
    for (k, cn), a in sorted(agg.items()):
        d = dur.get(k)
        print(f"{k}\t{cn}\t{a[0]}\t{a[1] / a[0]:.1f}\t{(d[1] / d[0] / 1e3) if d else 0:.2f}")


def mfma_busy(c):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    return c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else 0.0


def load(files):
    tab = {}
    for f in files:
        try:
            lines = open(f).read().splitlines()
        except OSError:
            continue
        for ln in lines:
            p = ln.split("\t")
            if len(p) != 5:
                continue
            k, cn, calls, avg, us = p
            e = tab.setdefault(k, dict(calls=int(calls), us=[], c={}))
            e["c"][cn] = float(avg)
            e["us"].append(float(us))
    return tab


def merge(files):
    tab = load(files)
    order = sorted(tab.items(), key=lambda kv: -(sum(kv[1]["us"]) / max(len(kv[1]["us"]), 1)) * kv[1]["calls"])
    for k, e in order:
        c = e["c"]
        us = sum(e["us"]) / max(len(e["us"]), 1)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        line = [f"{k}", f"  calls {e['calls']}  avg {us:.1f} us (serialised pmc run)"]
        if wc:
            line.append("  parked %.2f  issue-stalled %.2f  issuing %.2f  (valu %.2f lds %.2f vmem %.2f sca %.2f of wave cycles)" % (
                c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
                c.get("SQ_ACTIVE_INST_SCA", 0) / wc))
            if c.get("SQ_BUSY_CYCLES"):
                line.append("  waves %.0f  wave-cycles/busy-cycles %.2f" % (c.get("SQ_WAVES", 0), wc / c["SQ_BUSY_CYCLES"]))
        if c.get("SQ_INSTS_VALU"):
            w = max(c.get("SQ_WAVES", 1.0), 1.0)
            line.append("  per wave: valu %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f smem %.0f salu %.0f mfma %.0f" % (
                c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_INSTS_VMEM_RD", 0) / w, c.get("SQ_INSTS_VMEM_WR", 0) / w,
                c.get("SQ_INSTS_SMEM", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_MFMA", 0) / w))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            line.append("  lds bank conflict / lds active %.3f  lds-issue-stall/wave-cycles %.3f" % (
                c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], c.get("SQ_WAIT_INST_LDS", 0) / wc if wc else 0))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            line.append("  mfma busy cycles %.0f  mops f32 %.0f f16 %.0f bf16 %.0f%s" % (
                c["SQ_VALU_MFMA_BUSY_CYCLES"], c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0),
                c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0),
                ("  matrix-pipe busy %.3f of 1024 SIMDs x kernel cycles" % mfma_busy(c)) if gui else ""))
        raw = "  raw: " + " ".join(f"{n}={v:.0f}" for n, v in sorted(c.items()))
        print("\n".join(line))
        print(raw)
        print()


def to_json(out, kt_db, steps, files):
    import json
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    tab = load(files)
    rows = sorted(sqlite3.connect(kt_db).execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
    # per-step figures over a steady-state window delimited by the optimizer kernel (k_sgd: one launch per training step, its last
    # kernel), as tools/rocpd_queues.py does: the last `steps` optimizer steps of the trace (steps <= 0: 12).  Counting launches over the
    # whole trace and dividing by the optimizer launches over-counts (warm-up, capture and measuring passes run no optimizer).
    ends = [r[2] for r in rows if "k_sgd" in str(r[0])]
    w = int(steps) if steps > 0 else 12
    if len(ends) > w:
        t0, t1, steps = ends[-w - 1], ends[-1], float(w)
    else:
        t0, t1, steps = rows[0][1], rows[-1][2], float(max(len(ends), 1))
        print(f"note: {len(ends)} optimizer launches in the trace: whole trace divided by {steps:g}", file=sys.stderr)
    dur = {}
    for name, st, en in rows:
        if st < t0 or en > t1:
            continue
        a = dur.setdefault(short(name), [0, 0])
        a[0] += 1; a[1] += en - st
    kernels, fam = {}, {}
    for k, e in tab.items():
        c = e["c"]
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        d = dur.get(k)
        if not d or not wc:
            continue
        avg_us, per_step = d[1] / d[0] / 1e3, d[0] / float(steps)
        row = dict(avg_us=round(avg_us, 2), launches_per_step=round(per_step, 2), us_per_step=round(avg_us * per_step, 1),
                   parked=round(c.get("SQ_WAIT_ANY", 0) / wc, 3), issue_stalled=round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                   issuing=round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), valu=round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                   lds=round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3), waves=int(c.get("SQ_WAVES", 0)),
                   lds_bank_conflict=round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1.0), 3),
                   mfma_busy=round(mfma_busy(c), 4), mfma_insts_per_launch=int(c.get("SQ_INSTS_MFMA", 0)))
        kernels[k] = row
        f = re.match(r"(?:void )?(rl2|rl|flm|fls|fl|td)::", k)
        if f:
            a = fam.setdefault(f.group(1), dict(us_per_step=0.0, mfma_busy_us_per_step=0.0))
            a["us_per_step"] += avg_us * per_step
            a["mfma_busy_us_per_step"] += avg_us * per_step * mfma_busy(c)
    dense = {k: v for k, v in fam.items()}
    tot = sum(v["us_per_step"] for v in dense.values())
    busy = sum(v["mfma_busy_us_per_step"] for v in dense.values())
    res = dict(kernel_source_hash=bench.kernel_source_hash(), steps_in_kernel_trace=steps,
               note="SQ counters: rocprofv3 --pmc passes of tools/prof_sq.sh (serialised kernels, bench.py --throttle); durations: the "
                    "un-profiled kernel trace of the same state",
               dense_families_us_per_step={k: round(v["us_per_step"], 1) for k, v in dense.items()},
               dense_kernel_ms_per_step=round(tot / 1e3, 3), matrix_pipe_busy_share_of_dense_kernel_time=round(busy / max(tot, 1e-9), 4),
               kernels=dict(sorted(kernels.items(), key=lambda kv: -kv[1]["us_per_step"])[:60]))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("dense_families_us_per_step", "dense_kernel_ms_per_step", "matrix_pipe_busy_share_of_dense_kernel_time")}))


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2:])
    elif sys.argv[1] == "--json":
        to_json(sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5:])
    else:
        dump(sys.argv[1])
