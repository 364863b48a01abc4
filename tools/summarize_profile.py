"""Turn one tools/collect_profiles.sh output directory into the files kept under profiles/ (run in the build container, after the
gpurun call has merged gpurun_out/prof_TAG back).

usage: summarize_profile.py TAG [PROF_DIR] [STEPS]
  PROF_DIR  default gpurun_out/prof_TAG
  STEPS     executions of the step in the kernel-trace process (warm-up 3 + capture warm-up 1 + timed 10 + instrumented 1 = 15)

Every file is stamped with the commit the profiled tree was pushed from (PROF_DIR/git_head.txt, written on the box by the caller).
The HBM table pairs bytes and durations of ONE launch list: both come from the two eager counter passes themselves (FETCH_SIZE and
WRITE_SIZE need separate passes; their counter CSVs carry the dispatch's start / end time stamps), never from the replayed run.
"""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_CMD = "rocprofv3 --kernel-trace --pmc %s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
SQ = "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:48]


def find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def counter_pass(path):
    """-> {kernel: [launches, counter sum, duration sum ns]} of one single-counter pass."""
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(row["Counter_Value"])
        d[k][2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return d


# device-kernel name prefixes behind the members of bench.py's dominant family (bench.py REPLAY_PREFIX)
REPLAY_PREFIX = {"conv_ring_kernel": "conv_ring_kernel", "conv_ring_generic_kernel": "conv_ring_generic_kernel", "conv_patch_kernel": "conv_patch_kernel", "igemm_dma_kernel": "igemm_dma_",
                 "conv_wreg_kernel": "conv_wreg_kernel"}


def restamp(d, prof, tag, head):
    """The bench line of a collection was printed BEFORE this collection's kernel trace / counter passes existed, so its
    frac_replayed and traffic quote the previous committed files.  Recompute both from THIS collection's files exactly as bench.py
    does (same prefixes, same FLOPs), so that the stored line and the stored profiles are one set."""
    r = d.get("roofline") or {}
    mem = r.get("members")
    meta_p, pmc_p = os.path.join(prof, tag + "_kernel_meta.json"), os.path.join(prof, tag + "_pmc_traffic.json")
    if not mem or not os.path.exists(meta_p):
        return
    meta = json.load(open(meta_p))
    ms = sum(v["ms_per_step"] for k, v in meta["kernels"].items() for m in mem if k.startswith(REPLAY_PREFIX.get(m, m)))
    flops = r["algorithmic_gflop_per_launch"] * r["launches_per_step"] * 1e9
    if ms:
        r["frac_replayed"] = round(flops / (ms * 1e-3) / 1e12 / r["peak"], 4)
        r["frac_replayed_note"] = r["frac_replayed_note"].split("(rocprofv3")[0] + "(rocprofv3 --kernel-trace --stats: %s_kernel_meta.json @ %s, recomputed by tools/summarize_profile.py from this collection)" % (tag, head)
    if os.path.exists(pmc_p):
        pmc = json.load(open(pmc_p))
        n = b = 0
        for name, v in pmc["kernels"].items():
            if any(name.startswith(REPLAY_PREFIX.get(m, m)) for m in mem):
                n += v["launches"]
                b += v["traffic_bytes"] * v["launches"]
        if n:
            r["traffic"] = round(b / n)
            r["traffic_unit"] = "bytes per launch (PMC, %s_pmc_traffic.json @ %s)" % (tag, head)


def main():
    tag = sys.argv[1]
    pdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    prof = os.path.join(ROOT, "profiles")
    head = open(os.path.join(pdir, "git_head.txt")).read().strip() if os.path.exists(os.path.join(pdir, "git_head.txt")) else "unknown"

    # ---- rocprofv3 --kernel-trace --stats of the graph-replayed run
    rows = list(csv.DictReader(open(find(os.path.join(pdir, "kt"), "kernel_stats.csv"))))
    out = ["# rocprofv3 --kernel-trace --stats, `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` (%s @ %s)" % (tag, head), "",
           "%d executions of the step in the process; per-step = total / %d.  Kernels of the two lanes overlap," % (steps, steps),
           "so the per-step column sums to more than the wall time per step.", "",
           "| kernel | calls | avg us | ms per step | % |", "|---|---|---|---|---|"]
    for r in rows[:30]:
        ns = float(r["TotalDurationNs"])
        out.append("| `%s` | %s | %.1f | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, ns / 1e6 / steps, float(r["Percentage"])))
    out += ["", "sum of all kernel time per step: %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps)]
    open(os.path.join(prof, tag + "_summary.md"), "w").write("\n".join(out) + "\n")
    shutil.copy(find(os.path.join(pdir, "kt"), "kernel_stats.csv"), os.path.join(prof, tag + "_bench_kernel_stats.csv"))
    meta = {"tag": tag, "git_head": head, "steps_in_profiled_process": steps,
            "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline",
            "kernels": {short(r["Name"]): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                                           "ms_per_step": round(float(r["TotalDurationNs"]) / 1e6 / steps, 4)} for r in rows}}
    json.dump(meta, open(os.path.join(prof, tag + "_kernel_meta.json"), "w"), indent=1, sort_keys=True)

    # ---- HBM traffic: FETCH_SIZE and WRITE_SIZE passes, bytes and durations of the same eager launch lists
    fcsv, wcsv = find(os.path.join(pdir, "pmc_f"), "counter_collection.csv"), find(os.path.join(pdir, "pmc_w"), "counter_collection.csv")
    if fcsv and wcsv:
        f, w = counter_pass(fcsv), counter_pass(wcsv)
        res = {"command": PMC_CMD % "FETCH_SIZE" + "   and the same with WRITE_SIZE (two separate passes)",
               "unit": "bytes per launch (averages over every launch of the kernel in the process)",
               "correction": "FETCH_SIZE (KB) x 1024 x 2 on gfx950 (128-B requests are tallied as 64 B for 16-B-per-lane streaming reads: "
                             "MI355X_MICROARCH.md, HBM); WRITE_SIZE (KB) x 1024 as reported; Infinity-Cache hits are counted",
               "git_head": head, "kernels": {}}
        for k in f:
            n = f[k][0]
            fe = f[k][1] / n * 1024 * 2
            wr = w[k][1] / max(w[k][0], 1) * 1024 if k in w else 0.0
            us = 0.5 * (f[k][2] / n + (w[k][2] / max(w[k][0], 1) if k in w else f[k][2] / n)) / 1e3
            res["kernels"][k] = {"launches": n, "fetch_bytes": round(fe), "write_bytes": round(wr), "traffic_bytes": round(fe + wr),
                                 "avg_us_in_the_counter_passes": round(us, 2)}
        json.dump(res, open(os.path.join(prof, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
        tab = ["# HBM traffic per kernel (%s @ %s)" % (tag, head), "#",
               "# PMC bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted as traffic) over the kernel's average duration",
               "# IN THE SAME eager counter passes (mean of the FETCH and the WRITE pass: one launch list, launches serial, profiler attached -",
               "# durations read a few per cent longer than un-profiled ones, so the rates are lower bounds).  Peak HBM3E: 8 000 GB/s;",
               "# achievable with a plain copy: ~6 300 GB/s (MI355X_MICROARCH.md).  A kernel whose operands fit the 256 MiB Infinity Cache",
               "# can exceed the HBM rate here without touching HBM; no row may exceed the cache's own rate, and none above 100 % of 8 TB/s is",
               "# evidence of anything but cache residency.", "",
               "| kernel | launches per process | avg us | MB per launch | GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|"]
        order = sorted(res["kernels"].items(), key=lambda kv: -kv[1]["avg_us_in_the_counter_passes"] * kv[1]["launches"])
        for k, v in order:
            us, b = v["avg_us_in_the_counter_passes"], v["traffic_bytes"]
            if us <= 0:
                continue
            tab.append("| `%s` | %d | %.1f | %.2f | %.0f | %.0f |" % (k, v["launches"], us, b / 1e6, b / us / 1e3, b / us / 1e3 / 80))
        open(os.path.join(prof, tag + "_hbm_table.md"), "w").write("\n".join(tab) + "\n")

    # ---- SQ counters (one pass, 8 SQ slots)
    scsv = find(os.path.join(pdir, "sq"), "counter_collection.csv")
    if scsv:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(scsv)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
        sq = {"command": PMC_CMD % SQ, "git_head": head, "kernels": {}}
        for k, c in acc.items():
            d = lambda a, b: round(c.get(a, 0.0) / c[b], 4) if c.get(b) else None
            sq["kernels"][k] = {"launches": len(launches[k]), "mfma_busy_over_cu_busy": d("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"),
                                "wave_cycles_waiting": d("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "wave_cycles_issuing": d("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
                                "wave_cycles_issue_stalled": d("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                                "lds_bank_conflict_over_lds_active": d("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                                "lds_active_over_cu_busy": d("SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CU_CYCLES"),
                                "mfma_busy_share_of_simd_cycles": (round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CU_CYCLES"] / 4, 4)
                                                                   if c.get("SQ_BUSY_CU_CYCLES") else None)}
        json.dump(sq, open(os.path.join(prof, tag + "_pmc_sq.json"), "w"), indent=1, sort_keys=True)

    # ---- the plain records
    for src, dst in (("bench.json", "_bench.json"), ("bench_sustained.json", "_bench_sustained.json"), ("layer_table.txt", "_layer_table.txt"),
                     ("stream_gaps.txt", "_stream_gaps.txt"), ("step_timeline.csv", "_step_timeline.csv")):
        p = os.path.join(pdir, src)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            if src.endswith(".json"):
                lines = [l for l in open(p) if l.startswith("{")]
                if lines:
                    d = json.loads(lines[-1])
                    d["git_head"] = head
                    restamp(d, prof, tag, head)
                    json.dump(d, open(os.path.join(prof, tag + dst), "w"), indent=1)
            elif src.endswith(".txt"):
                open(os.path.join(prof, tag + dst), "w").write("# %s @ %s\n" % (tag, head) + open(p).read())
            else:
                shutil.copy(p, os.path.join(prof, tag + dst))
    print("profiles/%s_* written (commit %s)" % (tag, head))


main()
