"""Turn rocprofv3 output (kernel stats CSV, FETCH_SIZE / WRITE_SIZE counter CSVs) into the files kept under profiles/.

usage: summarize_profile.py TAG STATS_CSV STEPS [FETCH_CSV WRITE_CSV]
  STEPS = executions of the step in the profiled process (warm-up + capture warm-up + timed + instrumented)
"""
import collections, csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:48]


def main():
    tag, stats, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rows = list(csv.DictReader(open(stats)))
    out = ["# rocprofv3 --kernel-trace --stats, `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` (%s)" % tag, "",
           "%d executions of the step in the process; per-step = total / %d.  Kernels of the two backward streams overlap," % (steps, steps),
           "so the per-step column sums to more than the wall time per step.", "",
           "| kernel | calls | avg us | ms per step | % |", "|---|---|---|---|---|"]
    tot = 0.0
    for r in rows[:28]:
        ns = float(r["TotalDurationNs"])
        tot += ns
        out.append("| `%s` | %s | %.1f | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, ns / 1e6 / steps,
                                                        float(r["Percentage"])))
    out.append("")
    out.append("sum of all kernel time per step: %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps))
    open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(out) + "\n")
    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "exploration-of-potential_amd", "bench.py"], stdout=subprocess.PIPE, text=True).stdout.strip())
    meta = {"tag": tag, "git_head": head + ("+local changes" if dirty else ""), "steps_in_profiled_process": steps,
            "kernels": {short(r["Name"]): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                                           "ms_per_step": round(float(r["TotalDurationNs"]) / 1e6 / steps, 4)} for r in rows}}
    json.dump(meta, open(os.path.join(ROOT, "profiles", tag + "_kernel_meta.json"), "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 5:
        def agg(path):
            d = collections.defaultdict(lambda: [0, 0.0])
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                d[k][0] += 1
                d[k][1] += float(row["Counter_Value"])
            return d
        f, w = agg(sys.argv[4]), agg(sys.argv[5])
        res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                          "--no-cpu-baseline --no-graph",
               "unit": "bytes per launch (averages over every launch of the kernel in the process)",
               "correction": "FETCH_SIZE (KB) x 1024 x 2 on gfx950 (128-B requests are tallied as 64 B for 16-B-per-lane streaming reads: "
                             "MI355X_MICROARCH.md, HBM); WRITE_SIZE (KB) x 1024 as reported; Infinity-Cache hits are counted",
               "kernels": {}}
        for k in f:
            n = f[k][0]
            fe = f[k][1] / n * 1024 * 2
            wr = w[k][1] / max(w[k][0], 1) * 1024 if k in w else 0.0
            res["kernels"][k] = {"launches": n, "fetch_bytes": round(fe), "write_bytes": round(wr), "traffic_bytes": round(fe + wr)}
        res["git_head"] = meta["git_head"]
        json.dump(res, open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
        # the per-kernel HBM picture: PMC bytes per launch over the kernel's average duration in the graph-replayed run
        st = {short(r["Name"]): r for r in rows}
        tab = ["# HBM traffic per kernel (%s, %s): PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, eager pass) over the average duration" % (tag, meta["git_head"]),
               "# of the same kernel in the graph-replayed run (rocprofv3 --kernel-trace --stats).  Peak HBM3E: 8 000 GB/s.", "",
               "| kernel | launches | avg us | MB per launch | GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|"]
        for k, r in sorted(st.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
            if k in res["kernels"]:
                us = float(r["AverageNs"]) / 1e3
                b = res["kernels"][k]["traffic_bytes"]
                tab.append("| `%s` | %s | %.1f | %.2f | %.0f | %.0f |" % (k, r["Calls"], us, b / 1e6, b / us / 1e3, b / us / 1e3 / 80))
        open(os.path.join(ROOT, "profiles", tag + "_hbm_table.md"), "w").write("\n".join(tab) + "\n")


main()
