"""Per-queue busy time / gaps of one training step from a rocprofv3 kernel-trace CSV (usage: trace_gaps.py TRACE.csv [step])."""
import collections, csv, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sgd = [i for i, r in enumerate(rows) if "clear_flag_kernel" in r["Kernel_Name"]]     # once per step, behind the last part of the update
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(sgd) - 3
step = rows[sgd[k] + 1:sgd[k + 1] + 1]
t0, t1 = step[0]["s"], step[-1]["e"]
print("step wall %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, lst in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in lst)
    gaps = [lst[i + 1]["s"] - lst[i]["e"] for i in range(len(lst) - 1)]
    print("queue %s: %4d kernels  busy %.3f ms  gaps %.3f ms (median %.2f us)  span %.3f ms  first +%.3f ms last +%.3f ms" % (
        q, len(lst), busy / 1e6, sum(g for g in gaps if g > 0) / 1e6, statistics.median(gaps) / 1e3 if gaps else 0,
        (lst[-1]["e"] - lst[0]["s"]) / 1e6, (lst[0]["s"] - t0) / 1e6, (lst[-1]["e"] - t0) / 1e6))
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
cur, last, busy = 0, None, 0
for t, d in ev:
    if cur > 0:
        busy += t - last
    cur += d
    last = t
print("union busy %.3f ms, idle %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
main = max(byq.values(), key=len)
big = sorted([(main[i + 1]["s"] - main[i]["e"], main[i]["Kernel_Name"][:44], main[i + 1]["Kernel_Name"][:44]) for i in range(len(main) - 1)],
             reverse=True)[:6]
for g in big:
    print("  gap %.1f us between %s -> %s" % (g[0] / 1e3, g[1], g[2]))
