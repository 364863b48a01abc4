"""One replayed training step out of a rocprofv3 kernel-trace CSV as a compact table (usage: trace_step.py TRACE.csv OUT.csv [step]):
start_us,end_us,queue,kernel,grid,wg - relative to the step's first kernel.  The full trace is too large to keep; this is
what tools/ and DESIGN.md analyse (lane overlap, in-step vs isolated durations)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sgd = [i for i, r in enumerate(rows) if "clear_flag_kernel" in r["Kernel_Name"]]     # once per step, behind the last part of the update
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(sgd) - 3
step = rows[sgd[k] + 1:sgd[k + 1] + 1]
t0 = step[0]["s"]
with open(sys.argv[2], "w") as fh:
    fh.write("start_us,end_us,queue,kernel,grid,workgroup\n")
    for r in step:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:48]
        fh.write("%.2f,%.2f,%s,%s,%s,%s\n" % ((r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, r["Queue_Id"], name.replace(",", ";"),
                                             r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
print("step of %d kernels, %.3f ms" % (len(step), (step[-1]["e"] - t0) / 1e6))
