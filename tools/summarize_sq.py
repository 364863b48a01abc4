"""Per-kernel SQ counter ratios out of a rocprofv3 --pmc counter_collection CSV (usage: summarize_sq.py CSV OUT.json "COMMAND").
Ratios, summed over every launch of a kernel: MFMA busy / CU busy, and the shares of wave cycles spent waiting, issuing and
issue-stalled; LDS bank-conflict cycles over LDS-active cycles."""
import collections, csv, json, subprocess, sys, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:48]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
out = {"command": sys.argv[3] if len(sys.argv) > 3 else "",
       "git_head": os.environ.get("EP24_GIT_HEAD", ""),          # the GPU box has no .git: the caller passes the commit
       "kernels": {}}
for k, c in acc.items():
    d = lambda a, b: round(c.get(a, 0.0) / c[b], 4) if c.get(b) else None
    out["kernels"][k] = {"launches": len(launches[k]), "mfma_busy_over_cu_busy": d("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"),
                         "wave_cycles_waiting": d("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "wave_cycles_issuing": d("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
                         "wave_cycles_issue_stalled": d("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                         "lds_bank_conflict_over_lds_active": d("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                         "lds_active_over_cu_busy": d("SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CU_CYCLES"),
                         "mfma_busy_share_of_simd_cycles": (round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CU_CYCLES"] / 4, 4)
                                                            if c.get("SQ_BUSY_CU_CYCLES") else None)}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k in sorted(out["kernels"], key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(k, out["kernels"][k])
