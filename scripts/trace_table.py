"""Per-launch table from a rocprofv3 --kernel-trace CSV: every (kernel, grid, workgroup) with launches per step, average duration and time per step,
plus busy time per HIP stream (queue) -- what a step spends its time on, launch by launch.
usage: python scripts/trace_table.py <dir or *_kernel_trace.csv> <steps in the trace incl. warm-up> [min_us]"""
import collections, csv, glob, os, re, sys

src, steps = sys.argv[1], float(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"unetconv::", "", n)
    return n.split("(")[0][:70]


agg = collections.defaultdict(list)
per_q = collections.defaultdict(float)
t0, t1 = min(int(r["Start_Timestamp"]) for r in rows), max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    agg[(short(r["Kernel_Name"]), g, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))].append(d)
    per_q[r.get("Queue_Id", "?")] += d
tot = sum(sum(v) for v in agg.values())
print(f"{f}\n{len(rows)} launches, {len(rows) / steps:.0f} per step; kernel time {tot / steps / 1e6:.3f} ms per step; span {(t1 - t0) / 1e6:.1f} ms")
for q, d in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: {d / steps / 1e6:.3f} ms per step")
print(f"{'kernel':70s} {'wgs':>7s} {'lds':>6s} {'vgpr':>4s} {'n/step':>7s} {'avg us':>9s} {'ms/step':>8s}")
for (k, g, lds, vg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / steps / 1e3 < min_us:
        continue
    print(f"{k:70s} {g:7d} {lds:>6s} {vg:>4s} {len(v) / steps:7.2f} {sum(v) / len(v) / 1e3:9.1f} {sum(v) / steps / 1e6:8.3f}")
