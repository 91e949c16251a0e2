"""Launch sequence around the SelfAttention block in the LAST step of a rocprofv3 --kernel-trace CSV (start offset, duration, gap to the previous
launch of the same queue).  usage: python scripts/sa_window.py <dir or csv> [before] [after]"""
import csv, glob, os, re, sys
src = sys.argv[1]
before, after = (int(sys.argv[2]) if len(sys.argv) > 2 else 6), (int(sys.argv[3]) if len(sys.argv) > 3 else 12)
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|^void |unetconv::", "", n)
    return n.split("(")[0][:60]


def window(first_pat, last_pat):
    jdx = [i for i, r in enumerate(rows) if last_pat in r["Kernel_Name"]]
    if not jdx:
        return
    i1 = jdx[-1]
    idx = [i for i, r in enumerate(rows[:i1]) if first_pat in r["Kernel_Name"]]          # the last `first` in front of the last `last`
    if not idx:
        return
    i0 = idx[-1]
    t0 = int(rows[max(0, i0 - before)]["Start_Timestamp"])
    last_end = {}
    for r in rows[max(0, i0 - before):i1 + after]:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"  +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  q{q}  {short(r['Kernel_Name'])}")
    print(f"  window {(int(rows[min(len(rows) - 1, i1 + after - 1)]['End_Timestamp']) - t0) / 1e3:.1f} us")


print("forward:")
window("sa_pack_kernel", "sa_fwd_kernel")
print("backward:")
window("sa_rowdot_kernel", "sa_bwd_q_kernel")
