#!/bin/bash
# cfg1 kernel trace (timestamps per launch, both streams) -> gpurun_out/trace_cfg1
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_cfg1 -- python3 $R/scripts/prof_cfg1.py cfg1 > $R/gpurun_out/trace_cfg1.log 2>&1
echo rc=$?
cd $R
python - <<'PY'
import csv, glob, collections
f=sorted(glob.glob("gpurun_out/trace_cfg1/*/*_kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last step: find adam_kernel launches
ad=[i for i,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a,b=ad[-2]+1, ad[-1]+1
step=rows[a:b]
t0=int(step[0]["Start_Timestamp"]); t1=int(step[-1]["End_Timestamp"])
print("launches", len(step), "step us", (t1-t0)/1e3)
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in step)
print("sum of kernel durations us", busy/1e3)
# union of busy intervals
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in step)
cur_s,cur_e=iv[0]; un=0
for s,e in iv[1:]:
    if s>cur_e: un+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
un+=cur_e-cur_s
print("device busy (union) us", un/1e3, "idle us", (t1-t0-un)/1e3)
q=collections.Counter(r.get("Queue_Id","?") for r in step); print("queues", q)
agg=collections.defaultdict(lambda:[0,0])
for r in step:
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("(anonymous namespace)::","")[:60]
    agg[n][0]+=1; agg[n][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]: print(f"{t/1e3:8.1f} us {c:4d}  {n}")
PY
