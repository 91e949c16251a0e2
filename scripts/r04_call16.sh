#!/bin/bash
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sa_$dt -- python3 $R/scripts/prof_sa.py $dt > $O/prof_sa_$dt.log 2>&1 || echo "failed $dt"
done
cd $R
python - <<'PY'
import csv, glob
for dt in ("f32","bf16"):
    f=sorted(glob.glob(f"gpurun_out/prof_sa_{dt}/*/*_kernel_stats.csv"))[-1]
    rows=list(csv.DictReader(open(f)))
    print(dt, "total ms/step", sum(float(r["TotalDurationNs"]) for r in rows)/5e6)
    for r in rows[:40]:
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        print(f'{float(r["TotalDurationNs"])/5e6:8.3f} ms/step {int(r["Calls"])/5:7.1f} calls/step {float(r["AverageNs"])/1e3:9.1f} us  {n}')
PY
