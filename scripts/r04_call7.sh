#!/bin/bash
# PMC passes over the 1x1 GEMM kernel vs the generic kernel on the same launches (isolated): where does conv1x1_gemm_kernel lose?
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for on in 1 0; do
  export UNET_CONV1X1_GEMM=$on
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc1x1_sq_$on -- python3 $R/scripts/ab_conv1x1.py f32 > $O/pmc1x1_sq_$on.log 2>&1 || echo "sq pass $on failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc1x1_f_$on -- python3 $R/scripts/ab_conv1x1.py f32 > $O/pmc1x1_f_$on.log 2>&1 || echo "fetch pass $on failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc1x1_w_$on -- python3 $R/scripts/ab_conv1x1.py f32 > $O/pmc1x1_w_$on.log 2>&1 || echo "write pass $on failed"
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc1x1_i_$on -- python3 $R/scripts/ab_conv1x1.py f32 > $O/pmc1x1_i_$on.log 2>&1 || echo "inst pass $on failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
def load(d):
    f=sorted(glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv"))
    if not f: return {}
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        n=r["Kernel_Name"]
        if "conv1x1_gemm" in n: k="gemm1x1"
        elif "conv_igemm16_kernel" in n: k="igemm16 "+n.split("<")[1].split(">")[0]
        else: continue
        out[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
for on in (1,0):
    sq,fe,wr,ii=load(f"pmc1x1_sq_{on}"),load(f"pmc1x1_f_{on}"),load(f"pmc1x1_w_{on}"),load(f"pmc1x1_i_{on}")
    for key in sorted(sq, key=lambda k:int(k[1])):
        s=sq[key]; n=len(s["SQ_BUSY_CU_CYCLES"])
        busy=sum(s["SQ_VALU_MFMA_BUSY_CYCLES"])/max(1,4*sum(s["SQ_BUSY_CU_CYCLES"]))
        waves=sum(s["SQ_WAVE_CYCLES"])/max(1,sum(s["SQ_BUSY_CU_CYCLES"]))
        f=sum(fe.get(key,{}).get("FETCH_SIZE",[0]))/max(1,len(fe.get(key,{}).get("FETCH_SIZE",[0])))*2*1024
        w=sum(wr.get(key,{}).get("WRITE_SIZE",[0]))/max(1,len(wr.get(key,{}).get("WRITE_SIZE",[0])))*1024
        i=ii.get(key,{})
        wt=sum(i.get("SQ_WAIT_INST_ANY",[0]))/max(1,sum(i.get("SQ_ACTIVE_INST_ANY",[1])))
        print(f"on={on} {key[0]:28s} grid {key[1]:>9s} n={n:3d} mfma_busy {busy:.3f} waves/simd {waves:.2f} fetch {f/1e6:8.1f} MB write {w/1e6:8.1f} MB wait/active {wt:.2f}")
PY
