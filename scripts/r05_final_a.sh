#!/bin/bash
# end-of-round evidence, part A (final code): rocprofv3 stats + PMC passes of both storage types, the fused SelfAttention launches alone (timing +
# counters), the trace of the bf16 step with SA on
set -o pipefail
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_attention_gpu.py -q -x > $O/r05_zz_sa_tests.log 2>&1 || { tail -5 $O/r05_zz_sa_tests.log; exit 1; }
tail -1 $O/r05_zz_sa_tests.log
bash scripts/profile_round.sh r05_zz f32 || exit 1
bash scripts/profile_round.sh r05_zz_bf16 bf16 || exit 1
OUT=$PWD/gpurun_out/r05_sa_pmc bash scripts/r05_sa_pmc.sh > $O/r05_zz_sa_pmc.log 2>&1; cat $O/r05_sa_pmc/timing.txt
bash scripts/r05_sa_step.sh; cat $O/r05_sa_step/ab.txt
find $O -name "*_kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete
echo part A done
