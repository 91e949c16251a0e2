#!/bin/bash
# rebuild libunet_hip.so with the resource-usage remarks, print VGPR/scratch/occupancy of the conv16 kernels, run the ISA check
cd "$(dirname "$0")/.." || exit 1
rm -f unet_amd/lib/resource_usage.log
python -m unet_amd.build --force --verbose > /tmp/build.log 2>&1 || { grep -m5 -A8 "error" /tmp/build.log; exit 1; }
python - <<'PY'
import re
t=open('unet_amd/lib/resource_usage.log').read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name=b.split()[0]
    if 'conv_igemm16' not in name: continue
    g=lambda k: re.search(k+r": (\d+)",b).group(1)
    print(name[30:70], "vgpr",g("VGPRs"),"scratch",g(r"ScratchSize \[bytes/lane\]"),"occ",g(r"Occupancy \[waves/SIMD\]"))
PY
python -m unet_amd.isa_check
