"""cfg2 step with self-attention on (the reference's shipped default, params_and_main.py:81-83) for rocprofv3 --kernel-trace --stats: python scripts/prof_sa.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
r = B.step_bench("xresnet34", 4, 5, 512, 16, dt, 3, 2, 0, 1, torch.device("cuda", 0), lambda m: None, probe=False, self_attention=True)
print(dt, 16 * 3 / r["dt"])
