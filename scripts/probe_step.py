"""run N whole training steps (dis_update + gen_update) and nothing else, for rocprofv3 PMC passes over ONE step's kernels:
    python scripts/probe_step.py [dtype=fp32] [size=256] [batch=8] [steps=2]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yaml
import aclgan_amd  # noqa
from aclgan_amd.trainer import aclgan_Trainer
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml"))); cfg["display_size"] = 1
torch.manual_seed(0)
tr = aclgan_Trainer(cfg, compute_dtype=dtype)
g = torch.Generator().manual_seed(1)
x_a = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda(); x_b = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(3)]
from aclgan_amd import _lib as L
n0 = L.lib.aclgan_launch_count()
for _ in range(steps):
    tr.dis_update(x_a, x_b, cfg, z=z); tr.gen_update(x_a, x_b, cfg, z=z)
torch.cuda.synchronize()
lps = (L.lib.aclgan_launch_count() - n0) / float(steps)
if os.environ.get("PROBE_STEP_JSON"):
    import json
    json.dump({"launches_per_step": lps, "dtype": dtype, "size": S, "batch": B}, open(os.environ["PROBE_STEP_JSON"], "w"))
print("probe_step: %d steps %s %dx%d B=%d done, %.1f library launches per step" % (steps, dtype, S, S, B, lps))
