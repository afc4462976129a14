"""Diagnostic: per-tensor gradient error of the HIP train step at B=256 against the oracle in fp64, next to the oracle's own
fp32-vs-fp64 spread (demonstrate the tolerance).  Prints a table; run on the GPU box."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import unet_oracle as O
from test_gpu_unet import _build
from test_gpu_fullsize import _grad_table
from vec_vad_amd.trainer import FusedTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.set_num_threads(32)
net, sd, tot_of = _build('net4', False)
raw, flow = O.seeded_cubes(B, tot_of, 17)
x, x_of = O.cubes_to_inputs(raw, flow)
net.train()
tr = FusedTrainer(net)
ws = tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
gh = {k: v.detach().cpu().double() for k, v in _grad_table(net).items()}
res = {}
for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
    sdo = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    opt = O.AdamState(O.param_names(sdo))
    res[tag] = O.train_step(sdo, O.bank_spec('net4'), x.to(dt), x_of.to(dt), opt)[2]
torch.set_num_threads(8)
sdo = {k: v.clone() for k, v in sd.items()}
g8 = O.train_step(sdo, O.bank_spec('net4'), x, x_of, O.AdamState(O.param_names(sdo)))[2]
rows = []
for k, g64 in res['f64'].items():
    if k.endswith('.0.bias') or k.endswith('.3.bias'):
        continue
    n = float(g64.norm())
    rows.append((float((gh[k] - g64).norm()) / n, float((res['f32'][k].double() - g64).norm()) / n,
                 float((g8[k].double() - res['f32'][k].double()).norm()) / n, k))
rows.sort(reverse=True)
print('%-44s %10s %10s %10s' % ('tensor', 'hip-f64', 'orc32-f64', 'orc 8v32thr'))
for r in rows[:25]:
    print('%-44s %10.2e %10.2e %10.2e' % (r[3], r[0], r[1], r[2]))
print('median hip %.2e  orc32 %.2e ; max hip %.2e orc32 %.2e' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows]),
      max(r[0] for r in rows), max(r[1] for r in rows)))
