"""120 Adam steps of SelfCompleteNet4 on 256 seeded cubes (64 per step) through the fused engine: losses must stay finite and fall
(raw loss to less than half, flow loss to less than 0.8 of the first step).  A stability check of the kernels under real training
dynamics, longer than the 3- and 6-step oracle comparisons of tests/.   python tools/longtrain_check.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import unet_oracle as O
from model.unet import SelfCompleteNet4
from vec_vad_amd.trainer import FusedTrainer
torch.manual_seed(0)
net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True, padding=False).cuda().train()
tr = FusedTrainer(net)
raw, flow = O.seeded_cubes(256, 1, 3)
rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
g = torch.Generator(device='cpu').manual_seed(1)
hist = []
for s in range(120):
    idx = torch.randperm(256, generator=g)[:64].cuda()
    ws = tr.step_cubes(rawd, flowd, idx)
    if s % 10 == 0 or s == 119:
        lr_, lo_ = tr.losses(ws)
        hist.append((s, float(lr_), float(lo_)))
        print(s, float(lr_), float(lo_), flush=True)
assert all(np.isfinite(h[1]) and np.isfinite(h[2]) for h in hist)
assert hist[-1][1] < 0.5 * hist[0][1] and hist[-1][2] < 0.8 * hist[0][2], hist
print('ok: losses fall and stay finite')
