"""Diagnostic: after 3 / 6 fused train steps on the golden's 6 cubes, HIP eval scores and parameters vs the oracle in fp64, next to the
oracle's own fp32-vs-fp64 spread."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import unet_oracle as O
from test_gpu_unet import _build
from vec_vad_amd.trainer import FusedTrainer

def orc(dt, nthr, kind, tot_of, n, steps, seed=0):
    torch.set_num_threads(nthr)
    sd = O.seeded_state_dict(kind, nf=32, padding=False, seed=0)
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    raw, flow = O.seeded_cubes(n, tot_of, seed)
    x, xo = O.cubes_to_inputs(raw, flow)
    x, xo = x.to(dt), xo.to(dt)
    spec = O.bank_spec(kind)
    opt = O.AdamState(O.param_names(sd))
    losses = [O.train_step(sd, spec, x, xo, opt)[:2] for _ in range(steps)]
    rs, os_ = O.score_pass(sd, spec, x, xo, n)
    return np.array(losses), rs.astype(np.float64), os_.astype(np.float64), sd

for kind, tot_of, n in (('net4', 1, 6), ('full', 5, 4), ('net4', 1, 64)):
    for steps in (3, 6):
        net, sd, _ = _build(kind, False)
        raw, flow = O.seeded_cubes(n, tot_of, 0)
        rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
        net.train()
        tr = FusedTrainer(net)
        ls = []
        for s in range(steps):
            ws = tr.step_cubes(rawd, flowd, torch.arange(n, device='cuda'))
            ls.append([float(v) for v in tr.losses(ws)])
        net.eval()
        r, o = tr.score_cubes(rawd, flowd)
        r, o = r.cpu().numpy().astype(np.float64), o.cpu().numpy().astype(np.float64)
        a = orc(torch.float64, 32, kind, tot_of, n, steps)
        b = orc(torch.float32, 32, kind, tot_of, n, steps)
        c = orc(torch.float32, 4, kind, tot_of, n, steps)
        rel = lambda p, q: float((np.abs(p - q) / np.abs(q)).max())
        print('%s n=%d steps=%d' % (kind, n, steps))
        print('  loss   : hip-f64 %.2e  orc32-f64 %.2e' % (rel(np.array(ls), a[0]), rel(b[0], a[0])))
        print('  raw sc : hip-f64 %.2e  orc32-f64 %.2e  orc32(4thr)-f64 %.2e' % (rel(r, a[1]), rel(b[1], a[1]), rel(c[1], a[1])))
        print('  of sc  : hip-f64 %.2e  orc32-f64 %.2e  orc32(4thr)-f64 %.2e' % (rel(o, a[2]), rel(b[2], a[2]), rel(c[2], a[2])))
        sdn = net.state_dict()
        for tag, other in (('hip', {k: v.cpu().double() for k, v in sdn.items()}), ('orc32', {k: v.double() for k, v in b[3].items()})):
            num = den = 0.0
            for k in O.param_names(a[3]):
                if k.endswith('.0.bias') or k.endswith('.3.bias'):
                    continue
                d = other[k] - a[3][k]
                num += float((d ** 2).sum()); den += float((a[3][k] ** 2).sum())
            print('  params rel L2 %s-f64: %.2e' % (tag, (num / den) ** 0.5))
