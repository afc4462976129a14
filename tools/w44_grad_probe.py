"""Evidence for the VV_WINO44 default (round 5): every parameter gradient of one 256-cube Net4 train step against the oracle run in
float64, for VV_WINO44 = 0 / 1 / all, beside the distance of the reference's own fp32 arithmetic (the calibration of
tests/test_gpu_fullsize.py::test_net4_b256_train_step_gradients_vs_oracle).   python tools/w44_grad_probe.py   (GPU box, ~2 min)
Measured (profiles/r05_wino44_gradient_probe.txt): median 1.41e-3 / 1.81e-3 / 2.52e-3 against 1.35e-3 for the fp32 oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, './tests')
from oracle import unet_oracle as O
from test_gpu_unet import _build
from test_gpu_fullsize import _grad_table
from vec_vad_amd.trainer import FusedTrainer
torch.set_num_threads(32)
import os
B=256; seed=int(os.environ.get('PROBE_SEED','17'))
raw, flow = O.seeded_cubes(B, 1, seed)
x, x_of = O.cubes_to_inputs(raw, flow)
ref={}
net, sd, tot_of = _build('net4', False)
for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
    sdo = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    opt = O.AdamState(O.param_names(sdo))
    ref[tag] = O.train_step(sdo, O.bank_spec('net4'), x.to(dt), x_of.to(dt), opt)
g32, g64 = ref['f32'][2], ref['f64'][2]
for mode in os.environ.get('PROBE_MODES', '0,1,all').split(';'):
    os.environ['VV_WINO44'] = mode
    net, sd, tot_of = _build('net4', False)
    net.train()
    tr = FusedTrainer(net)
    ws = tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), torch.arange(B, device='cuda'))
    grads = {k: v.detach().cpu().double() for k, v in _grad_table(net).items()}
    eh=[]; er=[]; worst=(0,None)
    for k, g in g64.items():
        if k.endswith('.0.bias') or k.endswith('.3.bias'): continue
        nrm=float(g.norm()); a=float((grads[k]-g).norm())/nrm; b=float((g32[k].double()-g).norm())/nrm
        eh.append(a); er.append(b)
        if a/(3*b+2e-4) > worst[0]: worst=(a/(3*b+2e-4), k, a, b)
    over=[(round(a/(3*b+2e-4),2),k) for k,a,b in zip([k for k in g64 if not (k.endswith('.0.bias') or k.endswith('.3.bias'))],eh,er) if a>0.9*(3*b+2e-4)]
    print('   near/over bar:', over)
    print('VV_WINO44=%s: HIP median %.2e max %.2e | fp32 oracle median %.2e max %.2e | worst ratio to bar %.2f at %s (%.2e vs %.2e)' % (mode, np.median(eh), max(eh), np.median(er), max(er), worst[0], worst[1], worst[2], worst[3]), flush=True)
