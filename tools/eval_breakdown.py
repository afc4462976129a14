"""Per-launch times of the eval-mode scoring forward (test.py:312-345) on the folded model: HIP events around every launch of the plan.
    python tools/eval_breakdown.py [B=512] [reps=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_net, conv_flops
from vec_vad_amd.trainer import FusedTrainer
from vec_vad_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
net, tot_of = build_net('net4', 'fp32', dev)
net.eval()
tr = FusedTrainer(net)
bank = tr.bank
g = torch.Generator().manual_seed(7)
raw = torch.randint(0, 256, (B, 5, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
flow = (torch.randn((B, tot_of, 32, 32, 2), generator=g) * 2.0).to(dev)
ws = bank.set_input_cubes(raw, flow, None, B)
bank.prepare_eval()
plan = ws.fwdq[False]
st = bank._stream()
for _ in range(2):
    plan.run(st)
torch.cuda.synchronize()
per = {}
for _ in range(reps):
    for fn, args, label in plan.calls:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(fn(*args, st), label)
        e1.record()
        per.setdefault(label, []).append((e0, e1))
torch.cuda.synchronize()
fl = conv_flops(bank.lay, B, bank.Ga)
tot = 0.0
rows = []
for k, v in per.items():
    t = sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e-3
    tot += t
    rows.append((t, k))
for t, k in sorted(rows, reverse=True):
    w44 = k.startswith('conv') and k[4:].isdigit() and bank._w44(B, bank.lay.convs[int(k[4:])], False, evalm=True)
    print('%-12s %8.1f us %5.1f%% %s%s' % (k, t * 1e6, 100 * t / tot, ('%.1f TF/s alg' % (fl[k] / t / 1e12)) if k in fl else '', '  [F(4x4)]' if w44 else ''))
print('sum of launches %.3f ms for %d cubes = %.0f cubes/s' % (tot * 1e3, B, B / tot))
