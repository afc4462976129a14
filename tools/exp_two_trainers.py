"""Experiment: how much do two INDEPENDENT captured train steps (two whole SelfCompleteNet4 banks) gain from replaying concurrently on
two streams instead of back to back on one?   python tools/exp_two_trainers.py [B] [fp32|bf16]
Measured: fp32 B=256 18.9 -> 17.9 ms per pair (+5.6 %), B=32 4.24 -> 3.23 ms (+31 %), bf16 B=256 8.29 -> 7.05 ms (+17.7 %).
This is the gain of putting MORE UNets in flight at once -- which the grouped launches (one launch = all UNets of the bank) already
do inside one bank: splitting ONE bank's step into two half-bank branches of the graph was built and measured at +-0 (B=256:
9.27 vs 9.24 ms, B=32: 2.14 vs 1.93 ms with the weight-gradient branch, config 4: 10.39 vs 10.24 ms) and removed again."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ['VV_PRECISION'] = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
import torch
from model.unet import SelfCompleteNet4
from vec_vad_amd.trainer import FusedTrainer

g = torch.Generator().manual_seed(0)
raw = torch.randint(0, 256, (1024, 5, 32, 32, 3), dtype=torch.uint8, generator=g).cuda()
flow = (torch.randn(1024, 1, 32, 32, 2, generator=g) * 2).cuda()
idx = torch.arange(B).cuda()


def make():
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False).cuda()
    net.train()
    return FusedTrainer(net)


def run(trs, streams, n):
    for _ in range(n):
        for tr, st in zip(trs, streams):
            with torch.cuda.stream(st):
                tr.step_cubes(raw, flow, idx)


for sched in ('0', 'free'):
    os.environ['VV_GRAPH_OVERLAP'] = sched
    trs = [make(), make()]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for name, streams in (('one stream', (s0, s0)), ('two streams', (s0, s1))):
        run(trs, streams, 4)            # eager, capture, replays (a captured graph replays on whatever stream is current)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(trs, streams, 20)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print('B=%d %s backward schedule %-4s: two steps back to back %.3f ms, concurrently %.3f ms (%.1f %%)'
          % (B, os.environ['VV_PRECISION'], sched, res['one stream'], res['two streams'],
             100 * (res['one stream'] / res['two streams'] - 1)), flush=True)
    del trs
    torch.cuda.empty_cache()
