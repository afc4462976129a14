"""Per-layer timing of the FlowNet2 forward (eager, HIP events around every conv / deconv launch): shape, GFLOP, us, TFLOP/s."""
import os, sys
os.environ['VV_FN2_OVERLAP'] = '0'          # one stream: a launch's events bracket that launch alone
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import flownet2 as F2
import torch.nn as nn

rows = []
orig = F2._Runner.__call__


def timed(self, layer, src, dst, dst_coff=0):
    m = layer[0] if isinstance(layer, nn.Sequential) else layer
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(self, layer, src, dst, dst_coff)
    e1.record()
    de = isinstance(m, nn.ConvTranspose2d)
    K, N = (m.in_channels, m.out_channels)
    R = m.kernel_size[0]
    fl = 2.0 * src.B * (src.H * src.W if de else dst.H * dst.W) * R * R * K * N
    rows.append(('%s%dx%d s%d' % ('deconv' if de else 'conv', R, R, m.stride[0]), K, N, src.H, src.W, fl, e0, e1))
    return r


def main():
    H, W = (448, 1024) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
    torch.manual_seed(0)
    net = F2.FlowNet2().cuda().eval()
    x = (torch.rand(1, 3, 2, H, W) * 255).cuda()
    for _ in range(2):
        net(x)
    F2._Runner.__call__ = timed
    net(x)
    torch.cuda.synchronize()
    tot_t = tot_f = 0
    agg = {}
    for (kind, K, N, h, w, fl, e0, e1) in rows:
        t = e0.elapsed_time(e1) * 1e-3
        tot_t += t; tot_f += fl
        key = (kind, K, N, h, w)
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += t; a[2] += fl
    for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-12s %4d->%4d @%3dx%4d  x%d  %7.2f GF  %8.1f us  %6.1f TF/s' % (key + (n, fl / 1e9, t * 1e6, fl / t / 1e12)))
    print('conv launches: %.1f GF in %.1f us = %.1f TF/s' % (tot_f / 1e9, tot_t * 1e6, tot_f / tot_t / 1e12))


if __name__ == '__main__':
    main()
