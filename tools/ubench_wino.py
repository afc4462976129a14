"""micro-benchmark of vv_conv_wino on the UNet bank's layer geometries (G=6, B=256): per-launch time, executed / algorithmic TFLOP/s,
and a check against the direct kernel (vv_conv_mfma) on the same tensors.   python tools/ubench_wino.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L

lib = L.lib()
G, B = 6, int(os.environ.get('UB_B', '256'))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
IN_MODE = L.IN_PLAIN if os.environ.get('UB_PLAIN') == '1' else L.IN_ACT
st = torch.cuda.current_stream().cuda_stream
# (H, Cin, Cout, weight in the step: how many forward + dgrad launches of this shape a Net4 train step has)
LAYERS = [(32, 16, 32, 1, 'conv0'), (32, 32, 32, 4, 'conv1/13 dgrad1/13'), (32, 64, 32, 1, 'conv12'), (32, 32, 64, 1, 'dgrad12'),
          (16, 32, 64, 1, 'conv2'), (16, 64, 32, 1, 'dgrad2'), (16, 64, 64, 4, 'conv3/11 dgrad3/11'), (16, 128, 64, 1, 'conv10'), (16, 64, 128, 1, 'dgrad10'),
          (8, 64, 128, 1, 'conv4'), (8, 128, 64, 1, 'dgrad4'), (8, 128, 128, 4, 'conv5/9 dgrad5/9'), (8, 256, 128, 1, 'conv8'), (8, 128, 256, 1, 'dgrad8'),
          (4, 128, 256, 1, 'conv6'), (4, 256, 128, 1, 'dgrad6'), (4, 256, 256, 2, 'conv7 dgrad7')]


if os.environ.get('UB_ONLY32') == '1':
    LAYERS = [l for l in LAYERS if l[0] == 32 and l[1] <= 32]


def pack(fn, w, K, N, taps):
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, 0, K, K, N))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    out = torch.zeros(G, taps * K * N, device='cuda')
    L.check(fn(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), out.data_ptr(), out.stride(0), (9 if taps == 9 else 1) * K * N, st), 'pack')
    return out


tot_t = tot_exec = tot_alg = tot44 = 0.0
for H, Cin, Cout, mult, name in LAYERS:
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    bias = torch.randn(G, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    outs, times = [], []
    pkw = pack(lib.vv_pack_wino, w, Cin, Cout, 16)
    for fn, pk, nt, fl in ((lib.vv_conv_mfma, pack(lib.vv_pack_weights, w, Cin, Cout, 9), lib.vv_conv_ntiles(B, H, H), 0),
                           (lib.vv_conv_wino, pkw, lib.vv_wino_ntiles(B, H), 0),
                           (lib.vv_conv_wino, pkw, lib.vv_wino_ntiles(B, H), L.CONV_NO_RING),
                           (lib.vv_conv_wino44, pack(lib.vv_pack_wino44, w, Cin, Cout, 36), lib.vv_wino44_ntiles(B, H), 0)):
        if fl and not (H == 32 and Cin <= 32):
            outs.append(None); times.append(None)
            continue
        y = torch.zeros(G, B * H * H, Cout, device='cuda')
        s_ = torch.zeros(G, nt, 2, Cout, device='cuda')
        cp = L.ConvParams(L.CONV3, IN_MODE, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                          L.NULL_VIEW, 0, fl, None, pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.view(y, Cout, 0, y.stride(0)),
                          s_.data_ptr())
        for _ in range(2):
            L.check(fn(C.byref(cp), st), 'conv')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.check(fn(C.byref(cp), st), 'conv')
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / reps * 1e-3)
        outs.append((y, s_.sum(1)))
    err = (outs[0][0] - outs[1][0]).abs().max().item() / outs[0][0].abs().max().item()
    serr = (outs[0][1] - outs[1][1]).abs().max().item() / outs[0][1].abs().max().item()
    alg = 2.0 * B * H * H * 9 * Cin * Cout * G
    ring = ''
    e44 = (outs[0][0] - outs[3][0]).abs().max().item() / outs[0][0].abs().max().item()
    s44 = (outs[0][1] - outs[3][1]).abs().max().item() / outs[0][1].abs().max().item()
    w44 = ' | F(4x4) %7.1f us %6.1f TF/s exec (%.2f) alg %6.1f err %.1e stats %.1e' % (
        times[3] * 1e6, alg / 4 / times[3] / 1e12, alg / 4 / times[3] / 157.3e12, alg / times[3] / 1e12, e44, s44)
    tot44 += mult * times[3]
    if outs[2] is not None:
        ring = ' | per-tile kernel %7.1f us, ring bit-equal: %s' % (times[2] * 1e6, bool(torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])))
    print('%-20s H=%2d %3d->%3d x%d : wino %7.1f us %6.1f TF/s exec (%.2f of peak)  alg %6.1f | direct %7.1f us %6.1f TF/s | err %.1e stats %.1e %s'
          % (name, H, Cin, Cout, mult, times[1] * 1e6, alg * 16 / 36 / times[1] / 1e12, alg * 16 / 36 / times[1] / 157.3e12, alg / times[1] / 1e12,
             times[0] * 1e6, alg / times[0] / 1e12, err, serr, ('' if err < 2e-5 and serr < 1e-3 else '  <-- MISMATCH') + ring + w44), flush=True)
    tot_t += mult * times[1]
    tot_alg += mult * alg
print('weighted (27 launches of a Net4 step): %.3f ms, avg %.1f us/launch, executed %.1f TF/s = %.3f of peak, algorithmic %.1f TF/s'
      % (tot_t * 1e3, tot_t / 27 * 1e6, tot_alg * 16 / 36 / tot_t / 1e12, tot_alg * 16 / 36 / tot_t / 157.3e12, tot_alg / tot_t / 1e12))
print('F(4x4,3x3) on every launch: %.3f ms, avg %.1f us/launch' % (tot44 * 1e3, tot44 / 27 * 1e6))
