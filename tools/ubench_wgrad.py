"""micro-benchmark of the 3x3 weight-gradient kernels (vv_wgrad_mfma: Winograd form, pad0 bit 8, against the direct form) on the
14 conv layers of a Net4 train step (G=6, B=256): per-launch time, executed TFLOP/s, and the reduced gradients compared.
    python tools/ubench_wgrad.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L
from vec_vad_amd.bank import _pick_ksplit

lib = L.lib()
G, B = 6, int(os.environ.get('UB_B', '256'))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
WFLAG = 256                                            # vv_wgrad_params.pad0 bit 8: Winograd form
NCU = int(os.environ.get('UB_NCU', '768'))             # workgroup slots the k-split heuristic fills (three workgroups per CU)
st = torch.cuda.current_stream().cuda_stream
LAYERS = [(32, 16, 32, 1, 'conv0'), (32, 32, 32, 2, 'conv1/13'), (32, 64, 32, 1, 'conv12'), (16, 32, 64, 1, 'conv2'), (16, 64, 64, 2, 'conv3/11'),
          (16, 128, 64, 1, 'conv10'), (8, 64, 128, 1, 'conv4'), (8, 128, 128, 2, 'conv5/9'), (8, 256, 128, 1, 'conv8'), (4, 128, 256, 1, 'conv6'),
          (4, 256, 256, 1, 'conv7')]
tot = {0: 0.0, WFLAG: 0.0}
for H, Cin, Cout, mult, name in LAYERS:
    g = torch.Generator(device='cpu').manual_seed(H * 1000 + Cin)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    dy = torch.randn(G, B * H * H, Cout, generator=g).cuda()
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    nci, nco = (Cin + 31) // 32, Cout // 32
    nt = lib.vv_wgrad_ntiles(L.CONV3, B, H, H)
    ks = _pick_ksplit(G * nci * nco, nt, ncu=NCU, max_wg=4 * NCU)
    part = torch.zeros(G, nci * nco * ks * 9 * 1024, device='cuda')
    res, times = [], []
    for flag in (0, WFLAG):
        grad = torch.zeros(G, Cout * Cin * 9, device='cuda')
        wp = L.WgradParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                           L.NULL_VIEW, 0, flag, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
        for _ in range(2):
            L.check(lib.vv_wgrad_mfma(C.byref(wp), st), 'wgrad')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.check(lib.vv_wgrad_mfma(C.byref(wp), st), 'wgrad')
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / reps * 1e-3)
        L.check(lib.vv_wgrad_reduce(L.CONV3, G, Cin, Cin, Cout, ks, part.data_ptr(), part.stride(0), grad.data_ptr(), grad.stride(0), st), 'reduce')
        torch.cuda.synchronize()
        res.append(grad.clone())
        tot[flag] += mult * times[-1]
    err = (res[0] - res[1]).abs().max().item() / res[0].abs().max().item()
    alg = 2.0 * B * H * H * 9 * Cin * Cout * G
    print('%-10s H=%2d %3d->%3d x%d ks=%3d : wino %7.1f us %6.1f TF/s exec (%.2f of peak) | direct %7.1f us %6.1f TF/s (%.2f) | err %.1e %s'
          % (name, H, Cin, Cout, mult, ks, times[1] * 1e6, alg * 16 / 36 / times[1] / 1e12, alg * 16 / 36 / times[1] / 157.3e12,
             times[0] * 1e6, alg / times[0] / 1e12, alg / times[0] / 157.3e12, err, '' if err < 2e-5 else '  <-- MISMATCH'), flush=True)
print('weighted (14 launches of a Net4 step): wino %.3f ms, direct %.3f ms' % (tot[WFLAG] * 1e3, tot[0] * 1e3))
