"""micro-benchmark of the all-bf16 3x3 weight gradient (vv_wgrad_bf16 with VV_WGRAD_X_BF16 | VV_WGRAD_DY_BF16: the LDS-ring kernel) on
BASELINE config 4's shapes (SelfCompleteNetFull, B = 512, G = 10): time per launch, algorithmic HBM rate and MFMA rate per layer.
    python tools/ubench_wgrad_ring.py [layer ...]        layers by name: w0 w1 w12 w13 w2 w3 w10 w11 w4 w5 w8 w9 w6 w7"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L
import ctypes as C

lib = L.lib()
G, B = 10, 512
dev = 'cuda'
LAYERS = {'w0': (32, 16, 32), 'w1': (32, 32, 32), 'w12': (32, 64, 32), 'w13': (32, 32, 32), 'w2': (16, 32, 64), 'w3': (16, 64, 64),
          'w10': (16, 128, 64), 'w11': (16, 64, 64), 'w4': (8, 64, 128), 'w5': (8, 128, 128), 'w8': (8, 256, 128), 'w9': (8, 128, 128),
          'w6': (4, 128, 256), 'w7': (4, 256, 256)}


def run(name, H, Cin, Cout, reps=10):
    M = B * H * H
    act = torch.randn(G, M, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(G, M, Cout, device=dev).to(torch.bfloat16)
    a = torch.rand(G, Cin, device=dev) + 0.5
    b = torch.randn(G, Cin, device=dev) * 0.1
    flags = L.WGRAD_DY_BF16 | L.WGRAD_X_BF16
    nci, nco = (Cin + 31) // 32, Cout // 32
    nt, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.vv_wgrad_bf16_plan(L.CONV3 | (flags << 8), B, H, H, Cin, Cout, C.byref(nt), C.byref(nblk), C.byref(kw))
    ks = max(1, min(nt.value, 256 // (G * nblk.value)))
    part = torch.empty(G, nci * nco * ks * kw.value * 9 * 1024, device=dev)
    wp = L.WgradParams(L.CONV3, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(act, Cin, 0, act.stride(0) // 2), a.data_ptr(), b.data_ptr(), Cin,
                       L.NULL_VIEW, 0, flags, None, L.View(dy.data_ptr(), dy.stride(0) // 2, Cout, 0), part.data_ptr(), part.stride(0))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.vv_wgrad_bf16(C.byref(wp), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.vv_wgrad_bf16(C.byref(wp), st))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    by = 2.0 * M * (Cin + Cout) * G
    fl = 2.0 * M * 9 * Cin * Cout * G
    print('%-4s H=%2d %3d->%3d ks=%3d tiles %4d wg %3d: %7.1f us  %5.2f TB/s  %6.1f TF/s' % (name, H, Cin, Cout, ks, nt.value, G * nblk.value * ks, t * 1e6,
                                                                                          by / t / 1e12, fl / t / 1e12), flush=True)


for n in (sys.argv[1:] or list(LAYERS)):
    run(n, *LAYERS[n])
