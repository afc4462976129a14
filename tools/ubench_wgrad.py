"""micro-benchmark of vv_wgrad_mfma with bring-up switches (pad0 bits): which part of the kernel costs what."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L
import ctypes as C

lib = L.lib()
G, B = 6, 256
dev = 'cuda'
def run(H, Cin, Cout, dbg, ks=None, reps=10):
    M = B * H * H
    act = torch.randn(G, M, Cin, device=dev)
    dy = torch.randn(G, M, Cout, device=dev)
    a = torch.rand(G, Cin, device=dev) + 0.5
    b = torch.randn(G, Cin, device=dev) * 0.1
    nci, nco = (Cin + 31) // 32, Cout // 32
    nt = lib.vv_wgrad_ntiles(0, B, H, H)
    if ks is None:
        ks = max(1, min(nt, 512 // (G * nci * nco)))
    part = torch.empty(G, nci * nco * ks * 9 * 1024, device=dev)
    wp = L.WgradParams(0, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(act, Cin, 0, act.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                       L.NULL_VIEW, 0, dbg, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.vv_wgrad_mfma(C.byref(wp), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.vv_wgrad_mfma(C.byref(wp), st))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * M * 9 * Cin * Cout * G
    print('H=%2d Cin=%3d Cout=%3d ks=%3d dbg=%d : %7.1f us  %6.1f TF/s' % (H, Cin, Cout, ks, dbg, t * 1e6, fl / t / 1e12), flush=True)

for (H, ci, co) in ((32, 32, 32), (16, 64, 64), (8, 256, 128)):
    for dbg in (0, 1, 2):        # 1 = skip staging (how much of the time is the MFMA loop + epilogue alone)
        run(H, ci, co, dbg)
    for ks in (21, 42, 85, 170):
        if H == 32:
            run(H, ci, co, 0, ks)
