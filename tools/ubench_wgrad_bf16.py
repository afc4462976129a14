"""micro-benchmark of vv_wgrad_bf16 (mixed-precision weight gradient): time per launch and effective HBM rate per layer geometry
and k-split."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L
import ctypes as C

lib = L.lib()
G, B = 6, 256
dev = 'cuda'


def run(H, Cin, Cout, ks=None, reps=10, dbg=0):
    M = B * H * H
    act = torch.randn(G, M, Cin, device=dev)
    dy = torch.randn(G, M, Cout, device=dev)
    a = torch.rand(G, Cin, device=dev) + 0.5
    b = torch.randn(G, Cin, device=dev) * 0.1
    nci, nco = (Cin + 31) // 32, Cout // 32
    nt, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.vv_wgrad_bf16_plan(0, B, H, H, Cin, Cout, C.byref(nt), C.byref(nblk), C.byref(kw))
    if ks is None:
        ks = max(1, min(nt.value, 256 // (G * nblk.value)))
    part = torch.empty(G, nci * nco * ks * kw.value * 9 * 1024, device=dev)
    wp = L.WgradParams(0, L.IN_ACT, G, B, H, H, Cin, Cin, Cout, ks, L.view(act, Cin, 0, act.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                       L.NULL_VIEW, 0, dbg, None, L.View(dy.data_ptr(), dy.stride(0), Cout, 0), part.data_ptr(), part.stride(0))
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
    by = 4.0 * M * (Cin + Cout) * G
    print('dbg=%d ' % dbg + 'H=%2d Cin=%3d Cout=%3d ks=%3d (tiles %d, wg %d, kw %d): %7.1f us  %5.2f TB/s algorithmic, slabs %.0f MB'
          % (H, Cin, Cout, ks, nt.value, G * nblk.value * ks, kw.value, t * 1e6, by / t / 1e12, part.numel() * 4 / 1e6), flush=True)


for (H, ci, co) in ((32, 16, 32), (32, 32, 32), (32, 64, 32), (16, 32, 64), (16, 64, 64), (16, 128, 64), (8, 64, 128), (8, 128, 128), (8, 256, 128)):
    nblk = max(1, ((ci + 31) // 32) // 2) * max(1, (co // 32) // 2)
    for mult in (1, 2, 4):
        run(H, ci, co, ks=min((256 * mult) // (G * nblk), B * H * H // 256))
