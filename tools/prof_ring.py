"""phase profile of wino_ring_kernel (library built with -DVV_EXPR=128: VV_LIB_PATH=exp_libs/ring_prof.so python tools/prof_ring.py)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L

lib = L.lib()
G, B, H = 6, 256, 32
st = torch.cuda.current_stream().cuda_stream
NAMES = ['vmcnt wait', 'activation', 'chunk barrier', 'DMA issue', 'reads+transform', 'MFMA issue', 'col transform+ex writes',
         'exchange barrier', 'reads+rows+stores', 'statistics', 'whole tile']
for Cin, Cout, mode in ((32, 32, L.IN_PLAIN), (32, 32, L.IN_ACT), (16, 32, L.IN_PLAIN), (32, 64, L.IN_PLAIN)):
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(G, B * H * H, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    nwg = 512
    bias = torch.zeros(G * Cout + 4096 + nwg * 4 * 12 * 2 + 64, device='cuda')
    a = (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    b = (torch.randn(G, Cin, generator=g) * 0.2).cuda()
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, 0, Cin, Cin, Cout))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    pk = torch.zeros(G, 16 * Cin * Cout, device='cuda')
    L.check(lib.vv_pack_wino(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), pk.data_ptr(), pk.stride(0), Cin * Cout, st), 'pack')
    nt = lib.vv_wino_ntiles(B, H)
    y = torch.zeros(G, B * H * H, Cout, device='cuda')
    s_ = torch.zeros(G, nt, 2, Cout, device='cuda')
    cp = L.ConvParams(L.CONV3, mode, G, B, H, H, Cin, Cin, Cout, L.view(x, Cin, 0, x.stride(0)), a.data_ptr(), b.data_ptr(), Cin,
                      L.NULL_VIEW, 0, 0, None, pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.view(y, Cout, 0, y.stride(0)), s_.data_ptr())
    for _ in range(3):
        L.check(lib.vv_conv_wino(C.byref(cp), st), 'conv')
    torch.cuda.synchronize()
    d = bias[4096:4096 + nwg * 4 * 12 * 2].cpu().numpy().view(np.uint64).reshape(nwg, 4, 12).astype(np.float64)
    items = d[:, :, 11]
    ok = items[:, 0] > 0
    per = d[ok][:, :, :11] / items[ok][:, :, None]
    m = per.mean(axis=(0, 1))
    print('--- %d -> %d %s: cycles per tile and wave (mean over %d workgroups x 4 waves; %d tiles per workgroup)'
          % (Cin, Cout, 'act' if mode == L.IN_ACT else 'plain', int(ok.sum()), int(items[ok][0, 0])))
    for n, v in zip(NAMES, m):
        print('  %-26s %8.0f' % (n, v))
    print('  sum of phases %8.0f' % m[:10].sum())
