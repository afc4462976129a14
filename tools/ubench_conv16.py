"""micro-benchmark of the all-bf16 3x3 kernel (vv_conv_bf16.hip) on the 27 conv-family launches of a BASELINE config-4 train step
(SelfCompleteNetFull: G = 10 UNets, B = 512): per-launch time, TFLOP/s, algorithmic GB/s, weighted step sum.
    python tools/ubench_conv16.py [reps] [legacy]         # VV_LIB_PATH=<variant .so> for elimination builds (tools/build_variant.sh)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L

lib = L.lib()
G, B = int(os.environ.get('UB_G', '10')), int(os.environ.get('UB_B', '512'))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
legacy = L.CONV_NO_GEMM16 if len(sys.argv) > 2 and sys.argv[2] == 'legacy' else 0
if os.environ.get('UB_NO_RING') == '1':      # the 32x32 launches on the round-3 kernel instead of conv_ring16_kernel (round 5)
    legacy |= L.CONV_NO_RING
only = os.environ.get('UB_ONLY')
st = torch.cuda.current_stream().cuda_stream
LAYERS = [(32, 16, 32, 'plain', 1, 'conv0'), (32, 32, 32, 'act', 2, 'conv1/13'), (32, 64, 32, 'cat', 1, 'conv12'),
          (32, 32, 32, 'plain', 2, 'dgrad1/13'), (32, 32, 64, 'plain', 1, 'dgrad12'),
          (16, 32, 64, 'plain', 1, 'conv2'), (16, 64, 64, 'act', 2, 'conv3/11'), (16, 128, 64, 'cat', 1, 'conv10'),
          (16, 64, 32, 'plain', 1, 'dgrad2'), (16, 64, 64, 'plain', 2, 'dgrad3/11'), (16, 64, 128, 'plain', 1, 'dgrad10'),
          (8, 64, 128, 'plain', 1, 'conv4'), (8, 128, 128, 'act', 2, 'conv5/9'), (8, 256, 128, 'cat', 1, 'conv8'),
          (8, 128, 64, 'plain', 1, 'dgrad4'), (8, 128, 128, 'plain', 2, 'dgrad5/9'), (8, 128, 256, 'plain', 1, 'dgrad8'),
          (4, 128, 256, 'plain', 1, 'conv6'), (4, 256, 256, 'act', 1, 'conv7'), (4, 256, 128, 'plain', 1, 'dgrad6'), (4, 256, 256, 'plain', 1, 'dgrad7')]


def bf16_buf(shape, gen, scale=1.0):
    """random values stored as bf16 at the start of an fp32-sized [G, n] buffer (the kernels' *_BF16 storage convention)"""
    n = 1
    for s in shape[1:]:
        n *= s
    buf = torch.zeros(shape[0], (n + 1) // 2, device='cuda')
    buf.view(torch.bfloat16)[:, :n] = (torch.randn(shape[0], n, generator=gen, device='cuda') * scale).to(torch.bfloat16)
    return buf


tot_t = tot_f = tot_b = 0.0
nl = 0
gen = torch.Generator(device='cuda').manual_seed(1)
for H, Cin, Cout, mode, mult, name in LAYERS:
    if only and only not in name:
        continue
    if os.environ.get('UB_H') and int(os.environ['UB_H']) != H:
        continue
    csplit = Cin // 2 if mode == 'cat' else Cin
    x0 = bf16_buf((G, B * H * H, csplit), gen)
    x1 = bf16_buf((G, B * H * H, Cin - csplit), gen) if mode == 'cat' else None
    w = torch.randn(G, Cout, Cin, 3, 3, generator=gen, device='cuda') * 0.1
    bias = torch.randn(G, Cout, generator=gen, device='cuda')
    a = torch.rand(G, Cin, generator=gen, device='cuda') + 0.5
    b = torch.randn(G, Cin, generator=gen, device='cuda') * 0.2
    ent = (L.PackEntry * 1)(L.PackEntry(0, 0, L.PACK_BF16, Cin, Cin, Cout))
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).cuda()
    pk = torch.zeros(G, 9 * Cin * Cout, device='cuda')
    L.check(lib.vv_pack_weights(tab.data_ptr(), 1, G, w.data_ptr(), w[0].numel(), pk.data_ptr(), pk.stride(0), 9 * Cin * Cout, st), 'pack')
    flags = L.CONV_BF16 | L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16 | legacy
    nt = lib.vv_conv_ntiles2(B, H, H, L.CONV3, flags)
    y = torch.zeros(G, B * H * H * Cout // 2, device='cuda')
    s_ = torch.zeros(G, nt, 2, Cout, device='cuda')
    in_mode = {'plain': L.IN_PLAIN, 'act': L.IN_ACT, 'cat': L.IN_CAT}[mode]
    cp = L.ConvParams(L.CONV3, in_mode, G, B, H, H, Cin, Cin, Cout, L.View(x0.data_ptr(), x0.stride(0), csplit, 0),
                      a.data_ptr() if mode != 'plain' else None, b.data_ptr() if mode != 'plain' else None, Cin,
                      L.View(x1.data_ptr(), x1.stride(0), Cin - csplit, 0) if x1 is not None else L.NULL_VIEW, csplit, flags, None,
                      pk.data_ptr(), pk.stride(0), bias.data_ptr(), Cout, L.View(y.data_ptr(), y.stride(0), Cout, 0), s_.data_ptr())
    dbg = None
    if os.environ.get('UB_DEBUG'):        # elimination build -DVV_EXPG=512: per-workgroup cycle counters of the two roles
        dbg = torch.zeros(264 * 8, device='cuda')
        cp.bn_partial = dbg.data_ptr()
    for _ in range(2):
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
    torch.cuda.synchronize()
    if dbg is not None:
        d = dbg.view(-1, 8)[:256].cpu().double()
        m = d.mean(0)
        print('   consumer cycles/WG: loop %.0f  epilogue %.0f  barrier %.0f  total %.0f | producer: commit(wait+write) %.0f  stores %.0f  barrier %.0f  issue %.0f'
              % tuple(m.tolist()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.vv_conv_mfma(C.byref(cp), st), 'conv')
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * B * H * H * 9 * Cin * Cout * G
    by = 2.0 * B * H * H * G * (Cin + Cout)
    print('%-10s H=%2d %3d->%3d %-5s x%d : %7.1f us  %7.1f TF/s (%.2f of bf16 peak)  %6.0f GB/s (%.2f of 8 TB/s)'
          % (name, H, Cin, Cout, mode, mult, t * 1e6, fl / t / 1e12, fl / t / 2.5e15, by / t / 1e9, by / t / 8e12), flush=True)
    tot_t += mult * t
    tot_f += mult * fl
    tot_b += mult * by
    nl += mult
    del x0, x1, y, s_, pk, w
    torch.cuda.empty_cache()
print('weighted (%d launches): %.3f ms, avg %.1f us/launch, %.1f TF/s = %.3f of the bf16 MFMA peak, %.0f GB/s = %.3f of 8 TB/s'
      % (nl, tot_t * 1e3, tot_t / max(nl, 1) * 1e6, tot_f / tot_t / 1e12, tot_f / tot_t / 2.5e15, tot_b / tot_t / 1e9, tot_b / tot_t / 8e12))
