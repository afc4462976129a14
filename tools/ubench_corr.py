"""micro-benchmark of vv_correlation_nhwc at FlowNetC's geometry ([1,256,56,128] for a 1024x448 pair, [1,256,48,64] for 512x384)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for (H, W) in ((56, 128), (48, 64)):
    f1 = torch.randn(1, H, W, 256, device='cuda'); f2 = torch.randn(1, H, W, 256, device='cuda')
    out = torch.zeros(1, H, W, 476, device='cuda')
    for _ in range(3):
        L.check(lib.vv_correlation_nhwc(f1.data_ptr(), f2.data_ptr(), 256, 1, 256, H, W, out.data_ptr(), 476, 32, 0.1, st), 'corr')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(lib.vv_correlation_nhwc(f1.data_ptr(), f2.data_ptr(), 256, 1, 256, H, W, out.data_ptr(), 476, 32, 0.1, st), 'corr')
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 2.0 * H * W * 441 * 256
    print('correlation %dx%d: %.1f us  %.1f TFLOP/s (%.3f of the fp32 vector peak)' % (H, W, t * 1e6, fl / t / 1e12, fl / t / 157.3e12))
