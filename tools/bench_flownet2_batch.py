"""FlowNet2 forward throughput vs the number of frame pairs per launch (hipGraph replay, 1024x448, xavier weights)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd.flownet2 import FlowNet2
torch.manual_seed(0)
net = FlowNet2().cuda().eval()
g = torch.Generator().manual_seed(0)
for B in (1, 2, 4, 8):
    x = (torch.rand(B, 3, 2, 448, 1024, generator=g) * 255).cuda()
    for _ in range(2):
        out = net.forward_graphed(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 6
    for _ in range(n):
        out = net.forward_graphed(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(json.dumps({'pairs_per_launch': B, 'ms_per_launch': ms, 'ms_per_pair': ms / B, 'pairs_per_s': 1e3 * B / ms,
                      'tflops': 464.2 * B / ms, 'frac_fp32_mfma_peak': 464.2 * B / ms / 157.3, 'finite': bool(torch.isfinite(out).all())}), flush=True)
    net._graphs.clear()
    del x, out
    torch.cuda.empty_cache()
