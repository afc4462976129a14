"""BASELINE.json configs[4]: FlowNet2 forward (correlation + conv stacks as HIP kernels) on 1024x436 frame pairs.
436 is not a multiple of 64 (the reference itself fails there), so the pair is zero-padded to
1024x448.  Random xavier weights (no checkpoint offline).  Prints one JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd.flownet2 import FlowNet2

GFLOP_1024x448 = 464.2     # SURVEY.md appendix A.2 (algorithmic, 2 FLOP/MAC)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    H, W = (448, 1024) if len(args) < 2 else (int(args[0]), int(args[1]))
    torch.manual_seed(0)
    net = FlowNet2().cuda().eval()
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(1, 3, 2, H, W, generator=g) * 255).cuda()
    fwd = net.forward_graphed if '--eager' not in sys.argv else net
    for _ in range(3):
        out = fwd(x)
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        out = fwd(x)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    gpu = e0.elapsed_time(e1) / n * 1e-3
    gf = GFLOP_1024x448 * (H * W) / (448 * 1024)
    print(json.dumps({'metric': 'FlowNet2 forward ms/pair', 'H': H, 'W': W, 'ms_per_pair_wall': wall * 1e3, 'ms_per_pair_gpu': gpu * 1e3,
                      'pairs_per_s': 1.0 / wall, 'algorithmic_gflop': gf, 'tflops': gf / wall / 1e3,
                      'frac_fp32_mfma_peak': gf / wall / 1e3 / 157.3, 'finite': bool(torch.isfinite(out).all()), 'mode': 'eager' if '--eager' in sys.argv else 'hipGraph replay'}))


if __name__ == '__main__':
    main()
