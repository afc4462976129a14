"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def digest(t, k=8):
    """Same digest as tests/golden/make_goldens.py: [sum, sum|x|, sum x^2, cos-weighted sum, k strided samples]."""
    a = torch.as_tensor(t).detach().double().reshape(-1).cpu()
    idx = torch.linspace(0, a.numel() - 1, k).long()
    w = torch.cos(torch.arange(a.numel(), dtype=torch.float64) * 0.37)
    return np.concatenate([[a.sum().item(), a.abs().sum().item(), (a * a).sum().item(), (a * w).sum().item()],
                           a[idx].numpy()])


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def digest_close(d, g, rtol, atol_scale=1.0):
    """Compare digests: entries 0..3 are sums (tolerance relative to sum|x| / sqrt(sum x^2)), 4.. are samples."""
    d, g = np.asarray(d, dtype=np.float64), np.asarray(g, dtype=np.float64)
    scale_abs = max(g[1], 1e-30)
    ok = abs(d[0] - g[0]) <= rtol * scale_abs * atol_scale
    ok &= abs(d[1] - g[1]) <= rtol * scale_abs
    ok &= abs(d[2] - g[2]) <= 2 * rtol * max(g[2], 1e-30)
    ok &= abs(d[3] - g[3]) <= rtol * scale_abs * atol_scale
    n = max(1.0, float(len(g) - 4))
    rms = np.sqrt(max(g[2], 0.0)) if g[2] > 0 else 0.0
    samp_tol = rtol * (np.abs(g[4:]) + np.abs(g[4:]).max() + 1e-30)
    ok &= bool(np.all(np.abs(d[4:] - g[4:]) <= samp_tol * 4))
    return bool(ok)


def observe(name, **vals):
    """Observed deviations of a parity check: printed (pytest -s / -rP) and appended to gpurun_out/observed.jsonl so that the bars
    in the tests can be set from measurements (bars at ~2x the observed value)."""
    import json
    import os
    rec = dict(test=name, **{k: float(v) for k, v in vals.items()})
    print('OBSERVED', json.dumps(rec))
    try:
        root = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'observed.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
