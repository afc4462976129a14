"""Golden vectors for the FlowNet2 forward graph from the REAL reference python modules (authoring container only).

The reference's three CUDA ops cannot be built here; they are stubbed with the numpy restatements of
oracle/flow_ops_oracle.py, so this pins the conv / deconv / upsample / concat graph
(flownet2.py, FlowNet{C,S,SD,Fusion}.py, misc.py) -- not the native ops.  Weights are formula-seeded (no checkpoint
offline).  Only outputs are stored."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import flow_ops_oracle as ops  # noqa: E402
from oracle import flownet2_oracle as FO  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import digest  # noqa: E402

sys.modules['png'] = types.ModuleType('png')
import torch.nn.init as I  # noqa: E402
I.uniform = I.uniform_
I.xavier_uniform = I.xavier_uniform_


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.a = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)

    def forward(self, x, y):
        return _t(ops.correlation_fwd(x.detach().numpy(), y.detach().numpy(), *self.a))


class Resample2d(nn.Module):
    def __init__(self, kernel_size=1):
        super().__init__()

    def forward(self, x, f):
        return _t(ops.resample2d_fwd(x.contiguous().detach().numpy(), f.contiguous().detach().numpy()))


class ChannelNorm(nn.Module):
    def __init__(self, norm_deg=2):
        super().__init__()

    def forward(self, x):
        return _t(ops.channelnorm_fwd(x.contiguous().detach().numpy()))


m = types.ModuleType('FlowNet2_src.models.components.ops')
m.Correlation, m.Resample2d, m.ChannelNorm = Correlation, Resample2d, ChannelNorm
sys.modules['FlowNet2_src.models.components.ops'] = m
sys.path.insert(0, '/root/reference')
from FlowNet2_src.models.flownet2 import FlowNet2  # noqa: E402


def main():
    torch.manual_seed(0)
    net = FlowNet2()
    net.eval()
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = FO.seeded_state_dict(shapes, seed=0)
    net.load_state_dict(sd)
    H, W = 128, 192
    rng = np.random.default_rng(42)
    base = rng.uniform(0, 255, (1, 3, 1, H, W)).astype(np.float32)
    # second frame = first frame shifted by (2, 3) px + noise, so that the flow is not degenerate
    second = np.roll(base, (2, 3), axis=(3, 4)) + rng.normal(0, 2, base.shape).astype(np.float32)
    inp = torch.from_numpy(np.clip(np.concatenate([base, second], 2), 0, 255).astype(np.float32))
    with torch.no_grad():
        out = net(inp)
    np.savez_compressed(os.path.join(HERE, 'flownet2_128x192.npz'), out_digest=digest(out), out_shape=np.array(out.shape),
                        out_samples=out.numpy()[0, :, ::16, ::16], param_names=np.array([s[0] for s in shapes]),
                        param_numel=np.array([int(np.prod(s[1])) for s in shapes]))
    print('flownet2 golden written; |flow| max', float(out.abs().max()), 'params', sum(int(np.prod(s[1])) for s in shapes))


if __name__ == '__main__':
    main()
