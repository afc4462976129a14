"""VERDICT r4 item 2, decided on numbers: would Winograd F(4x4,3x3) with fp32 transforms hold the 1e-3 bars?

Not a test (no test_ prefix): a CPU experiment on the oracle, run in the build container:
    python tests/numerics_wino44.py > profiles/r05_wino44_numerics.txt
The oracle's 3x3 convolution (oracle/unet_oracle.py _conv3 -> F.conv2d) is replaced by a Winograd convolution whose input / filter /
output transforms and element-wise products are all fp32 -- F(2x2,3x3) (what csrc/vv_wino.hip executes) or F(4x4,3x3) -- on the
32x32 level only or on every level; autograd differentiates through the same transforms (the data / weight gradients then carry the
transposed transforms' rounding, like a Winograd gradient kernel would).  The golden recipe of tests/test_oracle_golden.py
(6 cubes, 3 and 6 Adam steps, eval scores) is run in float64 with the plain convolution (the exact trajectory), in float32 with the
plain convolution (the reference's arithmetic) and in float32 with each Winograd variant; reported: max relative deviation of
losses / eval scores from the float64 trajectory, and the single-layer forward error of each form."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_oracle as O

MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                  [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_conv(x, w, b, m):
    """3x3 / stride 1 / pad 1 convolution as F(m x m, 3x3), every step in x.dtype."""
    BT, G, AT = (torch.from_numpy(a).to(x.dtype) for a in MATS[m])
    t = m + 2
    B_, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, t, m).unfold(3, t, m)                       # [B, C, nH, nW, t, t]
    V = torch.einsum('ai,bcyxij,dj->bcyxad', BT, d, BT)          # B^T d B
    U = torch.einsum('ai,kcij,dj->kcad', G, w, G)                # G g G^T
    M = torch.einsum('bcyxad,kcad->bkyxad', V, U)
    Y = torch.einsum('ia,bkyxad,jd->bkyxij', AT, M, AT)          # A^T M A
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, K, H, W)
    return y + b.view(1, -1, 1, 1)


def make_conv(m, only32):
    def conv(x, w, b):
        if m and (not only32 or x.shape[-1] == 32) and x.shape[-1] % m == 0:
            return wino_conv(x, w, b, m)
        return F.conv2d(x, w, b, padding=1)
    return conv


def run(dt, conv, steps):
    O._conv3 = conv
    sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in O.seeded_state_dict('net4', nf=32, padding=False, seed=0).items()}
    raw, flow = O.seeded_cubes(6, 1, 0)
    x, xo = O.cubes_to_inputs(raw, flow)
    x, xo = x.to(dt), xo.to(dt)
    spec = O.bank_spec('net4')
    opt = O.AdamState(O.param_names(sd))
    losses = np.array([O.train_step(sd, spec, x, xo, opt)[:2] for _ in range(steps)])
    rs, os_ = O.score_pass(sd, spec, x, xo, 6)
    return losses, rs.astype(np.float64), os_.astype(np.float64)


def main():
    torch.set_num_threads(8)
    plain = make_conv(0, False)
    g = torch.Generator().manual_seed(0)
    print('single layer, forward (max |err| / max |y| against the float64 direct convolution), x ~ relu(N(0,1)), w ~ U(+-1/sqrt(9 Cin)):')
    for H, Cin, Cout in ((32, 32, 32), (32, 64, 32), (16, 64, 64), (8, 128, 128), (4, 256, 256)):
        x = torch.relu(torch.randn(4, Cin, H, H, generator=g))
        w = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * Cin)
        b = torch.zeros(Cout)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        row = []
        for name, y in (('direct fp32', F.conv2d(x, w, b, padding=1)), ('F(2x2) fp32', wino_conv(x, w, b, 2)), ('F(4x4) fp32', wino_conv(x, w, b, 4))):
            row.append('%s %.1e' % (name, float((y.double() - ref).abs().max() / ref.abs().max())))
        print('  H=%2d %3d->%3d : %s' % (H, Cin, Cout, '   '.join(row)))
    rel = lambda p, q: float((np.abs(p - q) / np.abs(q)).max())
    for steps in (3, 6):
        ref = run(torch.float64, plain, steps)
        print('after %d Adam steps (6 cubes), max relative deviation from the float64 trajectory [losses, raw scores, flow scores]:' % steps)
        for name, conv in (('direct fp32 (the reference arithmetic)', plain), ('F(2x2) fp32 on every level (= csrc/vv_wino.hip)', make_conv(2, False)),
                           ('F(4x4) fp32 on the 32x32 level only', make_conv(4, True)), ('F(4x4) fp32 on every level (4x4 level direct)', make_conv(4, False))):
            r = run(torch.float32, conv, steps)
            print('  %-52s %.2e  %.2e  %.2e' % (name, rel(r[0], ref[0]), rel(r[1], ref[1]), rel(r[2], ref[2])))
    O._conv3 = plain


if __name__ == '__main__':
    main()
