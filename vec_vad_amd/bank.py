"""UNet-bank engine: the 6 / 10 independent completion UNets of VEC_VAD run as ONE grouped sequence of gfx950 kernels.

This is the host side of the hot path.  It owns
  * the flat parameter / gradient / Adam-moment buffers ([G][U] fp32, PyTorch OIHW order inside each UNet block, so a
    reference ``state_dict`` maps 1:1 and Adam is a single fused launch and a single RCCL bucket per UNet),
  * the NHWC activation workspace for a given batch size,
  * a pre-built *launch plan* (ctypes argument blocks created once per batch size) for forward, backward and Adam,
and it calls nothing but the C ABI of libvecvad_hip.so (include/vecvad_hip.h).  No torch.nn op runs on this path.

Reference semantics implemented (paths relative to the reference root):
  model/unet.py:4-70     double_conv / inconv / down / up / outconv
  model/unet.py:172-267  SelfCompleteNet4.forward        model/unet.py:410-556  SelfCompleteNetFull.forward
  train.py:385-402       loss = l_raw*MSE + l_of*MSE, backward, Adam(eps=1e-7)
  train.py:421-426       per-cube squared-error scores (also test.py:330-335)
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import torch

from . import _lib as L

RAW_C, OF_C = 3, 2
HW0 = 32           # cube side (patch_size, config.cfg)


def _ceil(a, b):
    return (a + b - 1) // b * b


def _pick_ksplit(groups, ntiles, ncu=256, epilogue=1.5, max_wg=1024):
    """Pixel-tile split factor of the weight-gradient kernel (one workgroup per CU): minimise
    rounds * (tiles per workgroup + epilogue) where rounds = ceil(groups * ks / #CU)."""
    best, best_cost = 1, None
    for ks in range(1, ntiles + 1):
        if groups * ks > max_wg and ks > 1:
            break
        rounds = -(-groups * ks // ncu)
        cost = rounds * (-(-ntiles // ks) + epilogue)
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = ks, cost
    return best


class UnitSpec:
    """One UNet of the bank: which frame it erases from the input and what it predicts."""

    def __init__(self, role, erase, tgt):
        self.role, self.erase, self.tgt = role, erase, tgt
        self.out_c = RAW_C if role == 'raw' else OF_C


class ConvLayer:
    def __init__(self, idx, H, cin, cout, mode, src, skip=None, up=None):
        self.idx, self.H, self.cin, self.cout, self.mode, self.src, self.skip, self.up = idx, H, cin, cout, mode, src, skip, up
        self.cinp = _ceil(cin, 16)


class BankLayout:
    """Per-UNet block layout (identical for every UNet; the 1x1 output conv is padded to 4 output rows)."""

    def __init__(self, nf, in_ch):
        self.nf, self.in_ch = nf, in_ch
        c = nf
        Ls = [ConvLayer(0, 32, in_ch, nf, L.IN_CUBE, 'cube'), ConvLayer(1, 32, nf, nf, L.IN_ACT, 0)]
        H = 32
        for d in range(3):
            H //= 2
            Ls.append(ConvLayer(len(Ls), H, c, 2 * c, L.IN_POOL, len(Ls) - 1))
            Ls.append(ConvLayer(len(Ls), H, 2 * c, 2 * c, L.IN_ACT, len(Ls) - 1))
            c *= 2
        self.convT = []   # (src conv idx, Hin, cin, cout)
        skips = [5, 3, 1]
        for u in range(3):
            self.convT.append((len(Ls) - 1, H, c, c // 2))
            H *= 2
            Ls.append(ConvLayer(len(Ls), H, c, c // 2, L.IN_CAT, None, skip=skips[u], up=u))
            Ls.append(ConvLayer(len(Ls), H, c // 2, c // 2, L.IN_ACT, len(Ls) - 1))
            c //= 2
        self.convs = Ls
        # ---- parameter block
        off = 0
        self.p = OrderedDict()      # key -> (offset, shape)
        for l in Ls:
            self.p['c%d.w' % l.idx] = (off, (l.cout, l.cin, 3, 3)); off += _ceil(l.cout * l.cin * 9, 4)
            self.p['c%d.b' % l.idx] = (off, (l.cout,)); off += l.cout
            self.p['c%d.g' % l.idx] = (off, (l.cout,)); off += l.cout
            self.p['c%d.beta' % l.idx] = (off, (l.cout,)); off += l.cout
        for u, (_, _, ci, co) in enumerate(self.convT):
            self.p['t%d.w' % u] = (off, (ci, co, 3, 3)); off += ci * co * 9
            self.p['t%d.b' % u] = (off, (co,)); off += co
        self.p['o.w'] = (off, (4, nf, 1, 1)); off += 4 * nf      # rows >= out_c stay zero
        self.p['o.b'] = (off, (4,)); off += 4
        self.U = _ceil(off, 4)
        # ---- buffer block (running statistics)
        off = 0
        self.b = OrderedDict()
        for l in Ls:
            self.b['c%d.rm' % l.idx] = (off, (l.cout,)); off += l.cout
            self.b['c%d.rv' % l.idx] = (off, (l.cout,)); off += l.cout
        self.UB = _ceil(off, 4)
        # ---- packed (MFMA B-operand) block
        off = 0
        self.pk = OrderedDict()     # key -> (offset, mode, K, KP, N, src key)
        for l in Ls:
            self.pk['c%d.f' % l.idx] = (off, 0, l.cin, l.cinp, l.cout, 'c%d.w' % l.idx); off += 9 * l.cinp * l.cout
            if l.idx > 0:
                self.pk['c%d.d' % l.idx] = (off, 1, l.cout, l.cout, l.cin, 'c%d.w' % l.idx); off += 9 * l.cout * l.cin
        for u, (_, _, ci, co) in enumerate(self.convT):
            self.pk['t%d.f' % u] = (off, 2, ci, ci, co, 't%d.w' % u); off += 9 * ci * co
            self.pk['t%d.d' % u] = (off, 3, co, co, ci, 't%d.w' % u); off += 9 * ci * co
        # Winograd F(2x2,3x3) panels of the 3x3 convolutions (16 transformed taps; vv_pack_wino): mode 0 forward, 1 dgrad
        self.pkw = OrderedDict()
        for l in Ls:
            self.pkw['c%d.f' % l.idx] = (off, 0, l.cin, l.cinp, l.cout, 'c%d.w' % l.idx); off += 16 * l.cinp * l.cout
            if l.idx > 0:
                self.pkw['c%d.d' % l.idx] = (off, 1, l.cout, l.cout, l.cin, 'c%d.w' % l.idx); off += 16 * l.cout * l.cin
        # Winograd F(4x4,3x3) panels (36 transformed taps; vv_pack_wino44) -- space for every layer, packed only where a launch plan
        # routes a layer to vv_conv_wino44 (UNetBank._w44)
        self.pkw44 = OrderedDict()
        for l in Ls:
            self.pkw44['c%d.f' % l.idx] = (off, 0, l.cin, l.cinp, l.cout, 'c%d.w' % l.idx); off += 36 * l.cinp * l.cout
            if l.idx > 0:
                self.pkw44['c%d.d' % l.idx] = (off, 1, l.cout, l.cout, l.cin, 'c%d.w' % l.idx); off += 36 * l.cout * l.cin
        self.UP = _ceil(off, 4)
        self.cmax = 8 * nf


def conv_key_to_state_name(stems, key):
    """local key ('c5.w', 't1.b', 'o.w', 'c3.rm' ...) -> reference state_dict name (model/unet.py module tree)."""
    kind, field = key.split('.')
    if kind[0] == 'c':
        l = int(kind[1:])
        if l < 2:
            prefix = stems['inc'] + '.conv.conv'
        elif l < 8:
            prefix = stems['down'][(l - 2) // 2] + '.mpconv.1.conv'
        else:
            prefix = stems['up'][(l - 8) // 2] + '.conv.conv'
        ci, bi = (0, 1) if l % 2 == 0 else (3, 4)
        return {'w': '%s.%d.weight' % (prefix, ci), 'b': '%s.%d.bias' % (prefix, ci),
                'g': '%s.%d.weight' % (prefix, bi), 'beta': '%s.%d.bias' % (prefix, bi),
                'rm': '%s.%d.running_mean' % (prefix, bi), 'rv': '%s.%d.running_var' % (prefix, bi)}[field]
    if kind[0] == 't':
        return stems['up'][int(kind[1:])] + '.up.' + {'w': 'weight', 'b': 'bias'}[field]
    return stems['outc'] + '.conv.' + {'w': 'weight', 'b': 'bias'}[field]


class _Plan:
    def __init__(self):
        self.calls = []     # (fn, args tuple without stream, label) in a valid single-stream order
        self.meta = []      # per call: (stream id 0|1, events to wait for, event to record) for the two-stream executor
        self.keep = []      # ctypes objects that must stay alive

    def add(self, fn, args, label, stream=0, wait=(), record=None, pwait=None):
        """wait / pwait: events this launch waits for in the 'free' / 'paired' two-stream schedules; the tokens '*main' and
        '*side' mean "everything enqueued so far on that stream"."""
        self.calls.append((fn, args, label))
        self.meta.append((stream, tuple(wait), record, tuple(wait if pwait is None else pwait)))

    def run(self, stream):
        for fn, args, label in self.calls:
            rc = fn(*args, stream)
            if rc:
                L.check(rc, label)


class UNetBank:
    def __init__(self, units, nf=32, tot_raw_num=5, tot_of_num=1, padding=False, active=None, device='cuda',
                 lambda_raw=1.0, lambda_of=1.0):
        self.lib = L.lib()
        self.ncu = int(self.lib.vv_num_cus())      # compute units of the device: the launch policies below count rounds of this many workgroups
        self.units = list(units)
        self.G = len(self.units)
        self.nf, self.tot_raw, self.tot_of, self.padding = nf, tot_raw_num, tot_of_num, padding
        if nf % 32:
            raise L.VecVadHipError('features_root must be a multiple of 32 for the MFMA tiles (got %d)' % nf)
        if nf not in (32, 64):
            raise L.VecVadHipError('features_root = %d: nf = 32 (the value of every shipped config.cfg) and nf = 64 (the default of '
                                   'SelfCompleteNet1raw1of, model/unet.py:563) are implemented -- the fused 1x1 output conv + loss '
                                   'kernels are instantiated for these two widths' % nf)
        if nf != 32 and os.environ.get('VV_PRECISION', 'fp32').lower() != 'fp32':
            raise L.VecVadHipError('mixed precision is implemented for features_root = 32 only')
        self.in_ch = RAW_C * (tot_raw_num if padding else tot_raw_num - 1)
        self.lay = BankLayout(nf, self.in_ch)
        self.g0, self.Ga = active if active is not None else (0, self.G)
        self.lambda_raw, self.lambda_of = lambda_raw, lambda_of
        self.device = torch.device(device)
        self._alloc_state()
        self.ws = {}

    @property
    def adam_t(self):
        return self._adam_t

    @adam_t.setter
    def adam_t(self, t):
        self._adam_t = int(t)
        self._adam_t_dev.fill_(int(t))

    # ------------------------------------------------------------------------------------------ persistent state
    def _alloc_state(self):
        d, lay, G = self.device, self.lay, self.G
        self.params = torch.zeros(G, lay.U, device=d)
        self.bufs = torch.zeros(G, lay.UB, device=d)
        # gradients: BUCKET-MAJOR flat buffer.  Bucket k = parameter columns [gb[k], gb[k+1]) of every UNet, stored contiguously as
        # [G][gb[k+1]-gb[k]] at float offset G*gb[k]: the three data-parallel exchanges (decoder / deep encoder / shallow encoder,
        # trainer.GradBuckets) are in-place all-reduces of contiguous ranges -- no staging copies.  Kernels address a gradient
        # tensor through (pointer, per-UNet stride) pairs, so only the stride changes (self._g).
        self.gb = [0, lay.p['c4.w'][0], lay.p['c8.w'][0], lay.U]
        self._gb_c = (C.c_int64 * len(self.gb))(*self.gb)
        self.grads = torch.zeros(G * lay.U, device=d)
        self.adam_m = None
        self.adam_v = None
        # Adam's step counter and bias-correction scalars live on the device (vv_adam_tick), so a captured train step (hipGraph)
        # replays with the right values; the host copy (adam_t) is bookkeeping and is pushed to the device when it is assigned
        self._adam_t = 0
        self._adam_t_dev = torch.zeros(1, dtype=torch.long, device=d)
        self._adam_sc = torch.zeros(2, device=d)
        self.nbt = torch.zeros(G, len(lay.convs), dtype=torch.long, device=d)
        self.packed = torch.zeros(G, lay.UP, device=d)
        cin_p = lay.convs[0].cinp
        chmap = torch.full((G, cin_p), -1, dtype=torch.int32)
        oc = torch.zeros(G, dtype=torch.int32)
        tsrc = torch.zeros(G, dtype=torch.int32)
        tcoff = torch.zeros(G, dtype=torch.int32)
        for g, u in enumerate(self.units):
            if self.padding:      # model/unet.py:179-181: zero the erased frame, keep 15 channels
                for k in range(self.in_ch):
                    chmap[g, k] = -1 if u.erase * RAW_C <= k < (u.erase + 1) * RAW_C else k
            else:                 # model/unet.py:183: drop the erased frame, keep temporal order
                for k in range(self.in_ch):
                    chmap[g, k] = k if k < u.erase * RAW_C else k + RAW_C
            oc[g] = u.out_c
            tsrc[g] = 0 if u.role == 'raw' else 1
            tcoff[g] = u.tgt * (RAW_C if u.role == 'raw' else OF_C)
        self.chmap, self.oc, self.tsrc, self.tcoff = chmap.to(d), oc.to(d), tsrc.to(d), tcoff.to(d)
        # VV_WINOGRAD=0 keeps the direct implicit-GEMM kernel for the 3x3 convolutions (A/B comparisons, bit-for-bit fmaf
        # chains); default: Winograd F(2x2,3x3) for forward and data-gradient (2.25x fewer MFMA cycles, a few ulp apart)
        # VV_PRECISION=bf16 (config.cfg [mi355x] precision): mixed precision, BASELINE config 4 -- the convolutions / transposed
        # convolutions round their operands to bf16 and contract on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (direct kernel;
        # Winograd's transforms would amplify the bf16 rounding); tensors in HBM, BatchNorm, loss, master weights, Adam stay fp32
        self.precision = os.environ.get('VV_PRECISION', 'fp32').lower()
        if self.precision not in ('fp32', 'bf16'):
            raise ValueError("VV_PRECISION / [mi355x] precision must be 'fp32' or 'bf16', got %r" % self.precision)
        self.cflag = L.CONV_BF16 if self.precision == 'bf16' else 0
        if self.cflag and os.environ.get('VV_CONV_GEMM16', '1') == '0':      # A/B switch: the round-3 bf16 3x3 kernel
            self.cflag |= L.CONV_NO_GEMM16
        self.bf16_wgrad = os.environ.get('VV_BF16_WGRAD', '1') != '0'      # 0: keep the fp32 (Winograd) weight gradient in bf16 mode
        # BatchNorm backward stores dy as bf16 when both of its consumers round it to bf16 anyway (bit-identical results)
        self.dz16 = bool(self.cflag) and self.bf16_wgrad and os.environ.get('VV_BF16_DZ', '1') != '0'
        # activation gradients (outputs of the data-gradient kernels and of the output-conv backward) stored as bf16 -- NOT
        # neutral: it is torch.autocast's dtype for them; BatchNorm backward and the transposed conv's gradients read them back
        self.da16 = bool(self.cflag) and self.bf16_wgrad and os.environ.get('VV_BF16_DA', '1') != '0'
        # ... and so are the conv / transposed-conv OUTPUTS (pre-BatchNorm tensors; torch.autocast's output dtype, oracle 'y16') and
        # with them the pooled / frame-erased inputs: every tensor a bf16 kernel stages is then bf16 (VV_CONV_ALLSRC_BF16)
        self.y16 = self.dz16 and self.da16 and os.environ.get('VV_BF16_Y', '1') != '0'
        # concat layers' data gradient as two dense planes (backward_plan.dcat_views): bf16-output launches of vv_conv_mfma only
        self.split_dcat = self.da16 and os.environ.get('VV_SPLIT_DCAT', '1') != '0'        # (da16 => bf16 kernels => no Winograd path)
        self.fflag = self.cflag | ((L.CONV_OUT_BF16 | L.CONV_ALLSRC_BF16) if self.y16 else 0)      # forward launches
        # fused train / scoring steps (no reconstruction store): the 1x1 output conv's forward and backward in ONE pass over y
        # (vv_outconv_fwdbwd); the module API (forward -> outputs -> set_dout -> backward) keeps the two launches
        self.fuse_outconv = os.environ.get('VV_FUSE_OUTCONV', '1') != '0'
        wino_env = os.environ.get('VV_WINOGRAD', '1') != '0'
        self.wino = wino_env and not self.cflag
        self.wino_wgrad = wino_env and os.environ.get('VV_WINOGRAD_WGRAD', '1') != '0'
        # VV_WINO_RING=0 / VV_CONV_RING=0: A/B switch, the 32x32-level launches stay on the per-tile kernels instead of the persistent
        # LDS-DMA ring kernels of round 5 (fp32: wino_ring_kernel, all-bf16: conv_ring16_kernel; bit-identical results either way)
        self.wino_flag = L.CONV_NO_RING if '0' in (os.environ.get('VV_WINO_RING', '1'), os.environ.get('VV_CONV_RING', '1')) else 0
        self.wgrad_flag = 256                  # vv_wgrad_params.pad0 bit 8: Winograd form of the 3x3 weight gradient
        # VV_WINO44 (round 5): which fp32 3x3 launches of a TRAIN step run as Winograd F(4x4,3x3) (vv_conv_wino44: 1.78x fewer matrix-core
        # cycles than F(2x2); with the interpolation points scaled by 3/4 its rounding error is 1 - 3e-6 of the tensor maximum, 2 - 3 x
        # F(2x2)'s).  'dgrad' (default): the DATA-GRADIENT launches the policy of _w44 takes -- a data gradient's rounding feeds no ReLU /
        # max-pool gate and no BatchNorm statistic, and every parameter gradient's distance from float64 is the same to three digits
        # with and without it (tools/w44_grad_probe.py), while the forward pass -- losses, activations, statistics -- stays bit-identical
        # to the F(2x2) path; '1': forward launches too (median gradient distance as good, but a cancellation-heavy bias gradient
        # crosses its calibrated bar on one of the two full-size test batches); 'all': every launch it takes; '0': none;
        # 'set:f3,d10,...': explicit launches (probing)
        self.w44_mode = os.environ.get('VV_WINO44', 'dgrad').lower()
        self._w44_tables = {}
        self._w44_eval_used = set()      # forward panels some eval plan routes to F(4x4): the only ones prepare_eval packs
        # VV_WINO44_EVAL: the same choice for the eval-mode forward on the folded model (test.py:312-345 scoring; no gradients, scores
        # and AUROC judged at 1e-3: a few 1e-6 per layer are noise there).  '1' (default): the policy of _w44; '0': none; 'all'
        self.w44_eval_mode = os.environ.get('VV_WINO44_EVAL', '1').lower()
        # first reduction pass of the BatchNorm backward inside the data-gradient launch that produces dA (where it is the only producer)
        self.fuse_bn_sums = os.environ.get('VV_FUSE_BN_SUMS', '1') != '0'
        direct = [(k, v) for k, v in lay.pk.items() if not (self.wino and k[0] == 'c')]
        ents = (L.PackEntry * len(direct))()
        mx = 0
        for i, (k, (off, mode, K, KP, N, src)) in enumerate(direct):
            # mixed precision: the bf16-operand kernels read bf16 panels (same offsets, half the bytes)
            ents[i] = L.PackEntry(lay.p[src][0], off, mode | (L.PACK_BF16 if self.cflag else 0), K, KP, N)
            mx = max(mx, 9 * KP * N)
        self.pack_table = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(d)
        self.pack_n, self.pack_max = len(direct), mx
        # the panels of the first two conv layers come first in both tables: the forward plan packs them on the main stream and the
        # rest (every later layer's forward panel, all data-gradient and transposed-conv panels) as a parallel branch that joins in
        # front of the third conv (two-stream / captured schedules; a one-stream executor simply runs them in plan order)
        head = lambda k: k[0] == 'c' and int(k[1:k.index('.')]) <= 1
        self.pack_head_n = sum(1 for k, _ in direct if head(k))
        assert all(head(k) for k, _ in direct[:self.pack_head_n])
        wents = (L.PackEntry * len(lay.pkw))()
        wmx = 0
        for i, (k, (off, mode, K, KP, N, src)) in enumerate(lay.pkw.items()):
            wents[i] = L.PackEntry(lay.p[src][0], off, mode, K, KP, N)
            wmx = max(wmx, KP * N)
        self.pack_table_w = torch.frombuffer(bytearray(bytes(wents)), dtype=torch.uint8).to(d)
        self.pack_w_n, self.pack_w_max = len(lay.pkw), wmx
        self.pack_w_head_n = sum(1 for k in lay.pkw if head(k))
        assert all(head(k) for k in list(lay.pkw)[:self.pack_w_head_n])
        # ---- eval mode (test.py:255-257,312-345; train.py:413-427): running-statistics BatchNorm is a constant affine map, so it is
        # folded into the filter + bias of the convolution in front of it ONCE per model state (vv_fold_bn -> params_eval ->
        # packed_eval), the producing conv applies the ReLU (VV_CONV_RELU) and every consumer reads plain values: no per-forward
        # weight packing, no BatchNorm launches, no activation arithmetic on the consumers' load path.  fp32 only: the mixed mode keeps
        # the rounding points its oracle restates.  VV_EVAL_FOLD=0 keeps the train-mode kernel family with running statistics.
        self.eval_fold = (not self.cflag) and os.environ.get('VV_EVAL_FOLD', '1') != '0'
        self.eval_path = ('BatchNorm folded into filter + bias once per model state (vv_fold_bn), ReLU in the conv epilogue, plain loads'
                          if self.eval_fold else 'train-mode kernel family with running statistics')
        self.params_eval = self.packed_eval = None
        self._eval_key = None
        self._state_ver = 0          # bumped by everything that writes parameters / running statistics through raw pointers
        fents = (L.FoldEntry * len(lay.convs))()
        for i, l in enumerate(lay.convs):
            fents[i] = L.FoldEntry(lay.p['c%d.w' % l.idx][0], lay.p['c%d.b' % l.idx][0], lay.p['c%d.g' % l.idx][0],
                                   lay.p['c%d.beta' % l.idx][0], lay.b['c%d.rm' % l.idx][0], lay.b['c%d.rv' % l.idx][0],
                                   l.cout, l.cin * 9)
        self.fold_table = torch.frombuffer(bytearray(bytes(fents)), dtype=torch.uint8).to(d)
        self.ab_ident = torch.stack([torch.ones(G, lay.cmax), torch.zeros(G, lay.cmax)]).to(d)      # a = 1, b = 0: relu(1*y + 0) = y for y >= 0

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        for n in ('params', 'bufs', 'grads', '_adam_t_dev', '_adam_sc', 'nbt', 'packed', 'chmap', 'oc', 'tsrc', 'tcoff', 'pack_table', 'pack_table_w', 'fold_table', 'ab_ident'):
            setattr(self, n, getattr(self, n).to(device))
        self._w44_tables, self._w44_eval_table = {}, None
        if self.adam_m is not None:
            self.adam_m, self.adam_v = self.adam_m.to(device), self.adam_v.to(device)
        self.device = device
        self.ws = {}
        self.params_eval = self.packed_eval = None
        self._eval_key = None
        return self

    def param_view(self, g, key):
        off, shape = self.lay.p[key]
        n = 1
        for s in shape:
            n *= s
        return self.params[g, off:off + n].view(shape)

    def _bucket_of(self, off):
        for k in range(len(self.gb) - 1):
            if self.gb[k] <= off < self.gb[k + 1]:
                return k
        raise KeyError(off)

    def grad_offset(self, g, key):
        """float offset of the gradient of parameter ``key`` of UNet ``g`` inside the bucket-major buffer."""
        off = self.lay.p[key][0]
        k = self._bucket_of(off)
        lo, w = self.gb[k], self.gb[k + 1] - self.gb[k]
        return self.G * lo + g * w + (off - lo)

    def _g(self, key):
        """(device pointer, per-UNet stride in floats) of the gradient of ``key`` for the active window's first UNet."""
        k = self._bucket_of(self.lay.p[key][0])
        return self.grads.data_ptr() + 4 * self.grad_offset(self.g0, key), self.gb[k + 1] - self.gb[k]

    def grad_view(self, g, key, grads=None, shape=None):
        """gradient of parameter ``key`` of UNet ``g``; ``shape``: the module parameter's own shape where the bank pads (the 1x1 output
        conv has 4 rows in the bank, 3 / 2 in the module)."""
        off, lshape = self.lay.p[key]
        shape = lshape if shape is None else tuple(shape)
        n = 1
        for s in shape:
            n *= s
        o = self.grad_offset(g, key)
        return (self.grads if grads is None else grads).view(-1)[o:o + n].view(shape)

    def grads_gu(self, grads=None):
        """the gradient buffer re-assembled as [G][U] (a copy; tests / diagnostics)."""
        src = (self.grads if grads is None else grads).view(-1)
        out = torch.empty(self.G, self.lay.U, device=src.device, dtype=src.dtype)
        for k in range(len(self.gb) - 1):
            lo, hi = self.gb[k], self.gb[k + 1]
            out[:, lo:hi] = src[self.G * lo:self.G * hi].view(self.G, hi - lo)
        return out

    def buf_view(self, g, key):
        off, shape = self.lay.b[key]
        return self.bufs[g, off:off + shape[0]]

    # ------------------------------------------------------------------------------------------ workspace + plans
    def _p(self, t, off=0):
        return t.data_ptr() + 4 * off

    def workspace(self, B):
        ws = self.ws.get(B)
        if ws is None:
            ws = self._build_ws(B)
            if len(self.ws) > 3:
                self.ws = {}
            self.ws[B] = ws
        return ws

    def _build_ws(self, B):
        lib, lay, d, Ga = self.lib, self.lay, self.device, self.Ga
        nf = self.nf
        f = lambda *s: torch.empty(*s, device=d, dtype=torch.float32)
        ws = type('WS', (), {})()
        ws.B = B
        HWp = HW0 * HW0
        ws.cube = f(B, HWp, RAW_C * self.tot_raw)
        ws.flow = f(B, HWp, OF_C * self.tot_of)
        ws.y = [f(Ga, B * l.H * l.H, l.cout) for l in lay.convs]
        ws.t = [f(Ga, B * (2 * H) * (2 * H), co) for (_, H, ci, co) in lay.convT]
        ws.pooled = {l.idx: f(Ga, B * l.H * l.H, l.cin) for l in lay.convs if l.mode == L.IN_POOL}
        ws.erased = f(Ga, B * HWp, lay.convs[0].cinp)
        nt = [max(lib.vv_conv_ntiles2(B, l.H, l.H, L.CONV3, self.cflag), lib.vv_conv_ntiles2(B, l.H, l.H, L.CONV3, self.fflag),
                  lib.vv_wino_ntiles(B, l.H)) for l in lay.convs]
        ws.stats = f(Ga, max(n * 2 * l.cout for n, l in zip(nt, lay.convs)))
        ws.ab = torch.zeros(4, len(lay.convs), Ga, lay.cmax, device=d)
        ws.out4 = f(Ga, B * HWp, 4)
        ws.score = f(Ga, B)
        ws.dout4 = torch.zeros(Ga, B * HWp, 4, device=d)
        ws.gscale = torch.zeros(Ga, device=d)
        n_raw = sum(1 for u in self.units[self.g0:self.g0 + Ga] if u.role == 'raw')
        n_of = Ga - n_raw
        gs = []
        # train.py:385-392: loss = lambda_raw * L_raw + lambda_of * L_of with flow, plain L_raw (no lambda) without
        lam_raw = self.lambda_raw if n_of else 1.0
        for u in self.units[self.g0:self.g0 + Ga]:
            if u.role == 'raw':
                gs.append(2.0 * lam_raw / (B * n_raw * RAW_C * HWp))
            else:
                gs.append(2.0 * self.lambda_of / (B * n_of * OF_C * HWp))
        ws.gscale.copy_(torch.tensor(gs))
        ws.n_raw, ws.n_of = n_raw, n_of
        if self.fuse_outconv:          # written by the forward plan's vv_outconv_fwdbwd, read by the backward plan
            self._alloc_outconv_bwd(ws, B)
        ws.fwd = {True: self._plan_forward(ws, B, True),
                  False: self._plan_eval(ws, B) if self.eval_fold else self._plan_forward(ws, B, False)}
        # the same plans without the reconstruction store (FusedTrainer's train step and scoring pass only read the per-cube scores
        # and d(loss)/d(out); the module API -- forward() -> outputs_nchw() -- uses ws.fwd)
        ws.fwdq = {True: self._plan_forward(ws, B, True, out4=False),
                   False: self._plan_eval(ws, B, out4=False) if self.eval_fold else self._plan_forward(ws, B, False, out4=False)}
        ws.bwd = None
        ws.bwdq = None                 # the backward plan without its vv_outconv_bwd launch (behind a fused forward)
        ws.out4_valid = False          # True while ws.out4 holds the reconstructions of the LAST forward on this workspace
        return ws

    def _ks_bf16(self, groups, ntiles):
        """k-split of a bf16 weight-gradient launch (one workgroup per CU): the largest split that keeps the launch in ONE round of 256
        workgroups -- unless one round leaves more than a third of the chip idle (the 160-workgroup launches of the 4x4 level and the
        first transposed conv at G = 10): then up to VV_WGRAD_BF16_ROUNDS (default 2) rounds, choosing the split with the smallest
        rounds x ceil(tiles / split) + 1."""
        ncu = self.ncu
        one = max(1, min(ntiles, ncu // groups))
        rmax = int(os.environ.get('VV_WGRAD_BF16_ROUNDS', '2'))
        if groups * one * 3 > ncu * 2 or rmax <= 1:
            return one
        best, cost = one, -(-ntiles // one) + 1
        for ks in range(one + 1, min(ntiles, ncu * rmax // groups) + 1):
            c = -(-groups * ks // ncu) * (-(-ntiles // ks) + 1)
            if c < cost:
                best, cost = ks, c
        return best

    def _w44(self, B, l, dgrad, evalm=False):
        """Does the forward (dgrad=False) / data-gradient launch of conv layer l run as Winograd F(4x4,3x3) at batch size B (evalm: in
        the eval-mode plan)?  Policy
        from per-launch measurements (tools/ubench_wino.py, profiles/README.md): the kernel wins where the matrix pipe is the bound
        -- GEMM-K (input channels of the launch) >= 64 -- AND its coarser workgroups (64 tiles of 4x4 pixels x 32 channels, one per CU)
        still fill the chip evenly: at least two rounds of 256 workgroups with <= 10 % of the last round idle."""
        mode = self.w44_eval_mode if evalm else self.w44_mode
        if not self.wino or mode in ('0', 'off') or (dgrad and l.idx == 0):
            return False
        if mode.startswith('set:'):                      # explicit launches, e.g. VV_WINO44=set:f3,f10,d10 (probing a policy)
            return ('%s%d' % ('d' if dgrad else 'f', l.idx)) in mode[4:].split(',')
        if mode == 'dgrad' and not dgrad:
            return False
        K, N = (l.cout, l.cin) if dgrad else (l.cinp, l.cout)
        if N % 32 or K % 8 or K > 256:
            return False
        if mode == 'all':
            return True
        wgs = self.Ga * (N // 32) * -(-self.lib.vv_wino44_ntiles(B, l.H) // 2)
        rounds = wgs / float(self.ncu)
        return K >= 64 and wgs >= 2 * self.ncu and rounds / math.ceil(rounds) >= 0.9

    def _w44_pack(self, B):
        """device pack table (vv_pack_wino44) of the panels the plans for batch size B use; (tensor, entries, max K*N) or None"""
        if B not in self._w44_tables:
            lay = self.lay
            keys = [('c%d.f' % l.idx) for l in lay.convs if self._w44(B, l, False)] + [('c%d.d' % l.idx) for l in lay.convs if self._w44(B, l, True)]
            if not keys:
                self._w44_tables[B] = None
            else:
                ents = (L.PackEntry * len(keys))()
                mx = 0
                for i, k in enumerate(keys):
                    off, mode, K, KP, N, src = lay.pkw44[k]
                    ents[i] = L.PackEntry(lay.p[src][0], off, mode, K, KP, N)
                    mx = max(mx, KP * N)
                self._w44_tables[B] = (torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(self.device), len(keys), mx)
        return self._w44_tables[B]

    def _src_for(self, ws, l):
        """(load mode, src0 view, a, b, src1 view, csplit, chmap) of conv layer l's input."""
        lay, Ga = self.lay, self.Ga
        abg = lay.cmax
        if l.mode == L.IN_CUBE:       # frame-erased input, materialised once per step by vv_cube_erase
            return (L.IN_PLAIN, L.view(ws.erased, l.cinp, 0, ws.erased.stride(0)), None, None, L.NULL_VIEW, 0, None)
        if l.mode == L.IN_POOL:       # 2x2 max-pooled activation, materialised by vv_pool_act right after its BatchNorm
            pb = ws.pooled[l.idx]
            return (L.IN_PLAIN, L.view(pb, l.cin, 0, pb.stride(0)), None, None, L.NULL_VIEW, 0, None)
        if l.mode == L.IN_ACT:
            s = lay.convs[l.src]
            y = ws.y[s.idx]
            return (L.IN_ACT, L.view(y, s.cout, 0, y.stride(0)), self._p(ws.ab[0, s.idx]), self._p(ws.ab[1, s.idx]), L.NULL_VIEW, 0, None)
        s = lay.convs[l.skip]
        y, t = ws.y[s.idx], ws.t[l.up]
        return (L.IN_CAT, L.view(y, s.cout, 0, y.stride(0)), self._p(ws.ab[0, s.idx]), self._p(ws.ab[1, s.idx]),
                L.view(t, t.shape[2], 0, t.stride(0)), s.cout, None)

    def _alloc_outconv_bwd(self, ws, B):
        if hasattr(ws, 'gA_last'):
            return
        lib, lay, Ga = self.lib, self.lay, self.Ga
        d = self.device
        f = lambda *shape: torch.empty(*shape, device=d, dtype=torch.float32)
        HWp = HW0 * HW0
        ws.gA_last = f(Ga, B * HWp, self.nf)
        nblk = [lib.vv_bn_bwd_nblk(B, l.H, l.H, l.cout) for l in lay.convs]
        # rows: BatchNorm-backward blocks, or the pixel tiles of whichever conv kernel leaves the fused sums (Winograd tiles in fp32,
        # vv_conv_mfma's tiles for the bf16 32x32 launches: VV_BNBWD_PARTIALS_PER_TILE / _PER_CTILE) -- sized for all of them explicitly
        ws.bnpart = f(Ga, max(max(n, lib.vv_wino_ntiles(B, l.H), lib.vv_conv_ntiles(B, l.H, l.H),
                                  lib.vv_conv_ntiles2(B, l.H, l.H, L.CONV3, self.fflag),
                                  lib.vv_convt_dgrad_ntiles(B, l.H, l.H, 0))
                              * 2 * l.cout for n, l in zip(nblk, lay.convs)))
        ws.ocpart = f(Ga, B, 4 * self.nf + 4)

    def backward_plan(self, ws, fused):
        """The backward plan; fused = behind a forward that ran vv_outconv_fwdbwd (its vv_outconv_bwd launch is left out)."""
        if ws.bwd is None:
            ws.bwd = self._plan_backward(ws, ws.B)
        if not fused:
            return ws.bwd
        if ws.bwdq is None:
            q = _Plan()
            q.keep = ws.bwd.keep
            for c, m in zip(ws.bwd.calls, ws.bwd.meta):
                if c[2] != 'outconv_bwd':
                    q.calls.append(c)
                    q.meta.append(m)
            ws.bwdq = q
        return ws.bwdq

    def _plan_forward(self, ws, B, train, out4=True):
        lib, lay, Ga, g0 = self.lib, self.lay, self.Ga, self.g0
        U, UB, UP = lay.U, lay.UB, lay.UP
        P = _Plan()
        pbase, bbase, kbase = self._p(self.params, g0 * U), self._p(self.bufs, g0 * UB), self._p(self.packed, g0 * UP)
        es = C.sizeof(L.PackEntry)
        hn, whn = self.pack_head_n, self.pack_w_head_n
        pack_tail = []          # events the third conv layer waits for (the tail packs run on the side stream)
        ts = 0 if os.environ.get('VV_PACK_TAIL_MAIN', '0') == '1' else 1      # A/B switch: 1 = everything on the main stream
        if hn:
            P.add(lib.vv_pack_weights, (self.pack_table.data_ptr(), hn, Ga, pbase, U, kbase, UP, self.pack_max), 'pack')
        if self.wino and whn:
            P.add(lib.vv_pack_wino, (self.pack_table_w.data_ptr(), whn, Ga, pbase, U, kbase, UP, self.pack_w_max), 'pack_wino')
        if self.pack_n > hn:
            P.add(lib.vv_pack_weights, (self.pack_table.data_ptr() + hn * es, self.pack_n - hn, Ga, pbase, U, kbase, UP, self.pack_max),
                  'pack_tail', stream=ts, record='pack_tail')
            pack_tail.append('pack_tail')
        if self.wino and self.pack_w_n > whn:
            P.add(lib.vv_pack_wino, (self.pack_table_w.data_ptr() + whn * es, self.pack_w_n - whn, Ga, pbase, U, kbase, UP,
                                     self.pack_w_max), 'pack_wino_tail', stream=ts, record='pack_wino_tail')
            pack_tail.append('pack_wino_tail')
        w44p = self._w44_pack(B) if train else None
        if w44p:
            P.add(lib.vv_pack_wino44, (w44p[0].data_ptr(), w44p[1], Ga, pbase, U, kbase, UP, w44p[2]), 'pack_wino44', stream=ts,
                  record='pack_wino44')
            pack_tail.append('pack_wino44')
        abg = lay.cmax

        def conv(l):
            mode, s0, a, b, s1, csplit, chmap = self._src_for(ws, l)
            y = ws.y[l.idx]
            w44 = train and self._w44(B, l, False)
            panel = (lay.pkw44 if w44 else lay.pkw if self.wino else lay.pk)['c%d.f' % l.idx][0]
            cp = L.ConvParams(L.CONV3, mode, Ga, B, l.H, l.H, l.cin, l.cinp, l.cout, s0, a, b, abg, s1, csplit, self.fflag | self.wino_flag, chmap,
                              kbase + 4 * panel, UP, pbase + 4 * lay.p['c%d.b' % l.idx][0], U,
                              L.view(y, l.cout, 0, y.stride(0)), ws.stats.data_ptr() if train else None)
            P.keep.append(cp)
            P.add(lib.vv_conv_wino44 if w44 else lib.vv_conv_wino if self.wino else lib.vv_conv_mfma, (C.byref(cp),), 'conv%d' % l.idx,
                  # the third conv joins the side-stream packs; a launch of the FIRST TWO layers routed to F(4x4) (VV_WINO44=all | set:f0,f1)
                  # reads a panel of that side-stream pack too and waits for it on its own (under the two-stream schedules it
                  # read stale panels)
                  wait=tuple(pack_tail) if l.idx == 2 else (('pack_wino44',) if (w44 and l.idx < 2) else ()))
            nt = lib.vv_wino44_ntiles(B, l.H) if w44 else lib.vv_wino_ntiles(B, l.H) if self.wino else \
                lib.vv_conv_ntiles2(B, l.H, l.H, L.CONV3, self.fflag)
            P.add(lib.vv_bn_finalize,
                  (Ga, l.cout, nt, B * l.H * l.H, 1 if train else 0, 0.1, 1e-5, ws.stats.data_ptr(), nt * 2 * l.cout,
                   pbase + 4 * lay.p['c%d.g' % l.idx][0], pbase + 4 * lay.p['c%d.beta' % l.idx][0], U,
                   bbase + 4 * lay.b['c%d.rm' % l.idx][0], bbase + 4 * lay.b['c%d.rv' % l.idx][0], UB,
                   self._p(ws.ab[0, l.idx]), self._p(ws.ab[1, l.idx]), self._p(ws.ab[2, l.idx]), self._p(ws.ab[3, l.idx]), abg),
                  'bn%d' % l.idx)

        def convT(u):
            sidx, H, ci, co = lay.convT[u]
            y, t = ws.y[sidx], ws.t[u]
            cp = L.ConvParams(L.CONVT_FWD, L.IN_ACT, Ga, B, H, H, ci, ci, co, L.view(y, ci, 0, y.stride(0)),
                              self._p(ws.ab[0, sidx]), self._p(ws.ab[1, sidx]), abg, L.NULL_VIEW, 0, self.fflag, None,
                              kbase + 4 * lay.pk['t%d.f' % u][0], UP, pbase + 4 * lay.p['t%d.b' % u][0], U,
                              L.view(t, co, 0, t.stride(0)), None)
            P.keep.append(cp)
            P.add(lib.vv_conv_mfma, (C.byref(cp),), 'convT%d' % u)

        P.add(lib.vv_cube_erase, (Ga, B * HW0 * HW0, ws.cube.shape[2], lay.convs[0].cinp, ws.cube.data_ptr(),
                                  self._p(self.chmap, g0 * lay.convs[0].cinp), ws.erased.data_ptr(), ws.erased.stride(0),
                                  1 if self.y16 else 0), 'cube_erase')
        for l in lay.convs:
            if l.mode == L.IN_CAT:
                convT(l.up)
            if l.mode == L.IN_POOL:
                s = lay.convs[l.src]
                ys, pb = ws.y[s.idx], ws.pooled[l.idx]
                P.add(lib.vv_pool_act, (Ga, B, l.H, l.H, s.cout, ys.data_ptr(), ys.stride(0), self._p(ws.ab[0, s.idx]),
                                        self._p(ws.ab[1, s.idx]), abg, pb.data_ptr(), pb.stride(0), 1 if self.y16 else 0), 'pool%d' % l.idx)
            conv(l)
        last = lay.convs[-1]
        y = ws.y[last.idx]
        op = L.OutconvParams(Ga, B, HW0 * HW0, self.nf, y.data_ptr(), y.stride(0), self._p(ws.ab[0, last.idx]),
                             self._p(ws.ab[1, last.idx]), abg, pbase + 4 * lay.p['o.w'][0], pbase + 4 * lay.p['o.b'][0], U,
                             self._p(self.oc, g0), ws.cube.data_ptr(), ws.cube.shape[2], 1 if self.y16 else 0, ws.flow.data_ptr(),
                             ws.flow.shape[2], 0, self._p(self.tsrc, g0), self._p(self.tcoff, g0), ws.out4.data_ptr() if out4 else None,
                             ws.score.data_ptr(), ws.gscale.data_ptr() if train else None,
                             ws.dout4.data_ptr() if (train and not (self.fuse_outconv and not out4)) else None)
        P.keep.append(op)
        if train and self.fuse_outconv and not out4:
            P.fused_outconv = True
            P.add(lib.vv_outconv_fwdbwd, (C.byref(op), ws.gA_last.data_ptr(), ws.gA_last.stride(0), ws.ocpart.data_ptr(),
                                          self._p(ws.ab[2, last.idx]), self._p(ws.ab[3, last.idx]), ws.bnpart.data_ptr(),
                                          (1 if self.da16 else 0) | (2 if self.y16 else 0)), 'outconv')
        else:
            P.add(lib.vv_outconv_fwd, (C.byref(op),), 'outconv')
        return P

    def mark_dirty(self):
        """Parameters or running statistics were written by a kernel (Adam, train-mode BatchNorm): torch's version counters do not
        see raw-pointer writes, so the folded eval model is invalidated explicitly."""
        self._state_ver += 1

    def prepare_eval(self):
        """(Re)build the folded filters / biases and their packed panels when the parameters or running statistics changed since the
        last eval-mode forward (tensor version counters: Adam steps, load_state_dict, train-mode forwards all bump them)."""
        key = (self._state_ver, self.params._version, self.bufs._version, self.params.data_ptr())
        if self._eval_key == key:
            return
        lib, lay, G = self.lib, self.lay, self.G
        if self.params_eval is None:
            self.params_eval = torch.empty_like(self.params)
            self.packed_eval = torch.zeros_like(self.packed)
        st = self._stream()
        self.params_eval.copy_(self.params)          # transposed convs, 1x1 output conv: unchanged
        L.check(lib.vv_fold_bn(self.fold_table.data_ptr(), len(lay.convs), G, self.params.data_ptr(), lay.U, self.bufs.data_ptr(),
                               lay.UB, 1e-5, self.params_eval.data_ptr(), lay.U, st), 'fold_bn')
        L.check(lib.vv_pack_weights(self.pack_table.data_ptr(), self.pack_n, G, self.params_eval.data_ptr(), lay.U,
                                    self.packed_eval.data_ptr(), lay.UP, self.pack_max, st), 'pack (eval)')
        if self.wino:
            L.check(lib.vv_pack_wino(self.pack_table_w.data_ptr(), self.pack_w_n, G, self.params_eval.data_ptr(), lay.U,
                                     self.packed_eval.data_ptr(), lay.UP, self.pack_w_max, st), 'pack_wino (eval)')
            if self.w44_eval_mode not in ('0', 'off'):
                if getattr(self, '_w44_eval_table', None) is None and self._w44_eval_used:
                    keys = sorted(self._w44_eval_used)
                    ents = (L.PackEntry * len(keys))()
                    for i, k in enumerate(keys):
                        off, mode, K, KP, N, src = lay.pkw44[k]
                        ents[i] = L.PackEntry(lay.p[src][0], off, mode, K, KP, N)
                    self._w44_eval_table = (torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(self.device), len(keys),
                                            max(lay.pkw44[k][3] * lay.pkw44[k][4] for k in keys))
                t = getattr(self, '_w44_eval_table', None)
                if t is not None:
                    L.check(lib.vv_pack_wino44(t[0].data_ptr(), t[1], G, self.params_eval.data_ptr(), lay.U, self.packed_eval.data_ptr(), lay.UP,
                                               t[2], st), 'pack_wino44 (eval)')
        self._eval_key = key

    def _plan_eval(self, ws, B, out4=True):
        """Eval-mode forward on the folded model (see _alloc_state): cube_erase, 14 convs (+3 pools, +3 transposed convs), fused
        1x1 conv + score.  33 launches fewer arithmetic on every load path than the train-mode plan; same tensors otherwise."""
        lib, lay, Ga, g0 = self.lib, self.lay, self.Ga, self.g0
        U, UP = lay.U, lay.UP
        if self.params_eval is None:
            self.params_eval = torch.empty_like(self.params)
            self.packed_eval = torch.zeros_like(self.packed)
        P = _Plan()
        pbase, kbase = self._p(self.params_eval, g0 * U), self._p(self.packed_eval, g0 * UP)
        abg = lay.cmax
        one, zero = self._p(self.ab_ident[0], g0 * abg), self._p(self.ab_ident[1], g0 * abg)

        def src(l):
            if l.mode == L.IN_CUBE:
                return (L.IN_PLAIN, L.view(ws.erased, l.cinp, 0, ws.erased.stride(0)), None, None, L.NULL_VIEW, 0)
            if l.mode == L.IN_POOL:
                pb = ws.pooled[l.idx]
                return (L.IN_PLAIN, L.view(pb, l.cin, 0, pb.stride(0)), None, None, L.NULL_VIEW, 0)
            if l.mode == L.IN_ACT:
                y = ws.y[lay.convs[l.src].idx]
                return (L.IN_PLAIN, L.view(y, lay.convs[l.src].cout, 0, y.stride(0)), None, None, L.NULL_VIEW, 0)
            sk = lay.convs[l.skip]
            y, t = ws.y[sk.idx], ws.t[l.up]          # concat: the skip tensor is already activated -> identity (a, b)
            return (L.IN_CAT, L.view(y, sk.cout, 0, y.stride(0)), one, zero, L.view(t, t.shape[2], 0, t.stride(0)), sk.cout)

        P.add(lib.vv_cube_erase, (Ga, B * HW0 * HW0, ws.cube.shape[2], lay.convs[0].cinp, ws.cube.data_ptr(),
                                  self._p(self.chmap, g0 * lay.convs[0].cinp), ws.erased.data_ptr(), ws.erased.stride(0), 0), 'cube_erase')
        for l in lay.convs:
            if l.mode == L.IN_CAT:
                sidx, H, ci, co = lay.convT[l.up]
                y, t = ws.y[sidx], ws.t[l.up]
                cp = L.ConvParams(L.CONVT_FWD, L.IN_PLAIN, Ga, B, H, H, ci, ci, co, L.view(y, ci, 0, y.stride(0)), None, None, 0,
                                  L.NULL_VIEW, 0, 0, None, kbase + 4 * lay.pk['t%d.f' % l.up][0], UP,
                                  pbase + 4 * lay.p['t%d.b' % l.up][0], U, L.view(t, co, 0, t.stride(0)), None)
                P.keep.append(cp)
                P.add(lib.vv_conv_mfma, (C.byref(cp),), 'convT%d' % l.up)
            if l.mode == L.IN_POOL:
                sl = lay.convs[l.src]
                ys, pb = ws.y[sl.idx], ws.pooled[l.idx]
                P.add(lib.vv_pool_act, (Ga, B, l.H, l.H, sl.cout, ys.data_ptr(), ys.stride(0), one, zero, abg, pb.data_ptr(),
                                        pb.stride(0), 0), 'pool%d' % l.idx)
            mode, s0, a, b, s1, csplit = src(l)
            y = ws.y[l.idx]
            w44 = self._w44(B, l, False, evalm=True)
            if w44 and ('c%d.f' % l.idx) not in self._w44_eval_used:      # a panel no earlier plan used: repack on the next prepare_eval
                self._w44_eval_used.add('c%d.f' % l.idx)
                self._w44_eval_table, self._eval_key = None, None
            panel = (lay.pkw44 if w44 else lay.pkw if self.wino else lay.pk)['c%d.f' % l.idx][0]
            cp = L.ConvParams(L.CONV3, mode, Ga, B, l.H, l.H, l.cin, l.cinp, l.cout, s0, a, b, abg, s1, csplit, L.CONV_RELU | self.wino_flag, None,
                              kbase + 4 * panel, UP, pbase + 4 * lay.p['c%d.b' % l.idx][0], U, L.view(y, l.cout, 0, y.stride(0)), None)
            P.keep.append(cp)
            P.add(lib.vv_conv_wino44 if w44 else lib.vv_conv_wino if self.wino else lib.vv_conv_mfma, (C.byref(cp),), 'conv%d' % l.idx)
        last = lay.convs[-1]
        y = ws.y[last.idx]
        op = L.OutconvParams(Ga, B, HW0 * HW0, self.nf, y.data_ptr(), y.stride(0), one, zero, abg,
                             pbase + 4 * lay.p['o.w'][0], pbase + 4 * lay.p['o.b'][0], U, self._p(self.oc, g0), ws.cube.data_ptr(),
                             ws.cube.shape[2], 0, ws.flow.data_ptr(), ws.flow.shape[2], 0, self._p(self.tsrc, g0),
                             self._p(self.tcoff, g0), ws.out4.data_ptr() if out4 else None, ws.score.data_ptr(), None, None)
        P.keep.append(op)
        P.add(lib.vv_outconv_fwd, (C.byref(op),), 'outconv')
        return P

    def _plan_backward(self, ws, B):
        lib, lay, Ga, g0, d = self.lib, self.lay, self.Ga, self.g0, self.device
        U, UB, UP = lay.U, lay.UB, lay.UP
        nf = self.nf
        f = lambda *s: torch.empty(*s, device=d, dtype=torch.float32)
        P = _Plan()
        pbase, kbase = self._p(self.params, g0 * U), self._p(self.packed, g0 * UP)
        abg = lay.cmax
        HWp = HW0 * HW0
        last = lay.convs[-1]
        # buffers
        self._alloc_outconv_bwd(ws, B)
        ws.D = {l.idx: f(Ga, B * l.H * l.H, l.cin) for l in lay.convs if l.idx > 0}

        def dcat_views(m):
            """Data gradient of concat layer m as its two consumers read it: (skip half, upsampled half, split channel or 0).
            All-bf16 tensors: two dense planes [pixels][half] inside ws.D[m] (vv_conv_params.out1) -- an interleaved pixel row of
            2 x 32 bf16 channels hands each consumer 64 useful bytes of every 128-byte line (measured on the 32x32 level: BatchNorm
            backward of the skip layer 264 / 190 us against 172 / 113 us for the same tensor sizes read densely)."""
            dcat = ws.D[m.idx]
            skipc = lay.convs[m.skip].cout
            if self.split_dcat and 2 * skipc == m.cin and skipc & (skipc - 1) == 0:
                plane = B * m.H * m.H * skipc * 2                      # bytes
                return (L.view(dcat, skipc, 0, dcat.stride(0)), L.view(dcat.data_ptr() + plane, skipc, 0, dcat.stride(0)), skipc)
            return (L.view(dcat, m.cin, 0, dcat.stride(0)), L.View(dcat.data_ptr(), dcat.stride(0), m.cin, skipc), 0)
        ws.DT = [f(Ga, B * H * H, ci) for (_, H, ci, co) in lay.convT]
        ws.dz2 = [f(Ga, B * HWp * nf), f(Ga, B * HWp * nf)]   # dy of consecutive layers alternate (weight-grad runs on a side stream)
        ws.dz = ws.dz2[0]
        nblk = [lib.vv_bn_bwd_nblk(B, l.H, l.H, l.cout) for l in lay.convs]
        ws.bnscr = f(Ga, 2 * lay.cmax)
        ws.bscr = f(Ga, (B * HWp + 1023) // 1024 * lay.cmax)
        ws.dstats = f(Ga, max(max(lib.vv_conv_ntiles2(B, l.H, l.H, L.CONV3, self.cflag), lib.vv_conv_ntiles(B, l.H, l.H),
                                  lib.vv_wino_ntiles(B, l.H)) * 2 * l.cin for l in lay.convs if l.mode == L.IN_CAT))
        # wgrad split-K choice: ~1024 workgroups per launch
        wplan = {}
        wmax = 0
        for l in lay.convs:
            nci, nco = (l.cinp + 31) // 32, l.cout // 32
            nt = lib.vv_wgrad_ntiles(L.CONV3, B, l.H, l.H)
            if self.wino_wgrad and not self.cflag:
                # three workgroups per CU; 768 slots also at small batches (swept 256 / 384 / 512 / 768 at B = 32 and 256: fewer slabs
                # shorten the grouped reduction but lengthen the weight-gradient launches by more)
                ks = _pick_ksplit(Ga * nci * nco, nt, ncu=3 * self.ncu, max_wg=12 * self.ncu)
            else:
                ks = _pick_ksplit(Ga * nci * nco, nt, ncu=self.ncu, max_wg=4 * self.ncu)
            wplan['c%d' % l.idx] = (ks, nci * nco * ks)
            if self.cflag and self.bf16_wgrad:
                # bf16 weight gradient: HBM bound, one workgroup (up to 512 registers per lane) per CU
                ntb, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
                # (the flags of the launch go with the query: the all-bf16 weight gradient runs the LDS-ring kernel, whose tiling differs)
                wfl = (L.WGRAD_DY_BF16 | L.WGRAD_X_BF16) if (self.dz16 and self.y16) else 0
                if lib.vv_wgrad_bf16_plan(L.CONV3 | (wfl << 8), B, l.H, l.H, l.cinp, l.cout, C.byref(ntb), C.byref(nblk), C.byref(kw)):
                    ks = self._ks_bf16(Ga * nblk.value, ntb.value)
                    wplan['c%d' % l.idx] = (ks, nci * nco * ks * kw.value, kw.value)
            wmax = max(wmax, wplan['c%d' % l.idx][1])
        for u, (_, H, ci, co) in enumerate(lay.convT):
            nci, nco = ci // 32, co // 32
            nt = lib.vv_wgrad_ntiles(L.CONVT_FWD, B, H, H)
            ks = _pick_ksplit(Ga * nci * nco, nt, ncu=self.ncu, max_wg=4 * self.ncu)
            wplan['t%d' % u] = (ks, nci * nco * ks)
            if self.cflag and self.bf16_wgrad:
                ntb, nblk, kw = C.c_int32(), C.c_int32(), C.c_int32()
                if lib.vv_wgrad_bf16_plan(L.CONVT_FWD, B, H, H, ci, co, C.byref(ntb), C.byref(nblk), C.byref(kw)):
                    ks = self._ks_bf16(Ga * nblk.value, ntb.value)
                    wplan['t%d' % u] = (ks, nci * nco * ks * kw.value, kw.value)
            wmax = max(wmax, wplan['t%d' % u][1])
        # weight-gradient slabs: one region per layer (the reductions of a whole gradient bucket run as ONE grouped launch after the
        # bucket's last weight-gradient kernel, so every layer's slabs must still be there)
        woff, wtot = {}, 0
        for key, wpl in wplan.items():
            woff[key] = wtot
            wtot += wpl[1] * 9 * 1024
        ws.wpart = f(Ga, wtot)
        wpg = ws.wpart.stride(0)
        group_reduce = os.environ.get('VV_GROUP_REDUCE', '1') != '0'
        pending = []                     # (key, kind, cin, cinp, cout, nslab) of weight gradients whose reduction is still to be launched

        def flush_reduce(label, **kw_):
            """one vv_wgrad_reduce_grouped launch for everything in `pending` (all of it lies in one gradient bucket)"""
            ents = (L.ReduceEntry * len(pending))()
            start = 0
            for i, (key, kind, cin, cinp, cout, nslab) in enumerate(pending):
                gptr, gstride = self._g(key + '.w')
                ents[i] = L.ReduceEntry(kind, cin, cout, cout // 32, nslab, start, woff[key], (gptr - self.grads.data_ptr()) // 4, gstride)
                start += ((cinp + 31) // 32) * (cout // 32) * 36
            tab = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(d)
            P.keep.append(tab)
            P.add(lib.vv_wgrad_reduce_grouped, (tab.data_ptr(), len(pending), start, Ga, ws.wpart.data_ptr(), wpg, self.grads.data_ptr()),
                  label, **kw_)
            del pending[:]

        # 1x1 output conv
        y = ws.y[last.idx]
        P.add(lib.vv_outconv_bwd, (Ga, B, HWp, nf, ws.dout4.data_ptr(), y.data_ptr(), y.stride(0), self._p(ws.ab[0, last.idx]),
                                   self._p(ws.ab[1, last.idx]), abg, pbase + 4 * lay.p['o.w'][0], U, ws.gA_last.data_ptr(),
                                   ws.gA_last.stride(0), ws.ocpart.data_ptr(),
                                   # ... and the BatchNorm-backward partial sums of the last conv layer (no reduction pass for it)
                                   self._p(ws.ab[2, last.idx]), self._p(ws.ab[3, last.idx]), ws.bnpart.data_ptr(),
                                   (1 if self.da16 else 0) | (2 if self.y16 else 0)), 'outconv_bwd')
        assert self._g('o.w')[1] == self._g('o.b')[1]
        P.add(lib.vv_outconv_bwd_reduce, (Ga, nf, B, ws.ocpart.data_ptr(), self._p(self.oc, g0), self._g('o.w')[0],
                                          self._g('o.b')[0], self._g('o.w')[1]), 'outconv_bwd_reduce')

        def dA_for(l):
            """(dA view, dpool ptr, dpool gstride) feeding the BN backward of conv layer l."""
            i = l.idx
            if i == last.idx:
                return L.view(ws.gA_last, nf, 0, ws.gA_last.stride(0)), None, 0
            nxt = lay.convs[i + 1] if i + 1 < len(lay.convs) else None
            # who consumes act(y_i)?
            cons_t = [u for u, (sidx, _, _, _) in enumerate(lay.convT) if sidx == i]
            skip_of = [m for m in lay.convs if m.mode == L.IN_CAT and m.skip == i]
            pool_of = [m for m in lay.convs if m.mode == L.IN_POOL and m.src == i]
            if cons_t:
                t = ws.DT[cons_t[0]]
                return L.view(t, t.shape[2], 0, t.stride(0)), None, 0
            if skip_of:
                m = skip_of[0]
                dp = ws.D[pool_of[0].idx] if pool_of else None
                return (dcat_views(m)[0], dp.data_ptr() if dp is not None else None,
                        dp.stride(0) if dp is not None else 0)
            m = nxt
            dn = ws.D[m.idx]
            return L.view(dn, m.cin, 0, dn.stride(0)), None, 0

        order = [l.idx for l in reversed(lay.convs)]            # backward visiting order: 13, 12, ..., 0

        def fused_producer(l):
            """Index of the layer whose BatchNorm-backward sums the data-gradient launch of conv layer l can leave behind
            (vv_conv_params.bn_partial): l reads relu(bn(y_j)) of the layer right before it and nobody else consumes that activation
            (no pooling / skip / transposed-conv consumer), so dA_j is exactly this launch's output.  fp32 Winograd path (every
            level), and all-bf16 tensors on the 32x32 level (vv_conv_mfma's kernel with 32-wide N tiles: HBM-latency-bound launches
            that absorb the extra read of z; on the deeper levels the stores belong to the single-issue producer waves of
            vv_conv_bf16.hip, where the sums would cost more than the separate pass)."""
            if not self.fuse_bn_sums or l.mode != L.IN_ACT or l.idx == 0:
                return None
            if self.wino and not self.cflag:
                pass
            elif self.y16 and self.da16 and self.dz16 and len(wplan['c%d' % l.idx]) > 2 and l.H == 32 and l.cin % 64 != 0:
                pass                           # (that launch carries VV_CONV_ALLSRC_BF16 | VV_CONV_OUT_BF16: dgrad_flags)
            else:
                return None
            j = l.src
            if j != l.idx - 1 or j == last.idx:
                return None
            if any(sidx == j for (sidx, _, _, _) in lay.convT) or any(m.mode == L.IN_CAT and m.skip == j for m in lay.convs) \
                    or any(m.mode == L.IN_POOL and m.src == j for m in lay.convs):
                return None
            return j

        fused = {fused_producer(l) for l in lay.convs} - {None}
        # round 6: conv layers whose activation feeds ONLY a transposed conv (7 / 9 / 11): dA is that transposed conv's data gradient
        # (vv_conv_mfma, VV_CONVT_DGRAD), which leaves the first BatchNorm-backward pass in its epilogue -- fp32 path only (the all-bf16
        # form of that launch was measured slower than launch + separate pass on config 4); VV_FUSE_BN_SUMS_T=0 keeps the separate
        # reduce pass for these three (A/B)
        fused_t = {}
        if self.fuse_bn_sums and os.environ.get('VV_FUSE_BN_SUMS_T', '1') != '0' and not self.cflag:
            for u, (sidx, _, _, _) in enumerate(lay.convT):
                others = any(m.mode == L.IN_CAT and m.skip == sidx for m in lay.convs) or any(m.mode == L.IN_POOL and m.src == sidx for m in lay.convs) \
                    or any(m.mode == L.IN_ACT and m.src == sidx for m in lay.convs)
                if not others and sidx != last.idx:
                    fused_t[u] = sidx
        fused |= set(fused_t.values())

        def dgrad_flags(i):
            # pad0 of layer i's data-gradient launch (also decides its tile count: vv_conv_ntiles2)
            dz16 = bool(self.dz16 and len(wplan['c%d' % i]) > 2)
            return (self.cflag | ((L.CONV_ALLSRC_BF16 if self.y16 else L.CONV_SRC_BF16) if dz16 else 0) |
                    (L.CONV_OUT_BF16 if self.da16 else 0))

        def conv_bwd(l):
            i = l.idx
            y = ws.y[i]
            # mixed precision: both consumers of dy (data gradient, weight gradient) round it to bf16 -- let BatchNorm backward
            # store it that way (same values, half the bytes written once and read twice)
            wpl = wplan['c%d' % i]
            dz16 = bool(self.dz16 and len(wpl) > 2)
            pos = order.index(i)
            dzb = ws.dz2[pos % 2]
            # the layer visited two steps earlier used the same dy buffer: its weight-grad (side stream) must be done
            reuse_wait = ('wdone%d' % order[pos - 2],) if pos >= 2 else ()
            dA, dpool, dpg = dA_for(l)
            from_outconv = i == last.idx           # its partial sums were written by vv_outconv_bwd
            from_dgrad = i in fused                # ... by the data-gradient launch of layer i + 1
            bp = L.BnBwdParams(Ga, B, l.H, l.H, l.cout,
                               (L.BNBWD_DZ_BF16 if dz16 else 0) | (L.BNBWD_PARTIALS_PER_CUBE if from_outconv else 0) |
                               ((L.BNBWD_PARTIALS_PER_TTILE if i in fused_t.values() else
                                 L.BNBWD_PARTIALS_PER_TILE44 if self._w44(B, lay.convs[i + 1], True) else L.BNBWD_PARTIALS_PER_TILE if self.wino
                                 else L.BNBWD_PARTIALS_PER_CTILE) if from_dgrad else 0) |
                               (L.BNBWD_DA_BF16 if self.da16 else 0) | (L.BNBWD_Y_BF16 if self.y16 else 0), y.data_ptr(), y.stride(0), self._p(ws.ab[0, i]), self._p(ws.ab[1, i]),
                               self._p(ws.ab[2, i]), self._p(ws.ab[3, i]), abg, dA, dpool, dpg, dzb.data_ptr(), dzb.stride(0),
                               ws.bnpart.data_ptr())
            P.keep.append(bp)
            if not (from_outconv or from_dgrad):
                P.add(lib.vv_bn_bwd_reduce, (C.byref(bp),), 'bn_bwd_reduce%d' % i, wait=reuse_wait)
            P.add(lib.vv_bn_bwd_apply, (C.byref(bp), pbase + 4 * lay.p['c%d.g' % i][0], U, self._g('c%d.g' % i)[0],
                                        self._g('c%d.beta' % i)[0], self._g('c%d.g' % i)[1], ws.bnscr.data_ptr()), 'bn_bwd_apply%d' % i,
                  record='dy%d' % i, wait=reuse_wait if (from_outconv or from_dgrad) else ())
            # data gradient
            if i > 0:
                Dl = ws.D[i]
                w44 = self._w44(B, l, True)
                cp = L.ConvParams(L.CONV3, L.IN_PLAIN, Ga, B, l.H, l.H, l.cout, l.cout, l.cin,
                                  L.View(dzb.data_ptr(), dzb.stride(0), l.cout, 0), None, None, 0, L.NULL_VIEW, 0,
                                  dgrad_flags(i) | self.wino_flag, None,
                                  kbase + 4 * (lay.pkw44 if w44 else lay.pkw if self.wino else lay.pk)['c%d.d' % i][0], UP, None, 0,
                                  L.view(Dl, l.cin, 0, Dl.stride(0)),
                                  # concat layers: per-tile column sums of the data gradient = the transposed conv's bias gradient
                                  ws.dstats.data_ptr() if l.mode == L.IN_CAT else None)
                if l.mode == L.IN_CAT:
                    v0, v1, osplit = dcat_views(l)
                    if osplit:
                        cp.out, cp.out1, cp.osplit = v0, v1, osplit
                j = fused_producer(l)
                if j is not None:                  # the first pass of layer j's BatchNorm backward rides on this launch's epilogue
                    yj = ws.y[j]
                    cp.bn_z, cp.bn_z_gstride = yj.data_ptr(), yj.stride(0)
                    cp.bn_a, cp.bn_b = self._p(ws.ab[0, j]), self._p(ws.ab[1, j])
                    cp.bn_mean, cp.bn_invstd = self._p(ws.ab[2, j]), self._p(ws.ab[3, j])
                    cp.bn_gstride, cp.bn_partial = abg, ws.bnpart.data_ptr()
                P.keep.append(cp)
                # paired schedule: the MFMA data-gradient runs alone (the side stream has drained) ...
                P.add(lib.vv_conv_wino44 if w44 else lib.vv_conv_wino if self.wino else lib.vv_conv_mfma, (C.byref(cp),), 'dgrad%d' % i,
                      record='D%d' % i, pwait=('*side',))
            # weight gradient (side stream: only depends on dy_i and forward products).  Paired schedule: ... and the
            # weight-gradient starts when it is done, sharing the chip with the HBM-bound BatchNorm backward of the
            # next layer only.
            mode, s0, a, b, s1, csplit, chmap = self._src_for(ws, l)
            ks, kw = wpl[0], (wpl[2] if len(wpl) > 2 else 0)           # kw > 0: the bf16-operand kernel (mixed precision)
            # pad0 bit 8: Winograd F(2x2,3x3) form of the weight gradient (same tiles / slabs, 2.25x fewer MFMA cycles)
            wflag = ((L.WGRAD_DY_BF16 if dz16 else 0) | (L.WGRAD_X_BF16 if (dz16 and self.y16) else 0)) if kw else \
                (self.wgrad_flag if self.wino_wgrad else 0)
            wp = L.WgradParams(L.CONV3, mode, Ga, B, l.H, l.H, l.cin, l.cinp, l.cout, ks, s0, a, b, abg, s1, csplit,
                               wflag, chmap,
                               L.View(dzb.data_ptr(), dzb.stride(0), l.cout, 0), ws.wpart.data_ptr() + 4 * woff['c%d' % i], wpg)
            P.keep.append(wp)
            P.add(lib.vv_wgrad_bf16 if kw else lib.vv_wgrad_mfma, (C.byref(wp),), 'wgrad%d' % i, stream=1, wait=('dy%d' % i,),
                  record='wdone%d' % i, pwait=('*main',))
            if group_reduce:
                pending.append(('c%d' % i, L.CONV3, l.cin, l.cinp, l.cout, ks * max(kw, 1)))
                if i in (4, 0):            # last weight gradient of the deep-encoder / shallow-encoder bucket (self.gb)
                    flush_reduce('wgrad_reduce%d' % i, stream=1)
            else:
                P.add(lib.vv_wgrad_reduce, (L.CONV3, Ga, l.cin, l.cinp, l.cout, ks * max(kw, 1), ws.wpart.data_ptr() + 4 * woff['c%d' % i], wpg)
                      + self._g('c%d.w' % i), 'wgrad_reduce%d' % i, stream=1)

        def convT_bwd(u, m):
            """m: the CAT conv layer that consumed convT u; its data gradient holds d(convT out) in channels [skipC, cin)."""
            sidx, H, ci, co = lay.convT[u]
            skipc = lay.convs[m.skip].cout
            dy = dcat_views(m)[1]
            DT = ws.DT[u]
            cp = L.ConvParams(L.CONVT_DGRAD, L.IN_PLAIN, Ga, B, H, H, co, co, ci, dy, None, None, 0, L.NULL_VIEW, 0,
                              self.cflag | (((L.CONV_ALLSRC_BF16 if self.y16 else L.CONV_SRC_BF16) | L.CONV_OUT_BF16) if self.da16 else 0), None,
                              kbase + 4 * lay.pk['t%d.d' % u][0], UP, None, 0, L.view(DT, ci, 0, DT.stride(0)), None)
            if u in fused_t:                   # the first pass of layer sidx's BatchNorm backward rides on this launch's epilogue
                yj = ws.y[sidx]
                cp.bn_z, cp.bn_z_gstride = yj.data_ptr(), yj.stride(0)
                cp.bn_a, cp.bn_b = self._p(ws.ab[0, sidx]), self._p(ws.ab[1, sidx])
                cp.bn_mean, cp.bn_invstd = self._p(ws.ab[2, sidx]), self._p(ws.ab[3, sidx])
                cp.bn_gstride, cp.bn_partial = abg, ws.bnpart.data_ptr()
            P.keep.append(cp)
            P.add(lib.vv_conv_mfma, (C.byref(cp),), 'dgradT%d' % u, pwait=('*side',))
            ntd = lib.vv_wino44_ntiles(B, m.H) if self._w44(B, m, True) else lib.vv_wino_ntiles(B, m.H) if self.wino else \
                lib.vv_conv_ntiles2(B, m.H, m.H, L.CONV3, dgrad_flags(m.idx))
            P.add(lib.vv_bias_from_partials, (Ga, m.cin, ntd, skipc, co, ws.dstats.data_ptr(), ntd * 2 * m.cin)
                  + self._g('t%d.b' % u), 'convT_bias%d' % u)
            y = ws.y[sidx]
            wpl = wplan['t%d' % u]
            ks, kw = wpl[0], (wpl[2] if len(wpl) > 2 else 0)
            wp = L.WgradParams(L.CONVT_FWD, L.IN_ACT, Ga, B, H, H, ci, ci, co, ks, L.view(y, ci, 0, y.stride(0)),
                               self._p(ws.ab[0, sidx]), self._p(ws.ab[1, sidx]), abg, L.NULL_VIEW, 0,
                               ((L.WGRAD_DY_BF16 | (L.WGRAD_X_BF16 if self.y16 else 0)) if (kw and self.da16) else 0), None, dy,
                               ws.wpart.data_ptr() + 4 * woff['t%d' % u], wpg)
            P.keep.append(wp)
            # side stream: its dy is the concat layer's data gradient (main stream) -- wait for it
            P.add(lib.vv_wgrad_bf16 if kw else lib.vv_wgrad_mfma, (C.byref(wp),), 'wgradT%d' % u, stream=1,
                  wait=('D%d' % m.idx,), pwait=('*main',))
            if group_reduce:
                pending.append(('t%d' % u, L.CONVT_FWD, ci, ci, co, ks * max(kw, 1)))
                if u == 0:                 # last weight gradient of the decoder bucket
                    flush_reduce('wgradT_reduce0', stream=1, record='sideT0')
            else:
                P.add(lib.vv_wgrad_reduce, (L.CONVT_FWD, Ga, ci, ci, co, ks * max(kw, 1), ws.wpart.data_ptr() + 4 * woff['t%d' % u], wpg)
                      + self._g('t%d.w' % u), 'wgradT_reduce%d' % u, stream=1, record='sideT%d' % u)

        for l in reversed(lay.convs):
            conv_bwd(l)
            if l.mode == L.IN_CAT:
                convT_bwd(l.up, l)
        assert not pending
        return P

    # ------------------------------------------------------------------------------------------ public operations
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def set_input_nchw(self, x, x_of):
        """x [B,3T,32,32], x_of [B,2Tf,32,32] fp32 NCHW (the reference's forward(x, x_of) arguments)."""
        B = x.shape[0]
        ws = self.workspace(B)
        x = x.contiguous().float()
        L.check(self.lib.vv_nchw_to_nhwc(B, x.shape[1], HW0 * HW0, x.data_ptr(), ws.cube.data_ptr(), self._stream()), 'nchw_to_nhwc')
        if x_of is not None and ws.flow.numel():
            x_of = x_of.contiguous().float()
            L.check(self.lib.vv_nchw_to_nhwc(B, x_of.shape[1], HW0 * HW0, x_of.data_ptr(), ws.flow.data_ptr(), self._stream()),
                    'nchw_to_nhwc')
        return ws

    def set_input_cubes(self, raw_u8, flow, idx=None, B=None):
        """Device-resident cube store: raw uint8 [N,T,32,32,3], flow fp32 [N,Tf,32,32,2]; idx int64 [B] (or first B)."""
        B = int(idx.numel()) if idx is not None else (B if B is not None else raw_u8.shape[0])
        ws = self.workspace(B)
        L.check(self.lib.vv_cube_gather(B, self.tot_raw, self.tot_of, HW0 * HW0, idx.data_ptr() if idx is not None else None,
                                        raw_u8.data_ptr(), flow.data_ptr() if flow is not None else None, ws.cube.data_ptr(),
                                        ws.flow.data_ptr(), self._stream()), 'cube_gather')
        return ws

    def forward(self, ws, train, outputs=True):
        """Runs the grouped forward.  train=True: batch statistics, running-stat update, and the loss gradient of the 1x1 output
        conv: as ws.dout4 = d(loss)/d(out) (outputs=True, or VV_FUSE_OUTCONV=0), or -- outputs=False with the fused output conv --
        already carried through that conv by vv_outconv_fwdbwd (ws.gA_last / ocpart / bnpart; ws.dout4 is NOT written).  Which of
        the two happened is recorded in ws.fused_fwd and backward() follows it.
        outputs=False: only the per-cube scores are needed -- the reconstruction ws.out4 is not written."""
        if not train and self.eval_fold:
            self.prepare_eval()
        plan = (ws.fwd if outputs else ws.fwdq)[bool(train)]
        plan.run(self._stream())
        ws.out4_valid = bool(outputs)
        if train:
            ws.fused_fwd = bool(getattr(plan, 'fused_outconv', False))
        if train:
            self.bump_nbt()
            self.mark_dirty()
        return ws.score

    def nbt_call(self):
        """(fn, args, label) of the launch that advances BatchNorm's num_batches_tracked of the active UNets (first party: a captured
        train step then holds no framework kernel)."""
        n = len(self.lay.convs)
        return (self.lib.vv_counter_add, (self.nbt.data_ptr() + 8 * self.g0 * n, self.Ga * n, 1), 'nbt')

    def bump_nbt(self):
        fn, args, label = self.nbt_call()
        L.check(fn(*args, self._stream()), label)

    def backward(self, ws, fused=None):
        """Gradients of everything wrt ws.dout4 into self.grads (conv biases in front of BatchNorm get exact zeros).
        The plan follows the LAST train-mode forward on this workspace (ws.fused_fwd): behind the scores-only train plan
        (outputs=False), whose vv_outconv_fwdbwd already did the output conv's backward with the loss gradient, ws.dout4 is neither
        written nor read and the plan has no vv_outconv_bwd launch; behind every other forward the plan starts from ws.dout4.
        fused: optional assertion of which of the two the caller expects -- a mismatch raises instead of running a backward on
        stale buffers (True is accepted behind an unfused forward only when the library runs with VV_FUSE_OUTCONV=0, where
        "fused" has always meant "scores-only forward")."""
        ran = getattr(ws, 'fused_fwd', None)
        if ran is None:
            raise RuntimeError('backward() needs a train-mode forward on this workspace first')
        if fused is not None and bool(fused and self.fuse_outconv) != ran:
            raise RuntimeError('backward(fused=%r) behind a forward that %s the fused output-conv pass: the buffers of the other plan '
                               'are stale' % (fused, 'ran' if ran else 'did not run'))
        self.backward_plan(ws, ran).run(self._stream())

    def _score_row_index(self, device):
        """device index tensors of the raw / flow UNets' rows in ws.score, built once: indexing with a python list uploads it -- a
        synchronous pageable copy that stalls the host behind the stream -- on every call"""
        rows = getattr(self, '_score_rows', None)
        if rows is None or rows[0].device != device:
            units = self.units[self.g0:self.g0 + self.Ga]
            mk = lambda role: torch.tensor([i for i, u in enumerate(units) if u.role == role], dtype=torch.long, device=device)
            rows = self._score_rows = (mk('raw'), mk('of'))
        return rows

    def losses(self, ws):
        """(loss_raw, loss_of) as device scalars from the per-cube squared errors (train.py:385-392)."""
        HWp = HW0 * HW0
        sc = ws.score
        raw_rows, of_rows = self._score_row_index(sc.device)
        l_raw = sc.index_select(0, raw_rows).sum() / (ws.B * raw_rows.numel() * RAW_C * HWp)
        l_of = sc.index_select(0, of_rows).sum() / (ws.B * of_rows.numel() * OF_C * HWp) if of_rows.numel() else None
        return l_raw, l_of

    def cube_scores(self, ws):
        """per-cube raw / flow squared-error sums ([B] each), train.py:421-426."""
        rows = self._score_row_index(ws.score.device)
        r = ws.score.index_select(0, rows[0]).sum(0)
        o = ws.score.index_select(0, rows[1]).sum(0) if rows[1].numel() else None
        return r, o

    def adam_step(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, grads=None):
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        self._adam_t += 1
        self.mark_dirty()
        g = self.grads if grads is None else grads
        st = self._stream()
        # step counter + bias corrections on the device (1 - beta^t in double like torch.optim.Adam, train.py:376), then one
        # launch over the whole bank reading the bucket-major gradient buffer
        L.check(self.lib.vv_adam_tick(self._adam_t_dev.data_ptr(), lr, beta1, beta2, self._adam_sc.data_ptr(), st), 'adam_tick')
        L.check(self.lib.vv_adam_bucketed(self.G, self.lay.U, len(self.gb) - 1, self._gb_c, self.params.data_ptr(), g.data_ptr(),
                                          self.adam_m.data_ptr(), self.adam_v.data_ptr(), self._adam_sc.data_ptr(), beta1, beta2,
                                          eps, grad_scale, st), 'adam')

    def outputs_nchw(self, ws):
        """(of_out [B,2*n_of,32,32], raw_out [B,3*n_raw,32,32]) in the reference's channel order."""
        if not getattr(ws, 'out4_valid', False):
            raise RuntimeError('the last forward on this workspace did not store the reconstructions (fused train / scoring steps skip '
                               'that store): use forward(ws, train, outputs=True) or FusedTrainer.keep_outputs = True')
        B, HWp = ws.B, HW0 * HW0
        units = self.units[self.g0:self.g0 + self.Ga]
        raw_o = torch.empty(B, RAW_C * ws.n_raw, HW0, HW0, device=self.device)
        of_o = torch.empty(B, OF_C * ws.n_of, HW0, HW0, device=self.device) if ws.n_of else None
        ri = oi = 0
        for i, u in enumerate(units):
            src = ws.out4[i].data_ptr()
            if u.role == 'raw':
                L.check(self.lib.vv_out4_to_nchw(B, HWp, RAW_C, src, raw_o.data_ptr(), raw_o.shape[1], ri * RAW_C, self._stream()), 'out4')
                ri += 1
            else:
                L.check(self.lib.vv_out4_to_nchw(B, HWp, OF_C, src, of_o.data_ptr(), of_o.shape[1], oi * OF_C, self._stream()), 'out4')
                oi += 1
        return of_o, raw_o

    def set_dout_nchw(self, ws, d_of, d_raw):
        """Load upstream gradients (NCHW, same shapes as outputs_nchw) into ws.dout4."""
        B, HWp = ws.B, HW0 * HW0
        units = self.units[self.g0:self.g0 + self.Ga]
        ri = oi = 0
        for i, u in enumerate(units):
            dst = ws.dout4[i].data_ptr()
            if u.role == 'raw':
                if d_raw is None:
                    ws.dout4[i].zero_()
                else:
                    L.check(self.lib.vv_nchw_to_out4(B, HWp, RAW_C, d_raw.data_ptr(), d_raw.shape[1], ri * RAW_C, dst, self._stream()), 'to_out4')
                ri += 1
            else:
                if d_of is None:
                    ws.dout4[i].zero_()
                else:
                    L.check(self.lib.vv_nchw_to_out4(B, HWp, OF_C, d_of.data_ptr(), d_of.shape[1], oi * OF_C, dst, self._stream()), 'to_out4')
                oi += 1

    def train_step(self, ws, lr=1e-3, eps=1e-7, grad_scale=1.0, allreduce=None):
        """Fused fast path: forward (train) -> backward -> [gradient all-reduce] -> Adam.  No host sync."""
        self.forward(ws, True, outputs=False)
        self.backward(ws, fused=True)
        if allreduce is not None:
            allreduce(self.grads)
        self.adam_step(lr=lr, eps=eps, grad_scale=grad_scale)
        return ws.score
