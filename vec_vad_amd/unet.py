"""Drop-in module surface of the reference's ``model/unet.py`` on top of the HIP UNet-bank engine.

Same class names, constructor arguments, ``forward(x, x_of) -> (of_out, raw_out, of_tgt, raw_tgt)`` contract,
sub-module tree and therefore the same ``state_dict`` keys as the reference (model/unet.py:4-70, 73-170, 270-408,
559-617), so a reference-trained ``*_model_*.npy`` loads unchanged (and the other way round).  What differs is
what runs: the whole bank of UNets executes as one grouped sequence of hand-written gfx950 kernels
(vec_vad_amd/bank.py -> libvecvad_hip.so).  There is no PyTorch-op fallback: calling a sub-block on its own, or
running the model on a non-GPU tensor, raises.

Parameters are created with the stock torch initialisers in the reference's module order (so ``torch.manual_seed(s)``
gives the same initial weights as the reference), then moved into one flat fp32 buffer per bank; every
``nn.Parameter`` / BatchNorm buffer is a view into it.
"""
import torch
import torch.nn as nn

from . import _lib as L
from .bank import UNetBank, UnitSpec, conv_key_to_state_name, RAW_C, OF_C

_NO_STANDALONE = ('this block only carries parameters for the fused HIP UNet bank; call the enclosing '
                  'SelfCompleteNet* model (vec_vad_amd has no per-block PyTorch fallback)')


def engine_width(features_root):
    """Width the HIP bank runs a model of ``features_root`` at.  The MFMA tiles are built for 32- and 64-wide first levels
    (every shipped config.cfg has 32, SelfCompleteNet1raw1of defaults to 64, model/unet.py:74,271,563); any other width up to
    64 runs EMBEDDED in the next engine width: the extra channels carry zero filters, zero BatchNorm gamma / beta and zero
    biases, so they hold exact zeros in the forward pass (relu(0 * xhat + 0) = 0, and a zero adds exactly nothing to a
    consumer's sum), receive exact-zero gradients in the backward pass (dz = gamma * ... = 0, activation 0, xhat 0) and are
    left at zero by Adam (m = v = 0 -> update 0 / (0 + eps) = 0): the embedded model is the reference's model, not an
    approximation of it."""
    nf = int(features_root)
    if nf in (32, 64):
        return nf
    if 0 < nf < 32:
        return 32
    if 32 < nf < 64:
        return 64
    raise L.VecVadHipError('features_root = %d: the HIP UNet bank is built for first-level widths up to 64 '
                           '(32 = every shipped config.cfg, 64 = SelfCompleteNet1raw1of default)' % nf)


def _embed_pieces(lay, key, mshape):
    """How a module tensor of shape ``mshape`` sits inside the bank tensor ``key`` (engine layout ``lay``): a list of
    (module index, bank index) pairs.  One top-left block for everything except the first conv of an ``up`` block, whose input is
    cat([skip, upsampled]) (model/unet.py:59): the module's upsampled half starts at channel cin_m / 2, the engine's at cin_e / 2."""
    kind, field = key.split('.')
    if kind[0] == 'c' and field == 'w':
        l = lay.convs[int(kind[1:])]
        co, ci = mshape[0], mshape[1]
        if l.mode == L.IN_CAT:
            h, he = ci // 2, l.cin // 2
            return [((slice(None), slice(0, h)), (slice(0, co), slice(0, h))),
                    ((slice(None), slice(h, 2 * h)), (slice(0, co), slice(he, he + h)))]
        return [((slice(None), slice(None)), (slice(0, co), slice(0, ci)))]
    if len(mshape) >= 2:              # transposed conv [ci, co, 3, 3], 1x1 output conv [out_c, nf, 1, 1]
        return [((slice(None), slice(None)), (slice(0, mshape[0]), slice(0, mshape[1])))]
    return [((slice(None),), (slice(0, mshape[0]),))]


class double_conv(nn.Module):
    """(conv3x3 => BN => ReLU) * 2 -- parameter holder (reference: model/unet.py:4-20)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True),
            nn.Conv2d(out_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))

    def forward(self, x):
        raise RuntimeError(_NO_STANDALONE)


class inconv(nn.Module):
    """reference: model/unet.py:22-32"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x):
        raise RuntimeError(_NO_STANDALONE)


class down(nn.Module):
    """MaxPool2d(2) + double_conv (reference: model/unet.py:34-44); index 1 of ``mpconv`` holds the convs."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), double_conv(in_ch, out_ch))

    def forward(self, x):
        raise RuntimeError(_NO_STANDALONE)


class up(nn.Module):
    """ConvTranspose2d(k3,s2,p1,op1) + cat + double_conv (reference: model/unet.py:46-61)."""

    def __init__(self, in_ch, out_ch, bilinear=False):
        super().__init__()
        if bilinear:
            raise NotImplementedError('the bilinear branch is never used by VEC_VAD (model/unet.py:49-52)')
        self.bilinear = False
        self.up = nn.ConvTranspose2d(in_ch, in_ch // 2, 3, stride=2, padding=1, output_padding=1)
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x1, x2):
        raise RuntimeError(_NO_STANDALONE)


class outconv(nn.Module):
    """1x1 conv (reference: model/unet.py:63-70)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, 1)

    def forward(self, x):
        raise RuntimeError(_NO_STANDALONE)


class _BankFn(torch.autograd.Function):
    """autograd bridge: one node for the whole bank (forward + backward are grouped HIP launches)."""

    @staticmethod
    def forward(ctx, x, x_of, model, train, *params):
        bank = model._bank
        ws = bank.set_input_nchw(x, x_of)
        bank.forward(ws, train)
        ws.stamp = getattr(ws, 'stamp', 0) + 1
        ctx.model, ctx.ws, ctx.stamp, ctx.train = model, ws, ws.stamp, train
        of_o, raw_o = bank.outputs_nchw(ws)
        if of_o is None:
            of_o = x.new_zeros(0)
        return of_o, raw_o

    @staticmethod
    def backward(ctx, d_of, d_raw):
        model, ws = ctx.model, ctx.ws
        bank = model._bank
        if ws.stamp != ctx.stamp:
            raise RuntimeError('the UNet-bank workspace was overwritten by a later forward of the same batch size '
                               'before this backward ran; call backward before the next forward')
        if not ctx.train:
            raise RuntimeError('backward through an eval-mode (running-statistics) forward is not implemented; '
                               'the reference only trains in train mode (train.py:377)')
        if d_of is not None and d_of.numel() == 0:
            d_of = None
        bank.set_dout_nchw(ws, d_of.contiguous() if d_of is not None else None,
                           d_raw.contiguous() if d_raw is not None else None)
        bank.backward(ws)
        g = bank.grads.clone()          # fresh storage: autograd may keep / accumulate these views
        out = []
        for (gi, key, p) in model._param_index:
            if not (bank.g0 <= gi < bank.g0 + bank.Ga):
                out.append(None)
            elif model._embedded:          # narrower than the engine: gather the module's block(s) out of the engine-shaped gradient
                ge = bank.grad_view(gi, key, grads=g)
                gm = torch.empty(p.shape, device=ge.device, dtype=ge.dtype)
                for mi, bi in _embed_pieces(bank.lay, key, tuple(p.shape)):
                    gm[mi] = ge[bi]
                out.append(gm)
            else:
                out.append(bank.grad_view(gi, key, grads=g, shape=p.shape))      # (the 1x1 output conv is padded to 4 rows in the bank)
        return (None, None, None, None) + tuple(out)


class _SelfCompleteBase(nn.Module):
    KIND = None

    def _setup(self, features_root, tot_raw_num, tot_of_num, border_mode, rawRange, useFlow, padding, elastic=False):
        assert tot_of_num <= tot_raw_num
        if border_mode == 'predict' or (elastic and border_mode == 'elasticPredict'):
            self.raw_center_idx, self.of_center_idx = tot_raw_num - 1, tot_of_num - 1
        else:
            self.raw_center_idx, self.of_center_idx = (tot_raw_num - 1) // 2, (tot_of_num - 1) // 2
        if rawRange is None:
            self.rawRange = range(tot_raw_num)
        else:
            if rawRange < 0:
                rawRange += tot_raw_num
            assert rawRange < tot_raw_num
            self.rawRange = range(rawRange, rawRange + 1)
        self.raw_channel_num, self.of_channel_num = RAW_C, OF_C
        self.tot_of_num, self.tot_raw_num = tot_of_num, tot_raw_num
        self.raw_of_offset = self.raw_center_idx - self.of_center_idx
        self.useFlow, self.padding = useFlow, padding
        self.features_root = features_root
        assert self.raw_of_offset >= 0
        self._in_ch = RAW_C * tot_raw_num if padding else RAW_C * (tot_raw_num - 1)
        self._bank = None
        self._bank_dirty = True
        self._lambda = (1.0, 1.0)
        self._engine_nf = engine_width(features_root)
        self._embedded = self._engine_nf != features_root
        self._sync_ver = None           # embedded models: (module tensor versions, bank state version) of the last push / pull

    def _make_unet(self, tag, out_ch):
        """Registers inc<tag>, down<tag>{1,2,3} and returns the matching `up` builder (decoder is registered later,
        in the reference's order)."""
        nf = self.features_root
        setattr(self, 'inc%s' % tag, inconv(self._in_ch, nf))
        for k, m in ((1, 1), (2, 2), (3, 4)):
            setattr(self, 'down%s%d' % (tag, k), down(nf * m, nf * m * 2))

    def _make_decoder(self, tag, out_ch):
        nf = self.features_root
        for k, m in ((1, 8), (2, 4), (3, 2)):
            setattr(self, 'up%s%d' % (tag, k), up(nf * m, nf * m // 2))
        setattr(self, 'outc%s' % tag, outconv(nf, out_ch))

    @staticmethod
    def _stems(tag):
        return dict(inc='inc%s' % tag, down=['down%s%d' % (tag, k) for k in (1, 2, 3)],
                    up=['up%s%d' % (tag, k) for k in (1, 2, 3)], outc='outc%s' % tag)

    # ---- bank plumbing -------------------------------------------------------------------------------------
    def _unit_table(self):
        """[(UnitSpec, stems)] for every parameter-owning UNet, in flat-buffer order."""
        raise NotImplementedError

    def _active_window(self, table):
        raise NotImplementedError

    def _apply(self, fn, *a, **k):
        self._sync()                     # (embedded widths) what the bank trained must be in the tensors that are about to move
        r = super()._apply(fn, *a, **k)
        self._bank_dirty = True
        return r

    def set_loss_weights(self, lambda_raw=1.0, lambda_of=1.0):
        self._sync()
        self._lambda = (lambda_raw, lambda_of)
        self._bank_dirty = True

    def bank(self, device=None):
        """The HIP engine, with this module's parameters/buffers living inside its flat buffers."""
        p0 = next(self.parameters())
        device = torch.device(device) if device is not None else p0.device
        if device.type != 'cuda':
            raise L.VecVadHipError('the SelfComplete UNet bank runs on an MI355X only (got device %s); '
                                   'there is no CPU path in vec_vad_amd' % device)
        if self._bank is not None and not self._bank_dirty and self._bank.device == device:
            return self._bank
        table = self._unit_table()
        g0, ga = self._active_window(table)
        old = self._bank
        bank = UNetBank([u for u, _ in table], nf=self._engine_nf, tot_raw_num=self.tot_raw_num,
                        tot_of_num=self.tot_of_num, padding=self.padding, active=(g0, ga), device=device,
                        lambda_raw=self._lambda[0], lambda_of=self._lambda[1])
        index, bindex = [], []
        with torch.no_grad():
            for g, (u, stems) in enumerate(table):
                for key, (off, shape) in bank.lay.p.items():
                    p = self.get_parameter(conv_key_to_state_name(stems, key))
                    if not self._embedded:     # the module tensor IS the bank's block
                        dst = bank.params[g, off:off + p.numel()].view(p.shape)
                        dst.copy_(p.data)
                        p.data = dst
                    index.append((g, key, p))
                for key, (off, shape) in bank.lay.b.items():
                    name = conv_key_to_state_name(stems, key)
                    mod, _, leaf = name.rpartition('.')
                    m = self.get_submodule(mod)
                    if not self._embedded:
                        dst = bank.bufs[g, off:off + shape[0]]
                        dst.copy_(getattr(m, leaf))
                        m._buffers[leaf] = dst
                    bindex.append((g, key, m, leaf))
                    if leaf == 'running_mean':
                        l = int(key.split('.')[0][1:])
                        bank.nbt[g, l] = m.num_batches_tracked.to(device)
                        m._buffers['num_batches_tracked'] = bank.nbt[g, l]
        if old is not None and old.adam_m is not None and old.params.shape == bank.params.shape:
            bank.adam_m, bank.adam_v, bank.adam_t = old.adam_m.to(device), old.adam_v.to(device), old.adam_t
        self._param_index = index
        self._buf_index = bindex
        self._bank = bank
        self._bank_dirty = False
        if self._embedded:
            self._push()
        return bank

    # ---- embedded widths (features_root not 32 / 64): module tensors are copies of blocks of the engine's tensors ------------
    def _tensor_versions(self):
        return (tuple(p._version for (_, _, p) in self._param_index),
                tuple(getattr(m, leaf)._version for (_, _, m, leaf) in self._buf_index))

    def _push(self):
        """module -> bank (after construction, load_state_dict, or an optimizer that wrote the module's tensors)."""
        bank = self._bank
        with torch.no_grad():
            for (g, key, p) in self._param_index:
                off, shape = bank.lay.p[key]
                n = 1
                for d in shape:
                    n *= d
                dst = bank.params[g, off:off + n].view(shape)
                for mi, bi in _embed_pieces(bank.lay, key, tuple(p.shape)):
                    dst[bi] = p.data[mi].to(dst.device)
            for (g, key, m, leaf) in self._buf_index:
                off, shape = bank.lay.b[key]
                src = getattr(m, leaf)
                bank.bufs[g, off:off + src.shape[0]] = src.to(bank.bufs.device)
        bank.mark_dirty()
        self._sync_ver = (self._tensor_versions(), bank._state_ver)

    def _pull(self):
        """bank -> module (after the bank's own Adam / train-mode BatchNorm wrote parameters and running statistics)."""
        bank = self._bank
        with torch.no_grad():
            for (g, key, p) in self._param_index:
                off, shape = bank.lay.p[key]
                n = 1
                for d in shape:
                    n *= d
                src = bank.params[g, off:off + n].view(shape)
                for mi, bi in _embed_pieces(bank.lay, key, tuple(p.shape)):
                    p.data[mi] = src[bi].to(p.device)
            for (g, key, m, leaf) in self._buf_index:
                off, shape = bank.lay.b[key]
                dst = getattr(m, leaf)
                dst.copy_(bank.bufs[g, off:off + dst.shape[0]])
        self._sync_ver = (self._tensor_versions(), bank._state_ver)

    def _sync(self):
        """Embedded models: bring module tensors and bank back together, whichever side was written since the last sync (the
        module wins when both were: an explicit load / optimizer write is the caller's statement of what the model is)."""
        if not self._embedded or self._bank is None or self._bank_dirty:
            return
        tv, sv = self._sync_ver
        if self._tensor_versions() != tv:
            self._push()
        elif self._bank._state_ver != sv:
            self._pull()

    def state_dict(self, *args, **kwargs):
        self._sync()
        return super().state_dict(*args, **kwargs)

    # Embedded models (features_root other than 32 / 64): the module's tensors are COPIES of blocks of the engine's, so after the
    # fused trainer's steps they are stale until something pulls them.  state_dict() / forward() / _apply() do; a direct read through
    # parameters() / buffers() (an external optimizer's constructor, a norm, torch.save of p.data) must see the trained values
    # too.  No-ops for the 32- / 64-wide models, whose parameters are views of the engine's buffers.
    def named_parameters(self, *args, **kwargs):
        if getattr(self, '_embedded', False):
            self._sync()
        return super().named_parameters(*args, **kwargs)

    def named_buffers(self, *args, **kwargs):
        if getattr(self, '_embedded', False):
            self._sync()
        return super().named_buffers(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._sync()
        return r

    def forward(self, x, x_of):
        if not x.is_cuda:
            raise L.VecVadHipError('SelfComplete forward needs CUDA/HIP tensors (the hot path is HIP-only)')
        bank = self.bank(x.device)
        if self._embedded:
            self._sync()
        elif self._param_index[0][2].data_ptr() != bank.params.data_ptr() + 4 * bank.lay.p['c0.w'][0]:
            self._bank_dirty = True      # somebody re-assigned .data; re-adopt the parameters
            bank = self.bank(x.device)
        params = [p for (_, _, p) in self._param_index]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if need_grad:
            of_o, raw_o = _BankFn.apply(x, x_of, self, self.training, *params)
        else:
            ws = bank.set_input_nchw(x, x_of)
            bank.forward(ws, self.training)
            of_o, raw_o = bank.outputs_nchw(ws)
        units = bank.units[bank.g0:bank.g0 + bank.Ga]
        raw_t = torch.cat([x[:, u.tgt * RAW_C:(u.tgt + 1) * RAW_C] for u in units if u.role == 'raw'], dim=1)
        of_units = [u for u in units if u.role == 'of']
        if of_units:
            of_t = torch.cat([x_of[:, u.tgt * OF_C:(u.tgt + 1) * OF_C] for u in of_units], dim=1)
        else:
            of_o, of_t = [], []      # the reference returns empty lists when no flow UNet ran (model/unet.py:263-267)
        return of_o, raw_o, of_t, raw_t


class SelfCompleteNet4(_SelfCompleteBase):
    """5 raw UNets + 1 flow UNet ("5raw1of", reference: model/unet.py:73-267)."""

    def __init__(self, features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None,
                 useFlow=True, padding=True):
        super().__init__()
        self._setup(features_root, tot_raw_num, tot_of_num, border_mode, rawRange, useFlow, padding)
        if tot_raw_num != 5 or (useFlow and tot_of_num != 1):
            raise NotImplementedError('SelfCompleteNet4 is the 5raw+1of bank (train.py:261-268 asserts tot_frame_num == 5)')
        for i in range(5):
            self._make_unet(str(i), RAW_C)
        for i in range(5):
            self._make_decoder(str(i), RAW_C)
        if useFlow:
            self._make_unet('_of', OF_C)
            self._make_decoder('_of', OF_C)

    def _unit_table(self):
        t = [(UnitSpec('raw', i, i), self._stems(str(i))) for i in range(5)]
        if self.useFlow:
            # the single flow UNet runs for raw_i = raw_of_offset + of_i (model/unet.py:247-259)
            t.append((UnitSpec('of', self.raw_of_offset + self.tot_of_num - 1, self.tot_of_num - 1), self._stems('_of')))
        return t

    def _active_window(self, table):
        rr = list(self.rawRange)
        if len(rr) == self.tot_raw_num:
            return (0, len(table))
        r = rr[0]
        has_of = self.useFlow and 0 <= r - self.raw_of_offset < self.tot_of_num
        if has_of and r != 4:
            raise NotImplementedError('rawRange with a flow UNet is only contiguous for the last frame')
        return (r, 2 if has_of else 1)


class SelfCompleteNetFull(_SelfCompleteBase):
    """5 raw + 5 flow UNets ("5raw5of", reference: model/unet.py:270-556)."""

    def __init__(self, features_root=32, tot_raw_num=5, tot_of_num=5, border_mode='predict', rawRange=None,
                 useFlow=True, padding=True):
        super().__init__()
        self._setup(features_root, tot_raw_num, tot_of_num, border_mode, rawRange, useFlow, padding, elastic=True)
        for i in range(5):
            self._make_unet(str(i), RAW_C)
        for i in range(5):
            self._make_decoder(str(i), RAW_C)
        if useFlow:
            for i in range(5):
                self._make_unet('_of%d' % i, OF_C)
            for i in range(5):
                self._make_decoder('_of%d' % i, OF_C)

    def _unit_table(self):
        # flat order interleaves (raw_i, of_i) so that a rawRange window is contiguous
        t = []
        for i in range(5):
            t.append((UnitSpec('raw', i, i), self._stems(str(i))))
            of_i = i - self.raw_of_offset
            if self.useFlow and 0 <= of_i < self.tot_of_num:
                t.append((UnitSpec('of', i, of_i), self._stems('_of%d' % of_i)))
        if self.useFlow:   # flow UNets that can never run still own parameters
            used = {s['inc'] for _, s in t}
            for j in range(5):
                if 'inc_of%d' % j not in used:
                    t.append((UnitSpec('of', 0, j), self._stems('_of%d' % j)))
        return t

    def _active_window(self, table):
        runnable = [k for k, (u, s) in enumerate(table)
                    if u.role == 'raw' or 0 <= u.erase - self.raw_of_offset < self.tot_of_num]
        n_run = 5 + sum(1 for i in range(5) if self.useFlow and 0 <= i - self.raw_of_offset < self.tot_of_num)
        rr = list(self.rawRange)
        if len(rr) == self.tot_raw_num:
            return (0, n_run)
        r = rr[0]
        start = next(k for k, (u, s) in enumerate(table) if u.role == 'raw' and u.erase == r)
        has_of = self.useFlow and 0 <= r - self.raw_of_offset < self.tot_of_num
        return (start, 2 if has_of else 1)


class SelfCompleteNet1raw1of(_SelfCompleteBase):
    """single raw + single flow UNet predicting the last frame (reference: model/unet.py:559-652)."""

    def __init__(self, features_root=64, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None,
                 useFlow=True, padding=True):
        super().__init__()
        self._setup(features_root, tot_raw_num, tot_of_num, border_mode, rawRange, useFlow, padding)
        nf = features_root
        self.inc = inconv(self._in_ch, nf)
        self.down1, self.down2, self.down3 = down(nf, nf * 2), down(nf * 2, nf * 4), down(nf * 4, nf * 8)
        self.up1, self.up2, self.up3 = up(nf * 8, nf * 4), up(nf * 4, nf * 2), up(nf * 2, nf)
        self.outc = outconv(nf, RAW_C)
        if useFlow:
            self.inc_of = inconv(self._in_ch, nf)
            self.down_of1, self.down_of2, self.down_of3 = down(nf, nf * 2), down(nf * 2, nf * 4), down(nf * 4, nf * 8)
            self.up_of1, self.up_of2, self.up_of3 = up(nf * 8, nf * 4), up(nf * 4, nf * 2), up(nf * 2, nf)
            self.outc_of = outconv(nf, OF_C)

    def _unit_table(self):
        e = self.tot_raw_num - 1
        t = [(UnitSpec('raw', e, e), dict(inc='inc', down=['down1', 'down2', 'down3'], up=['up1', 'up2', 'up3'], outc='outc'))]
        if self.useFlow:
            t.append((UnitSpec('of', e, self.tot_raw_num - 1 - self.raw_of_offset),
                      dict(inc='inc_of', down=['down_of1', 'down_of2', 'down_of3'],
                           up=['up_of1', 'up_of2', 'up_of3'], outc='outc_of')))
        return t

    def _active_window(self, table):
        return (0, len(table))
