"""the build's test.py reads artifacts WRITTEN BY THE REFERENCE (tests/golden/ref_written/*, produced by
tests/golden/make_ref_checkpoint.py from the imported reference classes through the statement sequence of train.py:262-436).
CPU only: loading a file needs the host-side module surface, not the HIP bank (features_root = 4 keeps the fixtures at 3 MB)."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_written')
FG, METHOD, NF = 'obj', 'SelfComplete', 4


def _build(kind):
    from model.unet import SelfCompleteNet4, SelfCompleteNetFull
    cls, tot_of = {'net4': (SelfCompleteNet4, 1), 'full': (SelfCompleteNetFull, 5)}[kind]
    return lambda: cls(features_root=NF, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=None, useFlow=True,
                       padding=False)


def test_raw_file_layout_is_what_train_py_writes():
    """keys / prefix / nesting / dtypes of the reference-written model file (train.py:274,375,410,436)."""
    w = torch.load(os.path.join(REF, 'UCSDped2_model_%s_%s.npy' % (FG, METHOD)), map_location='cpu', weights_only=False)
    assert isinstance(w, list) and len(w) == 1 and len(w[0]) == 1 and len(w[0][0]) == 1          # [h][w] -> [state_dict]
    sd = w[0][0][0]
    assert all(k.startswith('module.') for k in sd)
    ours = _build('net4')().state_dict()
    assert set(k[len('module.'):] for k in sd) == set(ours.keys())
    for k, v in ours.items():
        r = sd['module.' + k]
        assert r.shape == v.shape and r.dtype == v.dtype, k
    nbt = [v for k, v in sd.items() if k.endswith('num_batches_tracked')]
    assert nbt and all(v.dtype == torch.int64 and int(v) == 6 for v in nbt)                       # 2 epochs x 3 batches
    s = torch.load(os.path.join(REF, 'UCSDped2_raw_training_scores_%s_%s.npy' % (FG, METHOD)), weights_only=False)
    assert isinstance(s[0][0], np.ndarray) and s[0][0].dtype == np.float32 and s[0][0].shape == (12,)


@pytest.mark.parametrize('ds,kind,shanghai', [('UCSDped2', 'net4', False), ('ShanghaiTech', 'full', True)])
def test_load_artifacts_reads_reference_written_files(ds, kind, shanghai):
    """test.py's own loader on the reference's files: module surface loads every tensor bit for bit, the statistics are the
    population mean / std of the stored scores (test.py:264-266), and the oracle evaluated on the LOADED weights reproduces the
    training scores the reference stored next to them (train.py:413-427) -- file -> loader -> forward closes on reference data."""
    import test as S
    base = os.path.join(REF, ds + '_')
    net_set, st_r, st_o = S.load_artifacts(base, FG, METHOD, shanghai, _build(kind), 'cpu')
    w = torch.load(base + 'model_%s_%s.npy' % (FG, METHOD), map_location='cpu', weights_only=False)
    raw_tr = torch.load(base + 'raw_training_scores_%s_%s.npy' % (FG, METHOD), weights_only=False)
    of_tr = torch.load(base + 'of_training_scores_%s_%s.npy' % (FG, METHOD), weights_only=False)
    scenes = range(len(w)) if shanghai else [None]
    for s in scenes:
        nets = net_set[s][0][0] if shanghai else net_set[0][0]
        sd_ref = (w[s][0][0] if shanghai else w[0][0])[0]
        r_ref, o_ref = (raw_tr[s][0][0], of_tr[s][0][0]) if shanghai else (raw_tr[0][0], of_tr[0][0])
        sr, so = (st_r[s][0][0], st_o[s][0][0]) if shanghai else (st_r[0][0], st_o[0][0])
        assert len(nets) == 1 and not nets[0].training
        got = nets[0].state_dict()
        for k, v in got.items():
            assert torch.equal(v, sd_ref['module.' + k]), k
        assert sr == (np.mean(r_ref), np.std(r_ref)) and so == (np.mean(o_ref), np.std(o_ref))
    # the forward on the loaded weights: the reference re-wraps ONE network object for every scene (train.py:261-265,290), so every
    # scene's saved dict aliases the final weights -> only the LAST scene's stored scores belong to the weights in the file
    last = len(w) - 1 if shanghai else None
    tot_of = 5 if kind == 'full' else 1
    raw, flow = O.seeded_cubes(8 if shanghai else 12, tot_of, (30 + last) if shanghai else 21)
    x, x_of = O.cubes_to_inputs(raw, flow)
    nets = net_set[last][0][0] if shanghai else net_set[0][0]
    sd = {k: v.clone() for k, v in nets[0].state_dict().items()}
    rs, os_ = O.score_pass(sd, O.bank_spec(kind), x, x_of, 4)
    r_ref, o_ref = (raw_tr[last][0][0], of_tr[last][0][0]) if shanghai else (raw_tr[0][0], of_tr[0][0])
    np.testing.assert_allclose(rs, r_ref, rtol=1e-5)
    np.testing.assert_allclose(os_, o_ref, rtol=1e-5)
    if shanghai:      # the aliasing itself: all scenes carry the same tensors
        a, b = w[0][0][0][0], w[last][0][0][0]
        assert all(torch.equal(a[k], b[k]) for k in a)


def test_saved_by_build_loads_like_reference_file(tmp_path):
    """The other direction: what train.py of the build saves ('module.' keys, [[[sd]]]) has the reference file's key set and dtypes."""
    ref = torch.load(os.path.join(REF, 'UCSDped2_model_%s_%s.npy' % (FG, METHOD)), map_location='cpu', weights_only=False)[0][0][0]
    net = _build('net4')()
    sd = {('module.' + k): v.detach().clone() for k, v in net.state_dict().items()}          # train.train_block's return value
    torch.save([[[sd]]], str(tmp_path / 'm.npy'))
    back = torch.load(str(tmp_path / 'm.npy'), map_location='cpu', weights_only=False)[0][0][0]
    assert list(back.keys()) == list(ref.keys())
    assert all(back[k].dtype == ref[k].dtype and back[k].shape == ref[k].shape for k in ref)


@pytest.mark.gpu
@pytest.mark.parametrize('ds,kind,shanghai', [('UCSDped2', 'net4', False), ('ShanghaiTech', 'full', True)])
def test_reference_written_checkpoint_scored_by_the_hip_bank(ds, kind, shanghai):
    """SURVEY 8(f-4) closed on the GPU: the model file the REFERENCE's classes wrote (features_root = 4) is loaded by test.py's
    loader onto the MI355X and the cubes the reference scored after training (train.py:413-427) go through test.py's own device
    scoring path (FusedTrainer.score_cubes: cube gather + folded eval forward + fused per-cube score sums, all HIP; the 4-wide
    model runs embedded in the 32-wide engine, vec_vad_amd/unet.py engine_width) -- the scores must be the ones the reference
    stored next to the weights, at the north-star bar (1e-3; observed ~1e-6)."""
    import test as S
    from vec_vad_amd.trainer import FusedTrainer
    base = os.path.join(REF, ds + '_')
    net_set, st_r, st_o = S.load_artifacts(base, FG, METHOD, shanghai, _build(kind), 'cuda')
    raw_tr = torch.load(base + 'raw_training_scores_%s_%s.npy' % (FG, METHOD), weights_only=False)
    of_tr = torch.load(base + 'of_training_scores_%s_%s.npy' % (FG, METHOD), weights_only=False)
    last = len(raw_tr) - 1 if shanghai else None      # every scene's dict aliases the final weights (see the CPU test above)
    tot_of = 5 if kind == 'full' else 1
    raw, flow = O.seeded_cubes(8 if shanghai else 12, tot_of, (30 + last) if shanghai else 21)
    net = (net_set[last][0][0] if shanghai else net_set[0][0])[0]
    assert net._embedded and net._engine_nf == 32 and next(net.parameters()).is_cuda and not net.training
    tr = FusedTrainer(net, reset_optimizer=False)
    cubes = [raw[:5], raw[5:5], raw[5:]]              # three "frames", one of them empty (test.py:276)
    flows = [flow[:5], flow[5:5], flow[5:]]
    r, o = S.score_cubes_device(tr, cubes, flows, score_batch=4)
    r_ref, o_ref = (raw_tr[last][0][0], of_tr[last][0][0]) if shanghai else (raw_tr[0][0], of_tr[0][0])
    np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-3)
    np.testing.assert_allclose(o.cpu().numpy(), o_ref, rtol=1e-3)
    from _util import observe
    observe('ref_written_%s_on_hip' % ds, raw_rel=float(np.abs(r.cpu().numpy() / r_ref - 1).max()),
            of_rel=float(np.abs(o.cpu().numpy() / o_ref - 1).max()))
