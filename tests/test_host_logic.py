"""CPU tests: C-ABI library loads and exports every declared symbol, host-side adapters / evaluation helpers,
layout bookkeeping and the multi-process gradient bucket logic (gloo, world_size 2)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_mirrors_match_the_compiled_structs():
    """vv_abi_sizeof: every ctypes mirror in _lib.py has the size of the struct the library was compiled with (vv_conv_params grew
    at its end in round 4), an unknown index answers -1, and lib() itself refuses a library whose structs differ."""
    import ctypes as C
    from vec_vad_amd import _lib
    l = _lib.lib()
    for which, cls in enumerate(_lib.ABI_STRUCTS):
        assert l.vv_abi_sizeof(which) == C.sizeof(cls), cls.__name__
    assert l.vv_abi_sizeof(len(_lib.ABI_STRUCTS)) == -1 and l.vv_abi_sizeof(-1) == -1
    assert C.sizeof(_lib.ConvParams) % 8 == 0 and _lib.ConvParams.out1.offset > _lib.ConvParams.bn_partial.offset


def test_library_exports_every_declared_symbol():
    from vec_vad_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'vecvad_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|int64_t|void|const char\*)\s+(vv_\w+)\s*\(', hdr, flags=re.M))
    assert declared, 'no declarations parsed'
    l = _lib.lib()                      # raises if the .so is missing or a bound symbol is absent
    for name in declared:
        assert hasattr(l, name), name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b'gfx950' in l.vv_version()
    # pure host helpers of the ABI (no GPU needed)
    assert l.vv_conv_ntiles(256, 32, 32) == 1024 and l.vv_conv_ntiles(5, 4, 4) == 1
    import ctypes as C
    oc, oh, ow = C.c_int32(), C.c_int32(), C.c_int32()
    assert l.vv_correlation_out_shape(256, 56, 128, 20, 1, 20, 1, 2, C.byref(oc), C.byref(oh), C.byref(ow)) == 0
    assert (oc.value, oh.value, ow.value) == (441, 56, 128)       # correlation_cuda.c:25-34


def test_no_cpu_fallback():
    from model.unet import SelfCompleteNet4
    from vec_vad_amd._lib import VecVadHipError
    net = SelfCompleteNet4(padding=False)
    with pytest.raises(VecVadHipError):
        net(torch.zeros(2, 15, 32, 32), torch.zeros(2, 2, 32, 32))
    with pytest.raises(RuntimeError):
        net.inc0(torch.zeros(1, 12, 32, 32))


def test_module_surface_matches_reference_state_dict():
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4, SelfCompleteNetFull, SelfCompleteNet1raw1of
    for kind, cls, kw in (('net4', SelfCompleteNet4, {}), ('full', SelfCompleteNetFull, {}),
                          ('1raw1of', SelfCompleteNet1raw1of, dict(features_root=32))):
        for padding in (False, True):
            net = cls(padding=padding, **kw)
            ref = O.seeded_state_dict(kind, nf=32, padding=padding)
            sd = net.state_dict()
            assert set(sd) == set(ref)
            assert all(sd[k].shape == ref[k].shape for k in ref)
            net.load_state_dict(ref)
    net = SelfCompleteNet4(padding=False)
    assert sum(p.numel() for p in net.parameters()) == 12876657
    assert sum(p.numel() for p in SelfCompleteNetFull(padding=False).parameters()) == 21460985


def test_cube_adapter_matches_reference_layout():
    from oracle import unet_oracle as O
    from vad_datasets import cube_to_train_dataset
    from _util import digest, digest_close, load_golden
    g = load_golden('net4_nf32_nopad')
    raw, flow = O.seeded_cubes(6, 1, 0)
    ds = cube_to_train_dataset(raw, target=flow[:, 0])
    x = torch.stack([ds[i][0] for i in range(6)])
    x_of = torch.stack([ds[i][1] for i in range(6)])
    assert digest_close(digest(x), g['x_digest'], 1e-6) and digest_close(digest(x_of), g['xof_digest'], 1e-6)
    assert torch.equal(torch.stack([ds[i][2] for i in range(6)]), x)
    xo, _ = O.cubes_to_inputs(raw, flow)
    assert torch.equal(x, xo)
    gf = load_golden('full_nf32_nopad')
    raw, flow = O.seeded_cubes(4, 5, 0)
    ds = cube_to_train_dataset(raw, target=flow)
    assert digest_close(digest(torch.stack([ds[i][1] for i in range(4)])), gf['xof_digest'], 1e-6)


def test_block_idx_and_auc():
    from utils import calc_block_idx, frame_roc_auc
    assert calc_block_idx(10, 50, 20, 80, 240.0, 360.0, 1) == [(0, 0)]
    cells = calc_block_idx(170, 190, 110, 130, 120.0, 180.0, 9)
    assert set(cells) == {(0, 0), (0, 1), (1, 0), (1, 1)}
    s = np.array([0.1, 0.4, 0.35, 0.8, 0.4])
    y = np.array([0, 0, 1, 1, 1])
    from oracle import unet_oracle as O
    assert abs(frame_roc_auc(s, y) - O.roc_auc(s, y)) < 1e-12
    try:
        from sklearn.metrics import roc_auc_score
        assert abs(frame_roc_auc(s, y) - roc_auc_score(y, s)) < 1e-12
    except ImportError:
        pass


def test_paint_frame_semantics():
    import test as S
    m = S.paint_frame(np.array([1.5, -2.0]), np.array([[10.2, 20.7, 30.0, 40.0], [0, 0, 400, 300]]), 240, 360)
    assert m[21, 11] == 1.5 and m[20, 11] == -2.0 and m[0, 0] == -2.0 and m.max() == 1.5
    e = S.paint_frame(np.zeros(0), np.zeros((0, 4)), 240, 360)
    assert e.max() == -100000


def test_bank_layout_and_plan_construction():
    from vec_vad_amd.bank import UNetBank, UnitSpec, BankLayout
    lay = BankLayout(32, 12)
    assert sum(int(np.prod(s)) for k, (o, s) in lay.p.items() if not k.startswith('o.')) + 3 * 32 + 3 == 2146115
    units = [UnitSpec('raw', i, i) for i in range(5)] + [UnitSpec('of', 4, 0)]
    b = UNetBank(units, nf=32, device='cpu')
    ws = b.workspace(5)
    fl = [c[2] for c in ws.fwd[True].calls]
    packs = [x for x in fl if x.startswith('pack')]
    # the weight panels: the first two conv layers on the main stream, the rest as a side branch that joins in front of conv2
    assert packs == (['pack_wino', 'pack_tail', 'pack_wino_tail'] if b.wino else ['pack', 'pack_tail']) and fl[:len(packs)] == packs
    meta = dict(zip(fl, ws.fwd[True].meta))
    assert meta['pack_tail'][0] == 1 and 'pack_tail' in meta['conv2'][1] and meta['conv1'][1] == ()
    assert len(fl) - len(packs) == 36                    # cube_erase, 14 x (conv + bn), 3 pool, 3 convT, 1x1 out
    ws.bwd = b._plan_backward(ws, 5)
    labels = [c[2] for c in ws.bwd.calls]
    # decoder bucket is complete before the encoder; round 6: the transposed conv's data gradient leaves layer 7's BatchNorm-backward
    # sums in its epilogue (fp32 path), so that layer has no separate reduce pass any more
    assert labels.index('dgradT0') < labels.index('bn_bwd_apply7') and (('bn_bwd_reduce7' not in labels) == (b.wino and b.fuse_bn_sums))
    assert [x for x in labels if x.startswith('bn_bwd_reduce')] == (['bn_bwd_reduce5', 'bn_bwd_reduce3', 'bn_bwd_reduce1'] if (b.wino and b.fuse_bn_sums) else
                                                                     [x for x in labels if x.startswith('bn_bwd_reduce')])
    assert b.chmap[2].tolist()[:12] == [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14]   # frame 2 erased (model/unet.py:183)


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vec_vad_amd.trainer import GradBuckets, shard_batch
    g = torch.arange(6 * 10, dtype=torch.float32) * (rank + 1)          # bucket-major flat buffer: [6][4] then [6][6]
    b = GradBuckets(g, 6, [0, 4, 10], dist.group.WORLD)
    b.launch(1)
    b.launch(0)
    b.finish()
    idx = shard_batch(torch.arange(8), rank, world)
    q.put((rank, g.numpy().copy(), idx.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    expect = torch.arange(60, dtype=torch.float32) * 3
    assert np.array_equal(res[0][1], expect.numpy()) and np.array_equal(res[1][1], expect.numpy())
    assert res[0][2].tolist() == [0, 1, 2, 3] and res[1][2].tolist() == [4, 5, 6, 7]


def _world8_worker(rank, world, port, q):
    """One rank of an 8-rank group on the REAL bucket geometry of the Net4 bank (BankLayout(32, 12): G = 6, bounds
    [0, c4.w, c8.w, U]) + the even shard and DataParallel's uneven last-batch scatter of train.train_block."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vec_vad_amd.bank import BankLayout
    from vec_vad_amd.trainer import GradBuckets, shard_batch
    lay = BankLayout(32, 12)
    G = 6
    bounds = [0, lay.p['c4.w'][0], lay.p['c8.w'][0], lay.U]
    # bucket-major buffer; element value = rank-dependent so that the sum identifies every contribution
    n = G * lay.U
    g = torch.full((n,), float(rank + 1))
    g[::1000] += torch.arange(0, n, 1000, dtype=torch.float32) * 1e-3
    b = GradBuckets(g, G, bounds, dist.group.WORLD)
    for k in (2, 1, 0):                         # the order the backward pass completes them
        b.launch(k)
    b.finish()
    even = shard_batch(torch.arange(256), rank, world)
    # the uneven last batch (train.train_block): chunks of ceil(n / world); trailing ranks may get nothing
    out = {}
    for n_glob in (250, 9, 3):
        per = -(-n_glob // world)
        out[n_glob] = torch.arange(n_glob)[rank * per:(rank + 1) * per].tolist()
    try:
        shard_batch(torch.arange(250), rank, world)
        raised = False
    except ValueError:
        raised = True
    q.put((rank, float(g[1]), float(g[1000]), float(g[-1]), [v.numel() for v in b.views], even.tolist(), out, raised, bounds, G * lay.U))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_and_sharding_gloo_world8():
    """BASELINE configs[2]: 8 ranks.  The three in-place bucket all-reduces cover the whole gradient buffer of the real Net4
    geometry exactly once, a 256-cube batch shards 32 per rank, and the last partial batch follows DataParallel's scatter
    (chunks of ceil(n / 8), empty shards for trailing ranks) and covers every cube exactly once."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 1000)
    W = 8
    ps = [ctx.Process(target=_world8_worker, args=(r, W, port, q)) for r in range(W)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    tot = sum(range(1, W + 1))
    bounds, n = res[0][8], res[0][9]
    assert bounds[0] == 0 and bounds == sorted(bounds) and 6 * bounds[-1] == n
    for r, g1, g1000, glast, widths, even, uneven, raised, _, _ in res:
        assert g1 == tot and glast == tot and abs(g1000 - (tot + W * 1.0)) < 1e-4      # every element summed over all 8 ranks once
        assert sum(widths) == n and widths == [6 * (b - a) for a, b in zip(bounds[:-1], bounds[1:])]
        assert even == list(range(32 * r, 32 * r + 32)) and raised
    for n_glob in (250, 9, 3):
        got = [i for r in res for i in r[6][n_glob]]
        assert got == list(range(n_glob))                                                # every cube once, in order
    assert [len(r[6][3]) for r in res] == [1, 1, 1, 0, 0, 0, 0, 0]                        # ranks without cubes
    assert [len(r[6][9]) for r in res] == [2, 2, 2, 2, 1, 0, 0, 0]


def _extract_worker(rank, world, port, root, fail, q):
    import time
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import foreground
    import train
    c = {'data_root_dir': root, 'modality': 'raw2flow', 'dataset_name': 'UCSDped2', 'mode_fg': 'detection'}
    probe = os.path.join(root, 'raw2flow', 'UCSDped2_foreground_train_detection-raw.npy')

    def fake_extract(c, device):
        time.sleep(1.5)                          # the other rank is already waiting on the side group by now
        if fail:
            raise ValueError('no frames')
        os.makedirs(os.path.dirname(probe), exist_ok=True)
        foreground.save_nested(probe, [[np.zeros((2, 3))]], 2)
    foreground.extract_train = fake_extract
    t0 = time.time()
    try:
        train._extract_once(c, 'cpu', dist, timeout_h=0.05)
        q.put((rank, 'ok', os.path.exists(probe), time.time() - t0))
    except Exception as e:
        q.put((rank, type(e).__name__, os.path.exists(probe), time.time() - t0))
    dist.destroy_process_group()


@pytest.mark.parametrize('fail', [False, True])
def test_extract_once_other_ranks_wait_on_a_side_group(tmp_path, fail):
    """train._extract_once under torchrun: rank 0 cuts the cubes, rank 1 waits on a gloo side group (no marker file that a
    crashed or restarted job could leave behind) and only continues when the atomically written file is there; a failure of
    rank 0 reaches rank 1 at once instead of after the timeout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000) + (7 if fail else 0)
    ps = [ctx.Process(target=_extract_worker, args=(r, 2, port, str(tmp_path), fail, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps])
    for p in ps:
        p.join(60)
    if fail:
        assert res[0][1] == 'ValueError' and res[1][1] == 'RuntimeError' and not res[1][2]
    else:
        assert res[0][1] == 'ok' and res[1][1] == 'ok' and res[0][2] and res[1][2]
        assert res[1][3] >= 1.0                  # rank 1 really waited for rank 0
    assert not [f for f in os.listdir(tmp_path) if f.startswith('.')]          # nothing left behind


def test_context_range_matches_reference_golden():
    """vad_datasets.context_range vs the real reference method on 459 synthetic video layouts (incl. the 30 layouts on
    which the reference raises NotImplementedError); fixture made by tests/golden/make_context_range_golden.py."""
    import json
    import vad_datasets
    cases = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'context_range.json')))
    assert len(cases) > 400
    n_err = 0
    for lens, mode, ctx, indice, expect in cases:
        vid = [v for v, n in enumerate(lens) for _ in range(n)]
        if expect == [-1]:
            n_err += 1
            with pytest.raises(NotImplementedError):
                vad_datasets.context_range(indice, mode, ctx, len(vid), vid)
        else:
            assert vad_datasets.context_range(indice, mode, ctx, len(vid), vid) == expect, (lens, mode, ctx, indice)
    assert n_err >= 10


def test_frame_indexers_on_a_synthetic_tree(tmp_path):
    """ped / avenue / shanghaiTech directory layouts (reference vad_datasets.py:203-275, 432-484, 645-715): frame order,
    per-video indices, context stacks, ground truth -- on tiny lossless frames written here."""
    from PIL import Image
    import vad_datasets as V
    rng = np.random.default_rng(0)

    def put(path, arr):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(arr).save(path)

    # ---- UCSDped2-like: Test001 (3 frames), Test001_gt, Test002 (2 frames), Test002_gt
    root = str(tmp_path / 'UCSDped2')
    frames = {}
    for v, n in (('Test001', 3), ('Test002', 2)):
        for k in range(n):
            g = rng.integers(0, 256, (6, 8), dtype=np.uint8)
            frames[(v, k)] = g
            put(os.path.join(root, 'Test', v, '%03d.tif' % (k + 1)), g)
            put(os.path.join(root, 'Test', v + '_gt', '%03d.bmp' % (k + 1)), (g > 128).astype(np.uint8) * 255)
    ds = V.unified_dataset_interface('UCSDped2', root, mode='test', context_frame_num=1, border_mode='hard')
    assert isinstance(ds, V.ped_dataset) and len(ds) == 5 and (ds.h, ds.w) == (240, 360)
    assert ds.frame_video_idx == [1, 1, 1, 2, 2] and ds.return_gt and len(ds.all_gt_addr) == 5
    img, gt = ds[3]                       # first frame of video 2: context [3,3,4] (video-1 frame dropped, first repeated)
    assert ds.context_range(3) == [3, 3, 4]
    assert img.shape == (3, 3, 6, 8) and img.dtype == torch.uint8
    assert np.array_equal(img[0, 0].numpy(), frames[('Test002', 0)])          # gray replicated to 3 (BGR) channels
    assert np.array_equal(img[0, 1].numpy(), frames[('Test002', 0)]) and np.array_equal(img[2, 2].numpy(), frames[('Test002', 1)])
    assert np.array_equal(gt.numpy(), (frames[('Test002', 0)] > 128).astype(np.uint8) * 255)
    ds0 = V.ped_dataset(dir=root, mode='test', context_frame_num=0)
    assert ds0[1][0].shape == (3, 6, 8) and ds0.file_format == '.tif'

    # ---- optical-flow tree of the same videos (.npy [h,w,2] float32), train split
    root2 = str(tmp_path / 'optical_flow' / 'UCSDped2')
    for k in range(4):
        os.makedirs(os.path.join(root2, 'Train', 'Train001'), exist_ok=True)
        np.save(os.path.join(root2, 'Train', 'Train001', '%03d.npy' % k), np.full((6, 8, 2), k, np.float32))
    os.makedirs(os.path.join(root2, 'Train', 'notes'), exist_ok=True)          # ignored: no 'Train' in the name
    dsf = V.unified_dataset_interface('UCSDped2', root2, mode='train', context_frame_num=2, border_mode='predict',
                                      file_format='.npy')
    assert len(dsf) == 4 and list(dsf.videos) == ['Train001']
    x, z = dsf[0]
    assert x.shape == (3, 2, 6, 8) and x.dtype == torch.float32 and z.shape == (1,)
    assert x[:, 0, 0, 0].tolist() == [0, 0, 0] and dsf[3][0][:, 0, 0, 0].tolist() == [1, 2, 3]

    # ---- ShanghaiTech-like test split (two parts, scene from the folder name, frame-level gt)
    root3 = str(tmp_path / 'ShanghaiTech')
    for part, v, n in ((1, '01_0014', 2), (2, '03_0031', 3)):
        for k in range(n):
            put(os.path.join(root3, 'Testing', 'frames_part%d' % part, v, '%03d.png' % k),
                rng.integers(0, 256, (4, 4, 3), dtype=np.uint8))
    os.makedirs(os.path.join(root3, 'Testing', 'test_frame_mask'))
    np.save(os.path.join(root3, 'Testing', 'test_frame_mask', '01_0014.npy'), np.array([0, 1]))
    np.save(os.path.join(root3, 'Testing', 'test_frame_mask', '03_0031.npy'), np.array([0, 0, 1]))
    dss = V.unified_dataset_interface('ShanghaiTech', root3, mode='test', file_format='.png')
    assert len(dss) == 5 and dss.save_scene_idx == [1, 1, 3, 3, 3] and dss.scene_idx == [1] * 5 and dss.scene_num == 1
    assert [int(dss[i][1][0]) for i in range(5)] == [0, 1, 0, 0, 1]
    with pytest.raises(NotImplementedError):
        V.unified_dataset_interface('UCF', root3)


def test_read_config_accepts_both_layouts(tmp_path):
    """train.read_config on the shipped config.cfg (shared values in [DEFAULT]) and on a file written the way the reference
    writes its own (every key repeated inside the dataset section, `key=value` without spaces): same effective values."""
    import train as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = T.read_config(os.path.join(root, 'config.cfg'))
    assert (c['dataset_name'], c['mode_fg'], c['modality'], c['method']) == ('UCSDped2', 'obj_det_with_motion', 'raw2flow', 'SelfComplete')
    assert (c['h_block'], c['w_block'], c['tot_frame_num'], c['tot_of_num'], c['rawRange']) == (1, 1, 5, 5, None)
    assert c['cp'].getint('ShanghaiTech', 'saveSegNum') == 40000 and c['cp'].getint('avenue', 'patch_size') == 32
    assert c['precision'] == 'fp32'                     # [mi355x] precision: the shipped default is the reference's arithmetic
    mixed = tmp_path / 'mixed.cfg'
    mixed.write_text(open(os.path.join(root, 'config.cfg')).read().replace('precision = fp32', 'precision = BF16'))
    assert T.read_config(str(mixed))['precision'] == 'bf16'
    flat = '''[shared_parameters]
dataset_name = avenue
raw_dataset_dir = raw_datasets
foreground_extraction_mode = obj_det
data_root_dir = data
modality = raw2flow
method = SelfComplete
[avenue]
patch_size=32
h_block=2
w_block=3
train_bbox_saved = True
train_foreground_saved = True
test_bbox_saved = True
test_foreground_saved = True
scores_saved = False
train_block_mode = 1
test_block_mode = 1
motionThr = 0
[SelfComplete]
border_mode = predict
epochs = 3
batch_size = 64
nf = 32
useFlow = True
context_frame_num = 4
context_of_num = 0
rawRange = 3
padding = True
lambda_raw = 1.0
lambda_of = 2.0
w_raw =1
w_of =0.5
'''
    p = tmp_path / 'flat.cfg'
    p.write_text(flat)
    c = T.read_config(str(p))
    assert (c['dataset_name'], c['h_block'], c['w_block'], c['tot_of_num'], c['rawRange']) == ('avenue', 2, 3, 1, 3)
    assert c['padding'] is True and c['lambda_of'] == 2.0 and c['w_of'] == 0.5 and c['shuffle_seed'] == 0 and c['score_batch'] == 2048


def test_graph_cache_keeps_two_cube_stores():
    """a loop that alternates a training and a validation store must keep replaying both (the cache pins the two most
    recently used stores); a third store evicts the least recently used one; release_graphs(keep_store=...) drops the others."""
    from vec_vad_amd.trainer import FusedTrainer
    tr = FusedTrainer.__new__(FusedTrainer)
    tr._graphs = {}
    A, Bs, Cs = (100, 101), (200, 201), (300, 301)
    key = lambda kind, B, st: (kind, B) + st + (1e-3,)
    for st in (A, Bs, A, Bs):
        for kind in ('train', 'eval'):
            k = key(kind, 32, st)
            if tr._graph_lookup(k) is None:
                tr._graphs[k] = 'cap'
    assert len(tr._graphs) == 4 and all(tr._graph_lookup(key(kd, 32, st)) == 'cap' for st in (A, Bs) for kd in ('train', 'eval'))
    assert tr._graph_lookup(key('train', 32, A)) == 'cap'          # A is now the most recently used
    assert tr._graph_lookup(key('train', 32, Cs)) is None           # third store: evicts B (least recently used), keeps A
    tr._graphs[key('train', 32, Cs)] = 'cap'
    assert {k[2:4] for k in tr._graphs} == {A, Cs}
    tr.release_graphs(keep_store=Cs)
    assert {k[2:4] for k in tr._graphs} == {Cs} and tr._store_lru == [Cs]
    tr.release_graphs()
    assert tr._graphs == {} and tr._store_lru == []


def test_embedded_width_layout_roundtrip():
    """features_root outside {32, 64}: every module tensor maps to block(s) of the engine tensor and back (concat layers: the
    upsampled half starts at the ENGINE's half width), engine_width picks 32 / 64 and refuses widths above 64."""
    import torch
    from vec_vad_amd import _lib as L
    from vec_vad_amd.bank import BankLayout, conv_key_to_state_name
    from vec_vad_amd.unet import SelfCompleteNet4, _embed_pieces, engine_width
    assert [engine_width(n) for n in (1, 4, 31, 32, 33, 48, 64)] == [32, 32, 32, 32, 64, 64, 64]
    with pytest.raises(L.VecVadHipError):
        engine_width(96)
    for nf in (4, 20, 48):
        net = SelfCompleteNet4(features_root=nf, padding=False)
        lay = BankLayout(engine_width(nf), 12)
        stems = net._stems('_of')
        for key, (off, shape) in lay.p.items():
            p = net.get_parameter(conv_key_to_state_name(stems, key))
            dst = torch.zeros(shape)
            for mi, bi in _embed_pieces(lay, key, tuple(p.shape)):
                dst[bi] = p.data[mi]
            back = torch.empty(p.shape)
            for mi, bi in _embed_pieces(lay, key, tuple(p.shape)):
                back[mi] = dst[bi]
            assert torch.equal(back, p.data), key
            assert int((dst != 0).sum()) == int((p.data != 0).sum()), key
        l12 = lay.convs[12]
        w = net.get_parameter(conv_key_to_state_name(stems, 'c12.w'))
        (m0, b0), (m1, b1) = _embed_pieces(lay, 'c12.w', tuple(w.shape))
        assert b1[1].start == l12.cin // 2 and m1[1].start == nf and b0[1] == slice(0, nf)


def test_wino44_routing_policy(monkeypatch):
    """VV_WINO44 (round 5): default 'dgrad' = the data-gradient launches of the measured policy (GEMM-K >= 64 and workgroups that fill the
    chip evenly -- at B = 256 the 16x16-level 64- / 128-channel launches and dgrad8), forward launches stay on F(2x2); '1' adds the forward
    launches; '0' none; 'all' every launch; small batches route nothing."""
    from vec_vad_amd.bank import UNetBank, UnitSpec
    units = [UnitSpec('raw', i, i) for i in range(5)] + [UnitSpec('of', 4, 0)]
    monkeypatch.delenv('VV_WINO44', raising=False)
    b = UNetBank(units, nf=32, device='cpu')
    assert not any(b._w44(256, l, False) for l in b.lay.convs)
    assert sorted(l.idx for l in b.lay.convs if b._w44(256, l, True)) == [3, 8, 10, 11]
    assert b._w44_pack(256)[1] == 4 and b._w44_pack(5) is None
    monkeypatch.setenv('VV_WINO44', '0')
    b = UNetBank(units, nf=32, device='cpu')
    assert not any(b._w44(256, l, d) for l in b.lay.convs for d in (False, True)) and b._w44_pack(256) is None
    monkeypatch.setenv('VV_WINO44', '1')
    b = UNetBank(units, nf=32, device='cpu')
    fwd = sorted(l.idx for l in b.lay.convs if b._w44(256, l, False))
    dgr = sorted(l.idx for l in b.lay.convs if b._w44(256, l, True))
    assert fwd == [3, 10, 11, 12] and dgr == [3, 8, 10, 11], (fwd, dgr)
    assert not any(b._w44(5, l, d) for l in b.lay.convs for d in (False, True))        # small batches: too few workgroups
    tab, n, mx = b._w44_pack(256)
    assert n == 8 and mx == 128 * 256
    monkeypatch.setenv('VV_WINO44', 'all')
    b = UNetBank(units, nf=32, device='cpu')
    assert all(b._w44(5, l, False) for l in b.lay.convs) and all(b._w44(5, l, True) for l in b.lay.convs[1:])
    ws = b.workspace(5)
    calls = ws.fwd[True].calls
    assert [c[2] for c in calls if c[2].startswith('pack')] == ['pack_wino', 'pack_tail', 'pack_wino_tail', 'pack_wino44']
    assert all(c[0] is b.lib.vv_conv_wino44 for c in calls if c[2].startswith('conv') and c[2][4:].isdigit())
