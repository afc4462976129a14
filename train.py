#!/usr/bin/env python
"""``python train.py`` -- normal-event modelling stage of VEC_VAD on the MI355X UNet-bank engine.

Drop-in for the reference's train.py:239-437 ("Normal video event modeling"): same ``config.cfg`` keys, same input
artifacts (``<data_root>/<modality>/<ds>_foreground_train_<mode>-raw.npy / -flow.npy``, ShanghaiTech segment files)
and same outputs (``<ds>_model_<mode>_SelfComplete.npy`` = torch-pickled nested list of ``state_dict`` with the
DataParallel ``module.`` key prefix, ``<ds>_{raw,of}_training_scores_<mode>_SelfComplete.npy``).

What is different on purpose:
  * the cubes are uploaded once and stay on the GPU in their on-disk layout; batches are gathered by a HIP kernel
    (no DataLoader / ToTensor / H2D copy per step, reference train.py:372-381);
  * forward + backward + Adam are the fused HIP path (vec_vad_amd.trainer.FusedTrainer); no per-step ``.item()`` sync
    -- the running losses are accumulated on the device and fetched only when a log line is printed;
  * ``torchrun --nproc-per-node N train.py`` gives one process per GPU with RCCL gradient all-reduce instead of
    nn.DataParallel; plain ``python train.py`` stays valid (1 GPU);
  * the epoch permutation is seeded (``[mi355x] shuffle_seed``) -- the reference shuffles unseeded.
Cube extraction (train.py:102-226) runs through ``foreground.extract_train`` (crop + resize on the GPU) when
``train_foreground_saved = False``; the detector stage that produces the boxes (train.py:44-95, mmdet) is outside the hot
path -- keep ``train_bbox_saved = True``.
"""
import os
import sys
from configparser import ConfigParser

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from model.unet import SelfCompleteNet4, SelfCompleteNetFull  # noqa: E402
from vad_datasets import CubeStore  # noqa: E402
from vec_vad_amd.trainer import FusedTrainer, shard_batch  # noqa: E402


class AverageMeter:
    """Running average of the per-batch losses (reference helper/misc.py AverageMeter; never reset, train.py:365)."""

    def __init__(self, device):
        self.sum = torch.zeros((), device=device, dtype=torch.float64)
        self.count = 0

    def update(self, val, n):
        self.sum += val.double() * n
        self.count += n

    @property
    def avg(self):
        return float(self.sum) / max(1, self.count)


def read_config(path='config.cfg'):
    cp = ConfigParser()
    if not cp.read(path):
        raise FileNotFoundError(path)
    method = cp.get('shared_parameters', 'method')
    ds = cp.get('shared_parameters', 'dataset_name')
    c = dict(cp=cp, dataset_name=ds, raw_dataset_dir=cp.get('shared_parameters', 'raw_dataset_dir'),
             mode_fg=cp.get('shared_parameters', 'foreground_extraction_mode'),
             data_root_dir=cp.get('shared_parameters', 'data_root_dir'), modality=cp.get('shared_parameters', 'modality'),
             method=method, h_block=cp.getint(ds, 'h_block'), w_block=cp.getint(ds, 'w_block'))
    if method != 'SelfComplete':
        raise NotImplementedError(method)
    border_mode = cp.get(method, 'border_mode')
    if border_mode == 'predict':                       # train.py:246-251
        tot_frame = cp.getint(method, 'context_frame_num') + 1
        tot_of = cp.getint(method, 'context_of_num') + 1
    else:
        tot_frame = 2 * cp.getint(method, 'context_frame_num') + 1
        tot_of = 2 * cp.getint(method, 'context_of_num') + 1
    raw_range = cp.getint(method, 'rawRange')
    if raw_range >= tot_frame:                         # train.py:252-254
        raw_range = None
    c.update(border_mode=border_mode, tot_frame_num=tot_frame, tot_of_num=tot_of, rawRange=raw_range,
             epochs=cp.getint(method, 'epochs'), batch_size=cp.getint(method, 'batch_size'),
             useFlow=cp.getboolean(method, 'useFlow'), padding=cp.getboolean(method, 'padding'),
             lambda_raw=cp.getfloat(method, 'lambda_raw'), lambda_of=cp.getfloat(method, 'lambda_of'),
             w_raw=cp.getfloat(method, 'w_raw'), w_of=cp.getfloat(method, 'w_of'), nf=cp.getint(method, 'nf'),
             shuffle_seed=cp.getint('mi355x', 'shuffle_seed', fallback=0),
             score_batch=cp.getint('mi355x', 'score_batch', fallback=2048),
             save_score_masks=cp.getboolean('mi355x', 'save_score_masks', fallback=True),
             overlap_wgrad=cp.getboolean('mi355x', 'overlap_wgrad', fallback=False),
             precision=cp.get('mi355x', 'precision', fallback='fp32').strip().lower())
    assert c['modality'] == 'raw2flow'
    return c


def build_network(c):
    """train.py:261-268 / test.py:216-224."""
    os.environ['VV_PRECISION'] = c.get('precision', 'fp32')       # read by the UNet bank when it is built ([mi355x] precision)
    kw = dict(features_root=c['nf'], tot_raw_num=c['tot_frame_num'], tot_of_num=c['tot_of_num'],
              border_mode=c['border_mode'], rawRange=c['rawRange'], useFlow=c['useFlow'], padding=c['padding'])
    if c['tot_of_num'] == 1:
        net = SelfCompleteNet4(**kw)
    elif c['tot_of_num'] == 5:
        net = SelfCompleteNetFull(**kw)
    else:
        raise NotImplementedError('context_of_num must be 0 or 4 (config.cfg; the reference falls through at train.py:266)')
    assert c['tot_frame_num'] == 5
    return net


def _dist():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None, 0, 1
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    return dist, dist.get_rank(), dist.get_world_size()


def _extract_once(c, device, dist=None, timeout_h=48.0):
    """Rank 0 cuts the training cubes (train.py:102-226); every other rank of the job waits for it.

    The wait is a barrier on a gloo SIDE group with its own, long, explicit timeout -- cutting a large dataset takes longer than
    the default collective timeout, and the RCCL group must not carry a 48 h watchdog for its real collectives.  No marker
    files: nothing survives a crash or a ``torchrun --max-restarts`` restart that a later attempt could mistake for "done", and
    the .npy files are written atomically (foreground.save_nested: tmp + os.replace), so a rank that gets past the barrier reads
    complete files.  All ranks must see ``data_root_dir`` (one node, or a shared filesystem); the ranks check that after the
    barrier instead of assuming it."""
    rank = dist.get_rank() if dist is not None else 0
    if dist is None:
        from foreground import extract_train
        extract_train(c, device)
        return
    import datetime
    side = dist.new_group(backend='gloo', timeout=datetime.timedelta(hours=timeout_h))
    err = None
    try:
        if rank == 0:
            try:
                from foreground import extract_train
                extract_train(c, device)
            except BaseException as e:      # incl. KeyboardInterrupt / SystemExit: the other ranks must not wait 48 h for a rank that
                err = e                     # is on its way out (a SIGKILL cannot be announced: gloo then fails the peers' broadcast
                                            # when the socket closes)
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32)
        dist.broadcast(flag, src=0, group=side)          # = the barrier; carries rank 0's verdict
    finally:
        dist.destroy_process_group(side)                 # also when the broadcast itself raised
    if err is not None:
        raise err
    if int(flag.item()):
        raise RuntimeError('rank 0 failed while extracting the training cubes')
    base = os.path.join(c['data_root_dir'], c['modality'], c['dataset_name'] + '_')
    fg = c['mode_fg']
    probe = base + ('foreground_train_{}_seg_0-raw.npy' if c['dataset_name'] == 'ShanghaiTech' else 'foreground_train_{}-raw.npy').format(fg)
    if not os.path.exists(probe):
        raise RuntimeError('rank %d cannot see %s: train.py under torchrun needs data_root_dir on a filesystem every rank mounts'
                           % (rank, probe))


def train_block(net, segments, epochs, batch_size, lambda_raw=1.0, lambda_of=1.0, shuffle_seed=0, device='cuda',
                log=print, tag='(0, 0)', dist=None, overlap=False):
    """The loop of train.py:375-427 for one (h, w) block.

    ``segments``: list of callables returning (raw uint8 [N,5,32,32,3], flow fp32 [N,(Tf,)32,32,2]) -- one entry for
    UCSDped2 / avenue, ``totSegNum`` entries for ShanghaiTech (train.py:292-299).
    Returns (state_dict with 'module.' prefixed keys, raw_scores [N_total], of_scores [N_total])."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    net = net.to(device)
    net.train()
    trainer = FusedTrainer(net, lr=1e-3, eps=1e-7, lambda_raw=lambda_raw, lambda_of=lambda_of,
                           process_group=dist.group.WORLD if dist is not None else None, overlap=overlap)
    raw_losses, of_losses = AverageMeter(device), AverageMeter(device)
    rng = np.random.default_rng(shuffle_seed) if shuffle_seed is not None and shuffle_seed >= 0 else None
    stores = [None] * len(segments)

    def store(i):
        if stores[i] is None or len(segments) > 1:
            if len(segments) > 1:
                trainer.release_graphs()      # the captured steps pin the previous segment's store: free it before the next upload
            raw, flow = segments[i]()
            s = CubeStore(raw, flow, device)
            if len(segments) == 1:
                stores[i] = s
            return s
        return stores[i]

    for epoch in range(epochs):
        for si in range(len(segments)):
            st = store(si)
            n = len(st)
            perm = rng.permutation(n) if rng is not None else np.arange(n)
            perm = torch.from_numpy(perm).to(device)
            nb = (n + batch_size - 1) // batch_size
            for idx in range(nb):
                bidx = perm[idx * batch_size:(idx + 1) * batch_size]          # the last partial batch is kept (train.py:373)
                n_glob = int(bidx.numel())
                if world > 1:
                    if n_glob % world == 0:
                        bidx = shard_batch(bidx, rank, world)
                        ws = trainer.step_cubes(st.raw, st.flow, bidx)
                    else:       # DataParallel's uneven scatter (chunks of ceil(n / world), trailing replicas may get nothing)
                        per = -(-n_glob // world)
                        bidx = bidx[rank * per:(rank + 1) * per]
                        ws = trainer.step_cubes_uneven(st.raw, st.flow, bidx, n_glob)
                else:
                    ws = trainer.step_cubes(st.raw, st.flow, bidx)           # a single cube is a valid batch (BatchNorm over H x W)
                if ws is not None:
                    l_raw, l_of = trainer.losses(ws)
                    raw_losses.update(l_raw, bidx.numel())
                    of_losses.update(l_of if l_of is not None else torch.zeros((), device=device), bidx.numel())
                if idx % 5 == 0:
                    avg_r, avg_o = _global_avg(dist, raw_losses, of_losses, device)      # every rank takes part; rank 0 prints
                    if rank == 0:
                        log('Block: {}, epoch {}, seg {}, batch {} of {}, raw loss: {}, of loss: {}'.format(
                            tag, epoch, si, idx, n // batch_size, avg_r, avg_o))
    # DataParallel keeps replica 0's BatchNorm statistics (train.py:375): every rank scores with -- and rank 0 saves -- those
    trainer.sync_from_rank0(params=False)
    sd = {('module.' + k): v.detach().clone() for k, v in net.state_dict().items()}     # train.py:410 (DataParallel keys)

    # A forward pass to store the training scores (train.py:413-427), eval mode, shuffle=False
    net.eval()
    rs, os_ = [], []
    for si in range(len(segments)):
        st = store(si)
        n = len(st)
        lo, hi = (rank * n) // world, ((rank + 1) * n) // world
        r_loc, o_loc = [], []
        for s in range(lo, hi, batch_size):
            idx = torch.arange(s, min(hi, s + batch_size), device=device)
            r, o = trainer.score_cubes(st.raw, st.flow, idx)
            r_loc.append(r.clone())
            if o is not None:
                o_loc.append(o.clone())
        r_loc = torch.cat(r_loc) if r_loc else torch.zeros(0, device=device)
        o_loc = torch.cat(o_loc) if o_loc else torch.zeros(0, device=device)
        if world > 1:
            r_loc, o_loc = _gather_var(dist, r_loc, n, world), _gather_var(dist, o_loc, n, world, net.useFlow)
        rs.append(r_loc.cpu().numpy())
        os_.append(o_loc.cpu().numpy())
    return sd, np.concatenate(rs), np.concatenate(os_)


def _global_avg(dist, raw_m, of_m, device):
    """running-average losses over ALL ranks' shards (the reference logs the loss of the whole batch, train.py:394-399)."""
    if dist is None:
        return raw_m.avg, of_m.avg
    t = torch.stack([raw_m.sum, of_m.sum, torch.tensor(float(raw_m.count), device=device, dtype=torch.float64)])
    dist.all_reduce(t)
    cnt = max(1.0, float(t[2]))
    return float(t[0]) / cnt, float(t[1]) / cnt


def _gather_var(dist, t, n, world, present=True):
    """all-gather of per-rank contiguous score shards (tiny; no collective in the math).  present=False: this score does not
    exist (useFlow = False) -> empty on every rank."""
    if not present or (t.numel() == 0 and n == 0):
        return t[:0]
    mx = (n + world - 1) // world + 1
    pad = torch.zeros(mx, device=t.device, dtype=t.dtype)
    pad[:t.numel()] = t
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    parts = []
    for r in range(world):
        lo, hi = (r * n) // world, ((r + 1) * n) // world
        parts.append(out[r][:hi - lo])
    return torch.cat(parts)


def main(config_path='config.cfg'):
    c = read_config(config_path)
    cp, ds, fg, root, mod, method = c['cp'], c['dataset_name'], c['mode_fg'], c['data_root_dir'], c['modality'], c['method']
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    dist, rank, world = _dist()
    if not cp.getboolean(ds, 'train_foreground_saved'):      # train.py:102-226, cubes cut on the GPU (vv_crop_resize)
        _extract_once(c, device, dist)
    net = build_network(c)
    base = os.path.join(root, mod, ds + '_')
    shanghai = ds == 'ShanghaiTech'

    if shanghai:
        save_seg = cp.getint(ds, 'saveSegNum')
        names = sorted(f for f in os.listdir(os.path.join(root, mod))
                       if f.startswith('%s_foreground_train_%s_seg_' % (ds, fg)) and f.endswith('-raw.npy'))
        tot_seg = len(names)
        probe = np.load(base + 'foreground_train_{}_seg_0-raw.npy'.format(fg), allow_pickle=True)
        grid = [(s, h, w) for s in range(len(probe)) for h in range(len(probe[s])) for w in range(len(probe[s][h]))]
        del probe
    else:
        fset = np.load(base + 'foreground_train_{}-raw.npy'.format(fg), allow_pickle=True)
        fset2 = np.load(base + 'foreground_train_{}-flow.npy'.format(fg), allow_pickle=True)
        grid = [(None, h, w) for h in range(len(fset)) for w in range(len(fset[h]))]

    def nested(fill):
        if shanghai:
            ns = max(g[0] for g in grid) + 1
            return [[[fill() for _ in range(c['w_block'])] for _ in range(c['h_block'])] for _ in range(ns)]
        return [[fill() for _ in range(len(fset[h]))] for h in range(len(fset))]

    model_set, raw_scores_set, of_scores_set = nested(list), nested(list), nested(list)
    for (s, h, w) in grid:
        if shanghai:
            def seg_loader(k, s=s, h=h, w=w):
                a = np.load(base + 'foreground_train_{}_seg_{}-raw.npy'.format(fg, k), allow_pickle=True)
                b = np.load(base + 'foreground_train_{}_seg_{}-flow.npy'.format(fg, k), allow_pickle=True)
                return np.asarray(a[s][h][w]), np.asarray(b[s][h][w])
            segments = [lambda k=k: seg_loader(k) for k in range(tot_seg)]
        else:
            data = np.asarray(fset[h][w])
            if len(data) <= 1:          # train.py:370 "num > 1 for data parallel"
                continue
            data2 = np.asarray(fset2[h][w])
            segments = [lambda data=data, data2=data2: (data, data2)]
        sd, r, o = train_block(net, segments, c['epochs'], c['batch_size'], c['lambda_raw'], c['lambda_of'],
                               c['shuffle_seed'], device, tag='({}, {})'.format(h, w), dist=dist, overlap=c['overlap_wgrad'])
        tgt = (model_set[s][h][w], raw_scores_set[s][h], of_scores_set[s][h]) if shanghai else \
              (model_set[h][w], raw_scores_set[h], of_scores_set[h])
        tgt[0].append({k: v.cpu() for k, v in sd.items()})
        tgt[1][w] = r
        tgt[2][w] = o
    if rank == 0:
        torch.save(raw_scores_set, base + 'raw_training_scores_{}_{}.npy'.format(fg, method))
        torch.save(of_scores_set, base + 'of_training_scores_{}_{}.npy'.format(fg, method))
        print('training scores saved!')
        torch.save(model_set, base + 'model_{}_{}.npy'.format(fg, method))
        print('Training of {} for dataset: {} has completed!'.format(method, ds))
    if dist is not None:
        dist.barrier()


if __name__ == '__main__':
    main()
