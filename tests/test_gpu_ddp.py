"""GPU: the multi-process data-parallel train step (one process per rank, per-rank BatchNorm statistics, bucketed gradient
all-reduce overlapped with the encoder backward, 1/world folded into Adam) against an oracle emulation of the reference's
DataParallel semantics (train.py:375: scatter the batch, per-replica BN, summed gradients of the global-mean loss).
Both ranks share GPU 0 and talk over gloo here (RCCL needs one GPU per rank); the collective API calls are identical."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, precision='fp32'):
    import torch.distributed as dist
    os.environ['VV_PRECISION'] = precision
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer, shard_batch
    torch.cuda.set_device(0)
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    net = net.cuda().train()
    raw, flow = O.seeded_cubes(8, 1, 21)
    tr = FusedTrainer(net, process_group=dist.group.WORLD, overlap=True)      # exercises the side stream + bucket ordering
    idx = shard_batch(torch.arange(8, device='cuda'), rank, world)
    for _ in range(2):
        tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), idx)
    torch.cuda.synchronize()
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}))      # numpy: pickled by value
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_dataparallel_semantics():
    import torch.multiprocessing as mp
    from oracle import unet_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in ps)
    res = {r: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()} for r, d in res.items()}
    for p in ps:
        p.join(120)
    # oracle emulation: per-shard forward/backward (own BN statistics), gradients averaged, one Adam step -- twice
    torch.set_num_threads(8)
    spec = O.bank_spec('net4')
    raw, flow = O.seeded_cubes(8, 1, 21)
    x, x_of = O.cubes_to_inputs(raw, flow)
    sds = [O.seeded_state_dict('net4', nf=32, padding=False, seed=0) for _ in range(2)]     # replica buffers (BN stats) per rank
    names = O.param_names(sds[0])
    opt = O.AdamState(names)
    for step in range(2):
        grads = []
        for r in range(2):
            sd = sds[r]
            for n in names:
                sd[n] = sds[0][n].detach().clone().requires_grad_(True)      # same weights on both replicas
            of_o, raw_o, of_t, raw_t = O.bank_forward(sd, spec, x[r * 4:(r + 1) * 4], x_of[r * 4:(r + 1) * 4], True, False)
            loss, _, _ = O.train_loss(of_o, raw_o, of_t, raw_t)
            loss.backward()
            grads.append({n: sd[n].grad.detach().clone() for n in names})
        avg = {n: (grads[0][n] + grads[1][n]) * 0.5 for n in names}
        with torch.no_grad():
            for n in names:
                sds[0][n] = sds[0][n].detach()
            opt.step(sds[0], avg)
    ref0 = sds[0]
    # both ranks hold identical parameters; rank r's running statistics are those of its own shard
    for k in names:
        assert torch.equal(res[0][k], res[1][k]), k
    # After 2 Adam steps every parameter moved by ~2e-3 (Adam's first steps are lr*sign(g)); a gradient whose sign is decided
    # by fp32 round-off flips that direction, so compare (a) the relative L2 distance and (b) the fraction of such flips.
    num = den = 0.0
    flips = total = 0
    for k in names:
        if k.endswith('.0.bias') or k.endswith('.3.bias'):
            continue
        d = (res[0][k].double() - ref0[k].double())
        num += float((d ** 2).sum())
        den += float((ref0[k].double() ** 2).sum())
        flips += int((d.abs() > 5e-4).sum())
        total += d.numel()
    assert num <= (5e-3 ** 2) * den, (num, den)
    assert flips <= 1e-2 * total, (flips, total)      # ~0.5 % observed (4 cubes per rank: many near-zero gradients)
    for r in range(2):
        for k in sds[r]:
            if k.endswith('running_mean') or k.endswith('running_var'):
                # running statistics sit downstream of those Adam sign flips (a flipped weight moves by 2*lr = 2e-3)
                assert torch.allclose(res[r][k], sds[r][k], rtol=5e-3, atol=1.5e-3), (r, k)


def test_two_rank_step_mixed_precision_ranks_agree():
    """The same two-rank schedule in mixed precision (`precision = bf16`): the gradient exchange is fp32 either way, so both
    ranks must end with bit-identical, finite parameters that moved away from the initial ones, and per-rank BatchNorm statistics."""
    import torch.multiprocessing as mp
    from oracle import unet_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, 'bf16')) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(120)
    sd0 = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
    names = O.param_names(sd0)
    moved = 0.0
    for k in names:
        a, b = np.asarray(res[0][k]), np.asarray(res[1][k])
        assert np.array_equal(a, b), k
        assert np.isfinite(a).all(), k
        moved = max(moved, float(np.abs(a - sd0[k].numpy()).max()))
    assert 5e-4 < moved < 1e-2                      # two Adam steps of lr = 1e-3
    rm = [k for k in res[0] if k.endswith('running_mean')]
    assert any(not np.array_equal(np.asarray(res[0][k]), np.asarray(res[1][k])) for k in rm)      # per-rank statistics


def _worker_default_init(rank, world, port, q):
    """Ranks build the net from DIFFERENT unseeded-style initialisers (what `torchrun train.py` does): the trainer must
    broadcast rank 0's parameters / buffers before the first step (train.py:217)."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer, shard_batch
    torch.cuda.set_device(0)
    torch.manual_seed(1000 + rank)
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False).cuda().train()
    before = net.bank().params.clone()
    tr = FusedTrainer(net, process_group=dist.group.WORLD)
    after_init = tr.bank.params.clone()
    raw, flow = O.seeded_cubes(8, 1, 21)
    idx = shard_batch(torch.arange(8, device='cuda'), rank, world)
    for _ in range(2):
        tr.step_cubes(torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda(), idx)
    tr.sync_from_rank0(params=False)          # what train.py does before the eval-mode scoring pass / the save
    torch.cuda.synchronize()
    q.put((rank, before.cpu().numpy(), after_init.cpu().numpy(), tr.bank.params.cpu().numpy(), tr.bank.bufs.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_default_init_is_broadcast_from_rank0():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker_default_init, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=600) for _ in ps)}
    for p in ps:
        p.join(120)
    assert not np.array_equal(res[0][0], res[1][0])              # the two ranks really started from different weights
    assert np.array_equal(res[0][1], res[0][0])                  # rank 0 keeps its own
    assert np.array_equal(res[1][1], res[0][0])                  # rank 1 took rank 0's
    assert np.array_equal(res[0][2], res[1][2])                  # ... and they stay bit-identical through the steps
    assert not np.array_equal(res[0][2], res[0][1])
    assert np.array_equal(res[0][3], res[1][3])                  # running statistics: rank 0's after sync_from_rank0


def _worker_rccl_world1(port, q):
    """The real backend ('nccl' = RCCL) in a one-rank group on the one GPU of the box: init_process_group(device_id=...),
    GradBuckets staging + all_reduce on device tensors on the communication stream, 1/world in Adam."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer
    raw, flow = O.seeded_cubes(12, 1, 33)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    outs = []
    # last configuration: the hipGraph path -- the step captured as three segments with the collectives launched between them
    for group, overlap, graphed in ((None, False, False), (dist.group.WORLD, False, False), (dist.group.WORLD, True, False),
                                    (dist.group.WORLD, False, True)):
        net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                               padding=False)
        net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
        net = net.cuda().train()
        tr = FusedTrainer(net, process_group=group, overlap=overlap, always_bucket=True)
        assert (tr.buckets is not None) == (group is not None)
        if tr.buckets is not None and not graphed:
            tr.buckets.timing = []
        for _ in range(4):          # graphed: eager step, capturing step, two replays
            tr.step_cubes(rawd, flowd, torch.arange(12, device='cuda'))
        torch.cuda.synchronize()
        n_coll = len(tr.buckets.timing) if (tr.buckets is not None and not graphed) else 0
        if graphed:
            caps = [c for k, c in tr._graphs.items() if k[0] == 'train' and c != 'warm']
            n_coll = -len(caps[0].segments) if caps else 0
        outs.append((tr.bank.params.cpu().numpy(), n_coll))
    t = torch.ones(4, device='cuda')
    dist.all_reduce(t)
    q.put((outs, float(t.sum()), dist.get_backend()))
    dist.destroy_process_group()


def test_rccl_world1_bucketed_step_bitwise_equal_to_no_group():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30900 + (os.getpid() % 1000)
    p = ctx.Process(target=_worker_rccl_world1, args=(port, q))
    p.start()
    outs, s, backend = q.get(timeout=600)
    p.join(120)
    assert backend == 'nccl' and s == 4.0
    assert outs[0][1] == 0 and outs[1][1] == 12 and outs[2][1] == 12       # 3 buckets x 4 steps went through RCCL
    assert outs[3][1] == -4                                                  # captured as 4 segments (3 exchanges between them)
    for k in (1, 2, 3):
        assert np.array_equal(outs[0][0], outs[k][0]), k


def _worker_uneven(rank, world, port, q):
    """The last, uneven global batch of an epoch under torchrun (train.py:373 keeps it; DataParallel scatters ceil(n/world)
    chunks): 5 cubes -> 3 + 2, then 1 cube -> 1 + 0."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle as O
    from model.unet import SelfCompleteNet4
    from vec_vad_amd.trainer import FusedTrainer
    torch.cuda.set_device(0)
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False)
    net.load_state_dict(O.seeded_state_dict('net4', nf=32, padding=False, seed=0))
    net = net.cuda().train()
    raw, flow = O.seeded_cubes(6, 1, 31)
    rawd, flowd = torch.from_numpy(raw).cuda(), torch.from_numpy(flow).cuda()
    tr = FusedTrainer(net, process_group=dist.group.WORLD)
    out = {}
    per = 3                                             # ceil(5 / 2)
    idx = torch.arange(5, device='cuda')[rank * per:(rank + 1) * per]
    ws = tr.step_cubes_uneven(rawd, flowd, idx, 5)
    out['grads5'] = (tr.bank.grads / world).cpu().numpy()
    out['ws5'] = ws is not None
    idx = torch.arange(5, 6, device='cuda')[rank:rank + 1]          # 1 cube: rank 0 has it, rank 1 has none
    ws = tr.step_cubes_uneven(rawd, flowd, idx, 1)
    out['ws1'] = ws is not None
    out['params'] = tr.bank.params.cpu().numpy()
    torch.cuda.synchronize()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_global_batch_is_the_global_mean_gradient():
    import torch.multiprocessing as mp
    from oracle import unet_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31900 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(120)
    assert res[0]['ws5'] and res[1]['ws5'] and res[0]['ws1'] and not res[1]['ws1']
    assert np.array_equal(res[0]['grads5'], res[1]['grads5']) and np.array_equal(res[0]['params'], res[1]['params'])
    assert np.isfinite(res[0]['params']).all()
    # oracle: per-shard train-mode forward / backward (own BatchNorm statistics), gradients weighted by B_r / N
    torch.set_num_threads(8)
    spec = O.bank_spec('net4')
    raw, flow = O.seeded_cubes(6, 1, 31)
    x, x_of = O.cubes_to_inputs(raw, flow)
    sd0 = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
    names = O.param_names(sd0)
    tot = {n: torch.zeros_like(sd0[n]) for n in names}
    for lo, hi in ((0, 3), (3, 5)):
        sd = {k: v.clone() for k, v in sd0.items()}
        for n in names:
            sd[n].requires_grad_(True)
        of_o, raw_o, of_t, raw_t = O.bank_forward(sd, spec, x[lo:hi], x_of[lo:hi], True, False)
        loss, _, _ = O.train_loss(of_o, raw_o, of_t, raw_t)
        loss.backward()
        for n in names:
            tot[n] += sd[n].grad * ((hi - lo) / 5.0)
    # map the flat HIP gradient buffer back to names through a bank built in this process
    from model.unet import SelfCompleteNet4
    net = SelfCompleteNet4(features_root=32, tot_raw_num=5, tot_of_num=1, border_mode='predict', rawRange=None, useFlow=True,
                           padding=False).cuda()
    bank = net.bank()
    by_id = {id(p): (g, key) for (g, key, p) in net._param_index}
    flat = torch.from_numpy(res[0]['grads5'])
    num = den = 0.0
    for name, p in net.named_parameters():
        if name.endswith('.0.bias') or name.endswith('.3.bias'):
            continue
        g, key = by_id[id(p)]
        got = bank.grad_view(g, key, grads=flat, shape=p.shape).double()
        num += float(((got - tot[name].double()) ** 2).sum())
        den += float((tot[name].double() ** 2).sum())
    assert num <= (2e-2 ** 2) * den, (num, den)          # 3- and 2-cube train-mode BatchNorm: tie flips dominate (cf. the small-batch tests)


def test_bench_two_ranks_on_one_gpu_prints_one_valid_line():
    """bench.py's N > 1 path end to end (the driver's SCALE run): two ranks under torch.distributed.run -- both on GPU 0 with the gloo
    backend, the bring-up mode of bench.py (one GPU on this box; RCCL with N > 1 is the driver's to run) -- captured step in four
    hipGraph segments with the three gradient exchanges between them.  Rank 0 prints ONE JSON line with the whole-job rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VV_SINGLE_DEVICE='1', VV_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    port = 31900 + (os.getpid() % 1000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '3', '--batch', '16',
           '--pool', '64', '--no-cpu-baseline', '--dp-split-of', '8']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['scaling'] == 'weak' and d['unit'] == 'cubes/s'
    assert d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 32 * 6 / (d['ms_per_step'] * 6e-3)) <= 1e-6 * d['value']
    assert d['comm']['ranks'] == 2 and len(d['comm']['buckets']) == 3
    assert 0 < d['comm']['per_rank_cubes_per_s']['min'] <= d['comm']['per_rank_cubes_per_s']['max']
    assert '4 segment' in d['execution']['mode'], d['execution']
    assert np.isfinite(d['config']['loss_raw'])
    # graph mode reports the exposed communication (one event pair around the last bucket + wait, between the captured segments)
    assert d['comm']['exposed_comm_steps_timed'] >= 6 and d['comm']['exposed_comm_us_per_step'] > 0
    # ... and the second, DataParallel-faithful record beside the weak-scaling headline: one 8-cube batch over 2 ranks
    assert list(d['configs']) == ['dp_faithful_global_batch_8'], list(d.get('configs', {}))
    dp = d['configs']['dp_faithful_global_batch_8']
    assert dp['config']['batch_per_gpu'] == 4 and dp['config']['global_batch'] == 8 and dp['value'] > 0
    assert d['config']['dp_faithful_cubes_per_s'] == dp['value'] and 'weak: 16 cubes per GPU' in d['config']['scaling_mode']
    assert dp['comm']['exposed_comm_us_per_step'] > 0


def test_bench_eight_ranks_on_one_gpu_dataparallel_split():
    """BASELINE configs[2] host logic before the first real SCALE run: `bench.py --gpus 8 --batch 4` under torch.distributed.run,
    all eight ranks on GPU 0 over gloo (bring-up mode; RCCL with N > 1 is the driver's to run).  One JSON line, 8 ranks in `comm`,
    global batch 32, the captured step in four segments with the three exchanges between them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VV_SINGLE_DEVICE='1', VV_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    port = 32900 + (os.getpid() % 1000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '3', '--batch', '4',
           '--pool', '32', '--no-cpu-baseline', '--dp-split-of', '0']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp8'
    assert d['comm']['ranks'] == 8 and d['comm']['backend'] == 'gloo' and len(d['comm']['buckets']) == 3
    assert abs(d['value'] - 32 * 4 / (d['ms_per_step'] * 4e-3)) <= 1e-6 * d['value']
    assert '4 segment' in d['execution']['mode'], d['execution']
    assert np.isfinite(d['config']['loss_raw'])


def test_bench_under_the_drivers_launcher_on_rccl_world1():
    """First-SCALE-run hardening: the EXACT command line the driver uses at N = 8 -- python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ... -- at N = 1 with
    VV_FORCE_DIST=1, so that it runs on the real `nccl` (= RCCL) backend on this one-GPU box: init_process_group('nccl', device_id=),
    the step captured in four hipGraph segments with the three in-place bucket all-reduces between them (a sum over one rank is the
    identity), max-over-ranks timing through dist.all_reduce / all_gather on device tensors, the `comm` record, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VV_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('VV_DIST_BACKEND', None)
    env.pop('VV_SINGLE_DEVICE', None)
    port = 33900 + (os.getpid() % 1000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '3', '--batch', '32',
           '--pool', '128', '--no-cpu-baseline', '--no-secondary']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['config']['global_batch'] == 32
    assert d['comm']['ranks'] == 1 and d['comm']['backend'] == 'nccl' and len(d['comm']['buckets']) == 3
    assert '4 segment' in d['execution']['mode'], d['execution']
    assert np.isfinite(d['config']['loss_raw']) and d['value'] > 0
    assert abs(d['value'] - 32 * 6 / (d['ms_per_step'] * 6e-3)) <= 1e-6 * d['value']
