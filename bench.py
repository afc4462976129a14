#!/usr/bin/env python
"""Headline benchmark: spatio-temporal cubes/s of one TRAIN step (forward + backward + Adam [+ RCCL gradient
all-reduce]) of SelfCompleteNet4 (5raw+1of, nf=32) on synthetic 32x32x5 RGB + flow cubes, batch 256 per GPU, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: "UCSDped2 5raw+1of UNet train, batch 256, 1xMI355X, HIP conv kernels".
Cubes live on the GPU as uint8 / fp32 (the reference's on-disk cube layout); every step gathers a fresh random batch
(vv_cube_gather), runs the grouped HIP forward, backward and fused Adam.  Multi-GPU: one process per GPU, each rank
trains on its own 256 cubes per step (weak scaling) and the flat gradient buffer is all-reduced over RCCL.

Rank 0 prints ONE JSON line; `roofline` describes the dominant kernel (the MFMA 3x3 convolution used by forward and
data-gradient), timed with HIP events on the launch stream inside the timed region; `cpu_baseline` times the
reference's op sequence (oracle, stock PyTorch CPU ops) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP_NET4 = 1855520768      # per cube, SURVEY.md section 8(d)
TRAIN_FLOP_NET4 = 5524094976
FP32_MFMA_PEAK = 157.3e12       # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md
BF16_MFMA_PEAK = 2.5e15         # v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK = 8.0e12               # HBM3E, MI355X_MICROARCH.md


def conv_flops(lay, B, G):
    """algorithmic FLOPs (2/MAC, true Cin, no padding) of every MFMA-conv launch label."""
    fl = {}
    for l in lay.convs:
        f = 2.0 * B * l.H * l.H * 9 * l.cin * l.cout * G
        fl['conv%d' % l.idx] = f
        if l.idx > 0:
            fl['dgrad%d' % l.idx] = f
    return fl


def conv_bytes(lay, B, G, y_bytes=4, dy_bytes=4, da_bytes=4):
    """algorithmic HBM bytes of the same launches: the input tensor read once + the output tensor written once
    (every conv output round-trips HBM exactly once); weights (L2 resident) not counted.
    Element sizes: 4 everywhere in fp32; in mixed precision the bank stores conv outputs / inputs (y_bytes), the data gradient's
    input dy (dy_bytes) and the activation gradients it writes (da_bytes) as bf16."""
    by = {}
    for l in lay.convs:
        cin = l.cinp if l.idx == 0 else l.cin                 # layer 0 reads the 16-channel frame-erased buffer
        by['conv%d' % l.idx] = 1.0 * B * l.H * l.H * G * y_bytes * (cin + l.cout)
        if l.idx > 0:
            by['dgrad%d' % l.idx] = 1.0 * B * l.H * l.H * G * (dy_bytes * l.cout + da_bytes * l.cin)
    return by


def pmc_traffic(name='r01_pmc_hbm_traffic.json'):
    """HBM bytes per launch of the conv family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
    separately, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); None when no such profile is committed."""
    p = os.path.join(ROOT, 'profiles', name)
    try:
        d = json.load(open(p))
        return (d.get('conv_family') or d['conv_mfma_family'])['hbm_bytes_per_launch_corrected']
    except Exception:
        return None


def cpu_baseline(batch=32, steps=3, thread_choices=(8, 16, 32, 64)):
    """Reference op sequence (torch CPU ops, NCHW fp32, per-op BN/ReLU/pool/cat, Adam eps=1e-7) on the host cores:
    BASELINE.json configs[0] (Net4, B=32).  This is the oracle restatement ("port"); it is test/bench
    infrastructure and never part of the product path.  The host of an MI355X box has 256 hardware threads and the
    B=32 workload scales badly past ~16 of them (oneDNN/OpenMP oversubscription), so a few thread counts are tried
    (bounded: one warm-up + `steps` timed steps each) and the best one is reported with its thread count."""
    from oracle import unet_oracle as O
    ncpu = os.cpu_count() or 1
    spec = O.bank_spec('net4')
    raw, flow = O.seeded_cubes(batch, 1, 3)
    x, x_of = O.cubes_to_inputs(raw, flow)
    best = None
    tried = []
    for nt in [t for t in thread_choices if t <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        sd = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
        opt = O.AdamState(O.param_names(sd))
        O.train_step(sd, spec, x, x_of, opt)          # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            O.train_step(sd, spec, x, x_of, opt)
        dt = (time.perf_counter() - t0) / steps
        tried.append('%d thr: %.0f cubes/s' % (nt, batch / dt))
        if best is None or batch / dt > best[0]:
            best = (batch / dt, nt)
        if dt * steps > 20:
            break
    return {'value': best[0], 'unit': 'cubes/s', 'cores': best[1], 'kind': 'port',
            'sample': 'SelfCompleteNet4 train step (fwd+bwd+Adam) on stock torch %s CPU ops, batch %d, %d timed steps per '
                      'thread count; host has %d hardware threads; tried: %s'
                      % (torch.__version__, batch, steps, ncpu, '; '.join(tried))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='cubes per GPU per step')
    ap.add_argument('--pool', type=int, default=4096, help='device-resident synthetic cubes per GPU')
    ap.add_argument('--model', default='net4', choices=['net4', 'full'])
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="bf16 = BASELINE config 4's mixed precision (bf16 conv operands, fp32 accumulation / tensors / BatchNorm / "
                         "Adam); the headline number is fp32, like the reference")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='print a per-launch time table to stderr')
    ap.add_argument('--overlap', nargs='?', const='free', default='none', choices=('none', 'free', 'paired'),
                    help="side stream for the weight-gradient kernels: 'free' = under everything that follows (conv launches "
                         "are then contended), 'paired' = only under the next layer's BatchNorm backward (conv launches run alone)")
    args = ap.parse_args()
    os.environ['VV_PRECISION'] = args.precision

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    if os.environ.get('VV_SINGLE_DEVICE'):          # bring-up: all ranks share GPU 0 (gloo backend only)
        local_rank_dev = 0
    else:
        local_rank_dev = local_rank
    torch.cuda.set_device(local_rank_dev)
    dev = torch.device('cuda', local_rank_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('VV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; 'gloo' only for single-GPU bring-up tests
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from model.unet import SelfCompleteNet4, SelfCompleteNetFull
    from vec_vad_amd.trainer import FusedTrainer
    torch.manual_seed(0)
    tot_of = 1 if args.model == 'net4' else 5
    cls = SelfCompleteNet4 if args.model == 'net4' else SelfCompleteNetFull
    net = cls(features_root=32, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=None, useFlow=True,
              padding=False).to(dev)
    trainer = FusedTrainer(net, lr=1e-3, eps=1e-7, process_group=dist.group.WORLD if dist is not None else None,
                           overlap={'none': False, 'free': True, 'paired': 'paired'}[args.overlap])
    bank = trainer.bank
    B = args.batch
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    raw = torch.randint(0, 256, (args.pool, 5, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
    flow = (torch.randn((args.pool, tot_of, 32, 32, 2), generator=g) * 2.0).to(dev)
    perm = torch.stack([torch.randperm(args.pool, generator=g)[:B] for _ in range(args.steps + args.warmup)]).to(dev)

    for it in range(args.warmup):
        trainer.step_cubes(raw, flow, perm[it])
    torch.cuda.synchronize()
    ws = bank.workspace(B)
    fl = conv_flops(bank.lay, B, bank.Ga)
    by = conv_bytes(bank.lay, B, bank.Ga, 2 if getattr(bank, 'y16', False) else 4, 2 if getattr(bank, 'dz16', False) else 4,
                    2 if getattr(bank, 'da16', False) else 4)
    # HIP events around every MFMA 3x3-conv launch (forward conv + data-gradient) of the timed region
    ev = []
    trainer.event_hook = lambda label, a, b: ev.append((label, a, b))
    trainer.event_labels = set(fl.keys()) if not args.breakdown else None

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        trainer.step_cubes(raw, flow, perm[args.warmup + it])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    trainer.event_hook = None
    # a few extra (untimed) steps with the side stream disabled: the same launches without a concurrent weight-grad kernel
    iso = []
    if args.overlap == 'free':
        trainer.event_hook = lambda label, a, b: iso.append((label, a, b))
        trainer.event_labels = set(fl.keys())
        trainer.overlap = False
        for it in range(3):
            trainer.step_cubes(raw, flow, perm[it % perm.shape[0]])
        torch.cuda.synchronize()
        trainer.overlap = True
        trainer.event_hook = None
    # forward-only rate (BASELINE.json north_star: ">= 50 % of the MFMA roofline on the UNet forward at batch 256"):
    # cube gather + train-mode forward (BatchNorm batch statistics, loss + per-cube scores), outside the timed region
    fwd_n = 10
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bufs, nbt = bank.bufs.clone(), bank.nbt.clone()          # forward(train=True) moves the BatchNorm running statistics
    f0.record()
    for it in range(fwd_n):
        bank.set_input_cubes(raw, flow, perm[it % perm.shape[0]], B)
        bank.forward(ws, True)
    f1.record()
    torch.cuda.synchronize()
    bank.bufs.copy_(bufs)
    bank.nbt.copy_(nbt)
    fwd_ms = f0.elapsed_time(f1) / fwd_n
    iso_t = sum(a.elapsed_time(b) * 1e-3 for _, a, b in iso)
    iso_f = sum(fl[label] for label, _, _ in iso)
    l_raw, l_of = bank.losses(ws)
    loss_now = (float(l_raw), float(l_of) if l_of is not None else 0.0)

    per = {}
    for label, a, b in ev:
        per.setdefault(label, []).append(a.elapsed_time(b) * 1e-3)
    conv_t = sum(sum(v) for k, v in per.items() if k in fl)
    conv_n = sum(len(v) for k, v in per.items() if k in fl)
    conv_f = sum(fl[k] * len(v) for k, v in per.items() if k in fl)
    conv_b = sum(by[k] * len(v) for k, v in per.items() if k in fl)
    if args.breakdown and rank == 0:
        tot = sum(sum(v) for v in per.values())
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            extra = ''
            if k in fl:
                extra = '  %.1f TF/s' % (fl[k] / (sum(v) / len(v)) / 1e12)
            sys.stderr.write('%-22s n=%3d avg %8.1f us  %5.1f%%%s\n' % (k, len(v), 1e6 * sum(v) / len(v), 100 * sum(v) / tot, extra))
        sys.stderr.write('sum of launches %.3f ms / step ; wall %.3f ms / step\n' % (1e3 * tot / args.steps, 1e3 * dt / args.steps))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    cubes = B * world * args.steps
    value = cubes / dt
    flop_per_cube = TRAIN_FLOP_NET4 if args.model == 'net4' else 9206169600
    fwd_flop_per_cube = 1855520768 if args.model == 'net4' else 3092316160
    pmc_ok = args.model == 'net4' and B == 256          # the committed PMC passes were taken on the default workload
    out = {
        'metric': 'spatio-temporal cubes/sec (train step)', 'value': value, 'unit': 'cubes/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16 operands, f32 accumulate',
        'data': 'synthetic',
        'config': {'workload': 'UCSDped2-shaped 5raw+1of UNet bank (SelfCompleteNet4, nf=32, padding=False) train step: '
                               'cube gather + forward + backward + Adam(eps=1e-7)' if args.model == 'net4' else
                               '5raw+5of UNet bank (SelfCompleteNetFull) train step',
                   'batch_per_gpu': B, 'global_batch': B * world, 'cube': '32x32x5 RGB uint8 + flow fp32',
                   'parallelism': 'dp%d' % world, 'train_tflops_per_gpu': value / world * flop_per_cube / 1e12,
                   'frac_of_fp32_mfma_peak_whole_step': value / world * flop_per_cube / FP32_MFMA_PEAK,
                   'forward_ms': fwd_ms, 'forward_tflops_per_gpu': B * fwd_flop_per_cube / (fwd_ms * 1e-3) / 1e12,
                   'forward_frac_of_fp32_mfma_peak': B * fwd_flop_per_cube / (fwd_ms * 1e-3) / FP32_MFMA_PEAK,
                   'loss_raw': loss_now[0], 'loss_of': loss_now[1]},
        'roofline': {'bound': 'mfma',
                     'kernel': ('wino_conv_kernel (3x3 conv as Winograd F(2x2,3x3) on MFMA, forward + data-gradient launches): '
                                'achieved = ALGORITHMIC (direct-convolution) FLOP / time, so it can exceed the peak; '
                                'executed_* = the 2.25x fewer multiply-adds the matrix cores actually run')
                     if bank.wino else 'conv_mfma_kernel (3x3 implicit-GEMM, forward + data-gradient launches)',
                     'achieved': (conv_f / conv_t / 1e12) if conv_t > 0 else None, 'peak': FP32_MFMA_PEAK / 1e12,
                     'unit': 'TFLOP/s', 'frac': (conv_f / conv_t / FP32_MFMA_PEAK) if conv_t > 0 else None,
                     'traffic': pmc_traffic() if pmc_ok else None, 'launches_timed': conv_n,
                     'avg_launch_us': (1e6 * conv_t / conv_n) if conv_n else None,
                     'algorithmic_gflop_per_launch': (conv_f / conv_n / 1e9) if conv_n else None,
                     'executed_tflops': (conv_f / conv_t / 1e12 / (2.25 if bank.wino else 1.0)) if conv_t > 0 else None,
                     'executed_frac': (conv_f / conv_t / FP32_MFMA_PEAK / (2.25 if bank.wino else 1.0)) if conv_t > 0 else None,
                     'side_stream_weight_grad': args.overlap,
                     'isolated_frac': (iso_f / iso_t / FP32_MFMA_PEAK) if iso_t > 0 else None},
    }
    if args.precision == 'bf16':
        # the bf16 matrix instruction needs 1/16 of the fp32 one's cycles: the same launches are bound by HBM (fp32 tensors)
        mf = out['roofline']
        out['roofline'] = {'bound': 'hbm',
                           'kernel': 'conv_mfma_kernel<..., BF=true> (3x3 implicit GEMM, bf16 operands / fp32 accumulation, forward + '
                                     'data-gradient launches): achieved = algorithmic bytes (input read once + output written once; bf16 '
                                     'tensors) / time.  At 325 FLOP/B these launches sit on the ridge of the bf16 roofline: see frac_of_bf16_mfma_peak',
                           'achieved': (conv_b / conv_t / 1e9) if conv_t > 0 else None, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                           'frac': (conv_b / conv_t / HBM_PEAK) if conv_t > 0 else None,
                           'traffic': pmc_traffic('r01_pmc_hbm_traffic_bf16.json') if pmc_ok else None, 'launches_timed': conv_n,
                           'avg_launch_us': mf['avg_launch_us'],
                           'algorithmic_mbytes_per_launch': (conv_b / conv_n / 1e6) if conv_n else None,
                           'algorithmic_gflop_per_launch': mf['algorithmic_gflop_per_launch'],
                           'mfma_tflops': mf['achieved'], 'frac_of_bf16_mfma_peak': (mf['achieved'] * 1e12 / BF16_MFMA_PEAK)
                           if mf['achieved'] else None, 'side_stream_weight_grad': args.overlap}
        cfgd = out['config']
        for k_ in ('frac_of_fp32_mfma_peak_whole_step', 'forward_frac_of_fp32_mfma_peak'):      # wrong denominator in this mode
            cfgd.pop(k_, None)
        cfgd['frac_of_bf16_mfma_peak_whole_step'] = value / world * flop_per_cube / BF16_MFMA_PEAK
        cfgd['forward_frac_of_bf16_mfma_peak'] = B * fwd_flop_per_cube / (fwd_ms * 1e-3) / BF16_MFMA_PEAK
        out['config']['precision'] = 'mixed bf16 (BASELINE config 4): conv / transposed-conv operands bf16, everything else fp32'
    if not args.no_cpu_baseline and world == 1:
        try:
            out['cpu_baseline'] = cpu_baseline()
        except Exception as e:   # the oracle is optional infrastructure for the bench line
            out['cpu_baseline'] = {'value': None, 'unit': 'cubes/s', 'cores': os.cpu_count(), 'kind': 'port',
                                   'sample': 'failed: %r' % (e,)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
