#!/usr/bin/env python
"""Headline benchmark: spatio-temporal cubes/s of one TRAIN step (forward + backward + Adam [+ RCCL gradient
all-reduce]) of SelfCompleteNet4 (5raw+1of, nf=32) on synthetic 32x32x5 RGB + flow cubes, batch 256 per GPU, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W            # weak scaling: 256 cubes per GPU
    ... bench.py --gpus 8 --batch 32                          # the reference's DataParallel split of ONE 256-cube batch

Workload = BASELINE.json configs[1]: "UCSDped2 5raw+1of UNet train, batch 256, 1xMI355X, HIP conv kernels".
Cubes live on the GPU as uint8 / fp32 (the reference's on-disk cube layout); every step gathers a fresh random batch
(vv_cube_gather), runs the grouped HIP forward, backward and fused Adam.  Multi-GPU: one process per GPU, each rank
trains on its own cubes and the flat gradient buffer is all-reduced over RCCL in three buckets.

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel family (3x3 convolution forward + data-gradient launches), timed with HIP events on the
                launch stream inside the timed region.  ``achieved`` / ``frac`` count the multiply-adds the matrix cores EXECUTE
                (the Winograd F(2x2,3x3) kernel executes 16/36 of the direct convolution's), so frac <= 1 is a hardware
                utilisation; the algorithmic (direct-convolution) rate is ``algorithmic_tflops`` and their ratio
                ``effective_vs_direct``.
  cpu_baseline  the reference's op sequence (oracle, stock PyTorch CPU ops) on the host cores: 3 warm-up + 10 timed steps per
                thread count, best-of sweep AND the all-cores figure, train step and eval forward.
  configs       secondary records measured in the same run (N=1 only): BASELINE config 4 (SelfCompleteNetFull, B=512, mixed
                bf16), config 5 (FlowNet2 forward on a 1024x448 pair) and the eval-mode scoring pass, each with its own roofline.
  comm          (N>1) ranks + backend, per-bucket all-reduce time and the exposed communication per step.
The scalar headline of every secondary record is repeated inside ``config`` (cfg4_cubes_per_s, flownet2_ms_per_pair, net4_b32_ms ...)
so that a consumer that keeps only the top-level keys of the line still has them.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP = {'net4': (1855520768, 5524094976), 'full': (3092316160, 9206169600)}      # (forward, train step) per cube, SURVEY 8(d)
FP32_MFMA_PEAK = 157.3e12       # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md
BF16_MFMA_PEAK = 2.5e15         # v_mfma_f32_32x32x16_bf16, dense
BF16_MFMA_CALIBRATED = 1.57e15  # the same instruction in a loop with the conv kernel's operand traffic, measured (profiles/r04_gemm16_loop_calibration.txt)
HBM_PEAK = 8.0e12               # HBM3E, MI355X_MICROARCH.md
WINO_EXEC = 16.0 / 36.0         # F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36
FLOWNET2_GFLOP = 464.2          # per 1024x448 pair, SURVEY appendix A.2


def conv_flops(lay, B, G):
    """algorithmic FLOPs (2/MAC, true Cin, no padding) of every MFMA-conv launch label."""
    fl = {}
    for l in lay.convs:
        f = 2.0 * B * l.H * l.H * 9 * l.cin * l.cout * G
        fl['conv%d' % l.idx] = f
        if l.idx > 0:
            fl['dgrad%d' % l.idx] = f
    return fl


def conv_exec_flops(lay, B, G, wino, w44=None):
    """FLOPs the matrix cores execute for the same launches: channels padded to the MFMA K granule (layer 0: 12 -> 16),
    x 16/36 for the Winograd F(2x2,3x3) form, x 36/(16*9) = 1/4 for the launches w44(l, dgrad) routes to F(4x4,3x3)."""
    fl = {}
    k = WINO_EXEC if wino else 1.0
    for l in lay.convs:
        fl['conv%d' % l.idx] = 2.0 * B * l.H * l.H * 9 * l.cinp * l.cout * G * (0.25 if (w44 and w44(l, False)) else k)
        if l.idx > 0:
            fl['dgrad%d' % l.idx] = 2.0 * B * l.H * l.H * 9 * l.cin * l.cout * G * (0.25 if (w44 and w44(l, True)) else k)
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


def wgrad_exec_flops(bank, B):
    """multiply-adds the matrix cores execute in the weight-gradient launches: the 3x3 layers in Winograd F(2x2,3x3) form (fp32 path:
    x16/36, K padded), the transposed convs and every bf16 launch direct."""
    lay, G = bank.lay, bank.Ga
    k = WINO_EXEC if (bank.wino_wgrad and not bank.cflag) else 1.0
    fx = {'wgrad%d' % l.idx: 2.0 * B * l.H * l.H * 9 * l.cinp * l.cout * G * k for l in lay.convs}
    fa = {'wgrad%d' % l.idx: 2.0 * B * l.H * l.H * 9 * l.cin * l.cout * G for l in lay.convs}
    for u, (_, H, ci, co) in enumerate(lay.convT):
        fx['wgradT%d' % u] = fa['wgradT%d' % u] = 2.0 * B * H * H * 9 * ci * co * G
    return fx, fa


def wgrad_bytes(bank, B):
    """algorithmic HBM bytes of the same launches: dy read once + the layer input read once (slabs / weights not counted)."""
    lay, G = bank.lay, bank.Ga
    yb, dzb, dab = (2 if bank.y16 else 4), (2 if bank.dz16 else 4), (2 if bank.da16 else 4)
    by = {'wgrad%d' % l.idx: 1.0 * B * l.H * l.H * G * (dzb * l.cout + yb * (l.cinp if l.idx == 0 else l.cin)) for l in lay.convs}
    for u, (_, H, ci, co) in enumerate(lay.convT):
        by['wgradT%d' % u] = 1.0 * B * H * H * G * (yb * ci + dab * 4 * co)
    return by


def bn_bwd_bytes(bank, B):
    """algorithmic HBM bytes of the BatchNorm-backward launches: the apply pass reads dA and z and writes dz, a separate
    reduce pass reads dA and z."""
    lay, G = bank.lay, bank.Ga
    yb, dzb, dab = (2 if bank.y16 else 4), (2 if bank.dz16 else 4), (2 if bank.da16 else 4)
    by = {}
    for l in lay.convs:
        n = 1.0 * B * l.H * l.H * l.cout * G
        by['bn_bwd_apply%d' % l.idx] = n * (dab + yb + dzb)
        by['bn_bwd_reduce%d' % l.idx] = n * (dab + yb)
    return by


def family_rooflines(bank, B, per, precision):
    """roofline.wgrad (matrix cores, or HBM for the bf16 path) and roofline.bn_bwd (HBM) from the same eager event steps."""
    out = {}
    fx, fa = wgrad_exec_flops(bank, B)
    wb = wgrad_bytes(bank, B)
    fam = [k for k in per if k in fx]
    t, n = sum(sum(per[k]) for k in fam), sum(len(per[k]) for k in fam)
    if n and t > 0:
        steps_ev = max(len(per[k]) for k in fam)
        x = sum(fx[k] * len(per[k]) for k in fam)
        a = sum(fa[k] * len(per[k]) for k in fam)
        b = sum(wb[k] * len(per[k]) for k in fam)
        red = [k for k in per if k.startswith('wgrad_reduce') or k.startswith('wgradT_reduce')]
        tr = sum(sum(per[k]) for k in red)
        r = {'launches_timed': n, 'avg_launch_us': 1e6 * t / n, 'family_ms_per_step': 1e3 * t / steps_ev,
             'slab_reductions_ms_per_step': 1e3 * tr / steps_ev, 'algorithmic_tflops': a / t / 1e12,
             'algorithmic_mbytes_per_launch': b / n / 1e6, 'hbm_gbytes_per_s_algorithmic': b / t / 1e9}
        if precision == 'fp32':
            r.update({'bound': 'mfma', 'kernel': 'wgrad_wino_kernel (3x3 layers, Winograd F(2x2,3x3) weight gradient) + wgrad_mfma_kernel '
                      '(transposed convs, direct 9-tap) on v_mfma_f32_32x32x2_f32', 'achieved': x / t / 1e12, 'peak': FP32_MFMA_PEAK / 1e12,
                      'unit': 'TFLOP/s', 'frac': x / t / FP32_MFMA_PEAK, 'accounting': 'executed multiply-adds (as the headline roofline)'})
        else:
            r.update({'bound': 'hbm', 'kernel': 'wgrad_ring_kernel / wgrad_bf16_kernel (bf16 operands, fp32 accumulation)', 'achieved': b / t / 1e9,
                      'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': b / t / HBM_PEAK, 'mfma_tflops': a / t / 1e12})
        out['wgrad'] = r
    bb = bn_bwd_bytes(bank, B)
    fam = [k for k in per if k in bb]
    t, n = sum(sum(per[k]) for k in fam), sum(len(per[k]) for k in fam)
    if n and t > 0:
        steps_ev = max(len(per[k]) for k in fam)
        b = sum(bb[k] * len(per[k]) for k in fam)
        ap = [k for k in fam if 'apply' in k]
        out['bn_bwd'] = {'bound': 'hbm', 'kernel': 'bn_bwd_apply_kernel (dz = f(dA, z, sums): reads dA + z, writes dz) + bn_bwd_reduce_kernel (the layers '
                         'whose sums no producing launch leaves behind: reads dA + z)', 'achieved': b / t / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': b / t / HBM_PEAK, 'launches_timed': n, 'avg_launch_us': 1e6 * t / n, 'family_ms_per_step': 1e3 * t / steps_ev,
                         'apply_launches_per_step': len(ap), 'reduce_launches_per_step': len(fam) - len(ap),
                         'algorithmic_mbytes_per_launch': b / n / 1e6}
    return out


_F44_BYTES = {}       # profile name -> HBM bytes per launch of the F(4x4) launches of the same counter passes, or None
_STEP_BYTES = {}      # profile name -> whole-step HBM bytes of the same counter passes (train steps only), or None


def pmc_traffic(name):
    """HBM bytes per launch of the conv family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
    separately, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); None when no such profile is committed.  This is a
    number read from profiles/, not something this run measured (PMC collection needs rocprofv3 around the process)."""
    for n in [name] + [name.replace('r06_', 'r0%d_' % k) for k in (5, 4, 3, 2, 1)]:
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', n)))
            _STEP_BYTES[name] = d.get('hbm_bytes_per_train_step')
            _F44_BYTES[name] = (d.get('conv_family_f4x4') or {}).get('hbm_bytes_per_launch_corrected')
            return (d.get('conv_family') or d['conv_mfma_family'])['hbm_bytes_per_launch_corrected'], _profile_tag(n, d)
        except Exception:
            continue
    return None, None


def _profile_tag(name, d):
    """'<file> (counters taken on library build X; this run: X = same kernels | Y = STALE)' -- a constant read from profiles/ must
    say which kernels it was measured on."""
    try:
        from vec_vad_amd import build as B
        cur = B.wanted()[1][:16]
    except Exception:
        cur = None
    was = d.get('library_build')
    if was is None:
        return '%s (library build of the counters not recorded)' % name
    return '%s (counters taken on library build %s; this run %s: %s)' % (name, was, cur, 'same kernels' if was == cur else 'STALE')


def _cpu_topology():
    """(model name, physical cores, hardware threads usable by this process) from /proc/cpuinfo."""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, (len(cores) or usable), usable


def cpu_child(nt, batch, warm, steps, limit_s):
    """One leg of the CPU baseline in its own process (threads bound by the parent's OMP_* environment): prints one JSON line.
    Timed steps stop early when ``limit_s`` is used up (at least 2 are timed) and the line says how many ran."""
    torch.set_num_threads(nt)
    from oracle import unet_oracle as O
    spec = O.bank_spec('net4')
    raw, flow = O.seeded_cubes(batch, 1, 3)
    x, x_of = O.cubes_to_inputs(raw, flow)
    sd = O.seeded_state_dict('net4', nf=32, padding=False, seed=0)
    opt = O.AdamState(O.param_names(sd))
    t_start = time.perf_counter()
    for _ in range(warm):
        O.train_step(sd, spec, x, x_of, opt)
        if time.perf_counter() - t_start > 0.4 * limit_s:
            break
    t0 = time.perf_counter()
    n = 0
    while n < steps and (n < 2 or time.perf_counter() - t_start < limit_s):
        O.train_step(sd, spec, x, x_of, opt)
        n += 1
    dt = time.perf_counter() - t0
    # eval-mode forward (test.py:319-335)
    e0 = time.perf_counter()
    m = 0
    while m < steps and (m < 2 or time.perf_counter() - e0 < 0.25 * limit_s):
        O.score_pass(sd, spec, x, x_of, batch)
        m += 1
    de = time.perf_counter() - e0
    print(json.dumps({'threads': nt, 'train_cubes_per_s': batch * n / dt, 'timed_steps': n, 'eval_cubes_per_s': batch * m / de}))


def cpu_baseline(batch=32, warm=3, steps=10, budget_s=55.0):
    """Reference op sequence (torch CPU ops, NCHW fp32, per-op BN/ReLU/pool/cat, Adam eps=1e-7) on the host cores:
    BASELINE.json configs[0] (Net4, B=32), SURVEY 8(d): 3 warm-up + 10 timed train steps per leg.  Every leg runs in its own
    process with its OpenMP threads BOUND (OMP_PROC_BIND=close, OMP_PLACES=cores): unbound, the 256-thread host of an MI355X box
    thrashed at its full thread count (round 2: the all-cores leg never finished).  Legs: 8, 16, 32, 64 ... up to the physical core
    count; ``value`` / ``cores`` = the best leg, ``all_cores`` = the leg on every physical core.  This is the oracle restatement
    ("port"): test / bench infrastructure, never part of the product path."""
    import subprocess
    model, ncore, usable = _cpu_topology()
    top = max(1, min(ncore, usable))
    legs = sorted({t for t in (8, 16, 32, 64, 128, 256) if t < top} | {top})
    t_start = time.perf_counter()
    res = {}
    for i, nt in enumerate(legs):
        left = budget_s - (time.perf_counter() - t_start)
        per = max(6.0, left / (len(legs) - i))
        env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PROC_BIND='close', OMP_PLACES='cores', MKL_NUM_THREADS=str(nt))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', str(nt), '--cpu-child-limit', '%.1f' % per,
                                '--batch', str(batch)], capture_output=True, text=True, timeout=per * 2.5 + 20, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            res[nt] = json.loads(line[-1]) if line else {'error': r.stderr[-200:]}
        except subprocess.TimeoutExpired:
            res[nt] = {'error': 'timeout after %.0f s' % (per * 2.5 + 20)}
    ok = {nt: v for nt, v in res.items() if 'train_cubes_per_s' in v}
    if not ok:
        raise RuntimeError('no CPU leg finished: %r' % (res,))
    best = max(ok, key=lambda nt: ok[nt]['train_cubes_per_s'])
    allc = res.get(top, {})
    sweep = '; '.join('%d thr: %s' % (nt, ('%.0f (%d steps)' % (v['train_cubes_per_s'], v['timed_steps'])) if 'train_cubes_per_s' in v else v['error'])
                      for nt, v in sorted(res.items()))
    return {'value': ok[best]['train_cubes_per_s'], 'unit': 'cubes/s', 'cores': best, 'kind': 'port',
            'cpu_model': model, 'physical_cores': ncore, 'hardware_threads': os.cpu_count(), 'usable_threads': usable,
            'all_cores': {'value': allc.get('train_cubes_per_s'), 'cores': top, 'timed_steps': allc.get('timed_steps'),
                          'note': allc.get('error')},
            'eval_value': ok[best]['eval_cubes_per_s'], 'eval_unit': 'cubes/s (eval-mode forward + per-cube scores)',
            'sample': 'SelfCompleteNet4 train step (fwd+bwd+Adam) on stock torch %s CPU ops, batch %d (BASELINE configs[0]), %d warm-up + up to '
                      '%d timed steps per leg, one process per leg with OMP_PROC_BIND=close OMP_PLACES=cores; %s, %d physical cores / %d '
                      'hardware threads; legs (cubes/s): %s' % (torch.__version__, batch, warm, steps, model, ncore, os.cpu_count() or 0, sweep)}


def build_net(model, precision, dev):
    from model.unet import SelfCompleteNet4, SelfCompleteNetFull
    os.environ['VV_PRECISION'] = precision          # read by the UNet bank when it is built
    torch.manual_seed(0)
    tot_of = 1 if model == 'net4' else 5
    cls = SelfCompleteNet4 if model == 'net4' else SelfCompleteNetFull
    net = cls(features_root=32, tot_raw_num=5, tot_of_num=tot_of, border_mode='predict', rawRange=None, useFlow=True,
              padding=False).to(dev)
    return net, tot_of


def conv_roofline(bank, B, per, precision, overlap, traffic):
    """roofline object of the 3x3-conv family from HIP-event timings {label: [seconds]}."""
    fl = conv_flops(bank.lay, B, bank.Ga)
    fx = conv_exec_flops(bank.lay, B, bank.Ga, bank.wino, lambda l, d: bank._w44(B, l, d))      # (train-mode plan: VV_WINO44, default: data gradients)
    by = conv_bytes(bank.lay, B, bank.Ga, 2 if bank.y16 else 4, 2 if bank.dz16 else 4, 2 if bank.da16 else 4)
    # fp32: the launches the bank routes to Winograd F(4x4,3x3) (vv_conv_wino44, another kernel executing x1/4 instead of x16/36 of the
    # direct multiply-adds) are reported beside the dominant F(2x2) kernel family, not mixed into its roofline
    w44k = [k for k in per if k in fl and bank._w44(B, bank.lay.convs[int(k.lstrip('convdgrad'))], k.startswith('dgrad'))] \
        if (precision == 'fp32' and bank.wino) else []
    fam = [k for k in per if k in fl and k not in w44k]
    t = sum(sum(per[k]) for k in fam)
    n = sum(len(per[k]) for k in fam)
    if not n or t <= 0:
        return None
    f_alg = sum(fl[k] * len(per[k]) for k in fam)
    f_exe = sum(fx[k] * len(per[k]) for k in fam)
    b_alg = sum(by[k] * len(per[k]) for k in fam)
    common = {'launches_timed': n, 'avg_launch_us': 1e6 * t / n, 'algorithmic_gflop_per_launch': f_alg / n / 1e9,
              'algorithmic_mbytes_per_launch': b_alg / n / 1e6, 'traffic': traffic[0], 'traffic_source': traffic[1],
              # the launches above were timed in eager steps on ONE stream (overlap == 'none') or on the eager two-stream
              # executor (--overlap); the graph-replayed steps of the timed region follow execution.backward_schedule
              'timed_on': 'one stream (eager event steps)' if overlap == 'none' else 'eager two-stream executor (%s)' % overlap}
    if precision == 'fp32':
        r = {'bound': 'mfma',
             'kernel': ('wino_conv_kernel<H> + wino_ring_kernel<KQ> (32x32 level, K <= 32: persistent, LDS-DMA ring): 3x3 conv forward + '
                        'data-gradient as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32'
                        if bank.wino else 'conv_mfma_kernel: 3x3 implicit GEMM on v_mfma_f32_32x32x2_f32, forward + data-gradient'),
             'achieved': f_exe / t / 1e12, 'peak': FP32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': f_exe / t / FP32_MFMA_PEAK,
             'accounting': 'achieved / frac = multiply-adds the matrix cores execute (x16/36 of the direct convolution for the '
                           'Winograd form, K padded to the MFMA granule); algorithmic_tflops = SURVEY 8(d) direct-convolution FLOP / time',
             'algorithmic_tflops': f_alg / t / 1e12, 'effective_vs_direct': f_alg / f_exe}
        if w44k:
            t4 = sum(sum(per[k]) for k in w44k)
            n4 = sum(len(per[k]) for k in w44k)
            x4, a4 = sum(fx[k] * len(per[k]) for k in w44k), sum(fl[k] * len(per[k]) for k in w44k)
            r['f4x4_launches'] = {'kernel': 'wino44_conv_kernel<H, NS>: the same convolution as Winograd F(4x4,3x3) (x1/4 of the direct multiply-adds) on '
                                            + ', '.join(sorted(w44k)), 'launches_timed': n4, 'avg_launch_us': 1e6 * t4 / n4,
                                  'achieved': x4 / t4 / 1e12, 'frac': x4 / t4 / FP32_MFMA_PEAK, 'algorithmic_tflops': a4 / t4 / 1e12,
                                  'algorithmic_mbytes_per_launch': sum(by[k] * len(per[k]) for k in w44k) / n4 / 1e6,
                                  'traffic': _F44_BYTES.get('r06_pmc_hbm_traffic.json')}
            r['all_3x3_launches'] = {'launches_timed': n + n4, 'avg_launch_us': 1e6 * (t + t4) / (n + n4),
                                     'executed_frac': (f_exe + x4) / (t + t4) / FP32_MFMA_PEAK,
                                     'algorithmic_tflops': (f_alg + a4) / (t + t4) / 1e12}
        if bank.wino and getattr(bank, 'fuse_bn_sums', False):
            r['also_in_these_launches'] = ('the data-gradient launches whose output has a single consumer (7 of 13 per step) also do the '
                                           'first reduction pass of that BatchNorm backward in their epilogue (z read + two sums per value); '
                                           'VV_FUSE_BN_SUMS=0 times the bare convolutions (frac +0.013, step +0.15 ms)')
    else:
        r = {'bound': 'hbm',
             'kernel': 'conv_gemm16p_kernel (16x16 / 8x8 / 4x4 levels: persistent producer / consumer GEMM-shaped kernel, round 4) + '
                       'conv_mfma_kernel<..., BF=true> (32x32 level): 3x3 conv forward + data-gradient, bf16 operands on '
                       'v_mfma_f32_32x32x16_bf16, fp32 accumulation; achieved = algorithmic bytes (input once + output once, bf16 tensors) / time',
             'achieved': b_alg / t / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': b_alg / t / HBM_PEAK,
             'mfma_tflops': f_alg / t / 1e12, 'frac_of_bf16_mfma_peak': f_alg / t / BF16_MFMA_PEAK,
             # what a loop of nothing but v_mfma_f32_32x32x16_bf16 with this kernel's operand traffic sustains on this chip
             # (profiles/r04_gemm16_loop_calibration.txt: 1.55 - 1.57 PFLOP/s; the chip clocks down under dense bf16 MFMA)
             'frac_of_calibrated_bf16': f_alg / t / BF16_MFMA_CALIBRATED, 'calibrated_bf16_tflops': BF16_MFMA_CALIBRATED / 1e12}
    r.update(common)
    return r


def run_unet(model, precision, B, steps, warmup, dev, rank, world, dist, overlap='none', breakdown=False, pool=4096,
             measure_forward=True, graph=True, ev_steps=1, small_batch_diag=False, ev_extra=4):
    """W untimed + exactly K timed train steps; returns the record (rank 0) -- value is the whole-job rate.

    graph=True (default): the train step is replayed from a hipGraph (FusedTrainer, captured during the warm-up).  The first
    ``ev_steps`` of the K timed steps run the eager launch loop with HIP events around every launch of the dominant kernel family
    (the roofline's live per-launch durations); the remaining steps replay the captured step.  ``ev_extra`` further eager event
    steps run right AFTER the timed region (same trainer, next batches) so that the roofline rests on (ev_steps + ev_extra) x 27
    launches instead of one sample per layer -- the record says how many were inside and how many after.  graph=False /
    --breakdown / a side-stream schedule: every timed step is eager with events (the round-2 behaviour)."""
    from vec_vad_amd.trainer import FusedTrainer
    net, tot_of = build_net(model, precision, dev)
    # VV_FORCE_DIST=1 (a one-rank group under torch.distributed.run --nproc-per-node 1): the bucketed exchange runs anyway -- the
    # command line, backend init, hipGraph segments and in-place collectives of the N > 1 run, on one GPU
    trainer = FusedTrainer(net, lr=1e-3, eps=1e-7, process_group=dist.group.WORLD if dist is not None else None,
                           overlap={'none': False, 'free': True, 'paired': 'paired'}[overlap],
                           always_bucket=dist is not None and world == 1)
    bank = trainer.bank
    graph = bool(graph and trainer.use_graph and overlap == 'none' and not breakdown)
    trainer.use_graph = graph
    if graph:
        warmup = max(warmup, 3)         # eager step (builds plans), capturing step, first replay
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    raw = torch.randint(0, 256, (pool, 5, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
    flow = (torch.randn((pool, tot_of, 32, 32, 2), generator=g) * 2.0).to(dev)
    ev_extra = ev_extra if graph else 0
    perm = torch.stack([torch.randperm(pool, generator=g)[:B] for _ in range(steps + warmup + ev_extra)]).to(dev)
    for it in range(warmup):
        trainer.step_cubes(raw, flow, perm[it])
    torch.cuda.synchronize()
    ws = bank.workspace(B)
    fl = conv_flops(bank.lay, B, bank.Ga)
    ev = []
    hook = lambda label, a, b: ev.append((label, a, b))
    n_ev = steps if not graph else min(ev_steps, steps)
    trainer.event_hook = hook if n_ev > 0 else None
    fam_labels = set(wgrad_exec_flops(bank, B)[0]) | set(bn_bwd_bytes(bank, B)) | {'wgrad_reduce0', 'wgrad_reduce4', 'wgradT_reduce0'}
    trainer.event_labels = (set(fl.keys()) | fam_labels) if not breakdown else None
    diag_comm = trainer.buckets is not None and not graph
    if diag_comm:                      # per-bucket collective timings need the eager loop (events between the launches)
        trainer.buckets.timing = []
    if trainer.buckets is not None:    # exposed communication: one event pair around the last bucket + the wait for all three, recorded
        trainer.comm_timing = []       # between the hipGraph segments (graph mode) or behind the last backward launch (eager loop)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        if it == n_ev:
            trainer.event_hook = None          # from here on: hipGraph replay
        trainer.step_cubes(raw, flow, perm[warmup + it])
    t_host = time.perf_counter() - t0          # the host has enqueued everything (no sync yet)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_local = dt
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tl = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allt, tl)
        per_rank = [B * steps / float(x.item()) for x in allt]
    n_in = len(ev)
    if ev_extra and n_ev > 0:          # more samples for the roofline, outside the timed region (the step itself is unchanged)
        trainer.event_hook = hook
        for it in range(ev_extra):
            trainer.step_cubes(raw, flow, perm[warmup + steps + it])
        torch.cuda.synchronize()
    trainer.event_hook = None
    comm = None
    if trainer.buckets is not None and not diag_comm:
        exposed = [a.elapsed_time(b) * 1e3 for a, b in (trainer.comm_timing or [])]
        comm = {'ranks': world, 'backend': dist.get_backend() if dist is not None else None,
                'buckets': [{'bucket': k, 'columns': [trainer.buckets.bounds[k], trainer.buckets.bounds[k + 1]],
                             'mbytes': 4e-6 * bank.G * (trainer.buckets.bounds[k + 1] - trainer.buckets.bounds[k])} for k in (2, 1, 0)],
                'exposed_comm_us_per_step': sum(exposed) / len(exposed) if exposed else None,
                'exposed_comm_steps_timed': len(exposed),
                'note': 'in-place all-reduce of three contiguous ranges of the bucket-major gradient buffer, launched between the '
                        'hipGraph segments of the step; exposed = main-stream time from the end of the last backward segment to the arrival of '
                        'the last sums (one HIP event pair around _finish_exchange, outside every captured segment: launch of the 3 % bucket + '
                        'wait for all three); per-bucket timings need the eager loop: bench.py --no-graph'}
        trainer.comm_timing = None
    if trainer.buckets is not None and diag_comm:
        bt = {}
        for k, a, b in trainer.buckets.timing:
            bt.setdefault(k, []).append(a.elapsed_time(b) * 1e3)
        exposed = [a.elapsed_time(b) * 1e3 for a, b in trainer.comm_timing]
        lay = bank.lay
        bounds = trainer.buckets.bounds
        comm = {'ranks': world, 'backend': dist.get_backend() if dist is not None else None,
                'buckets': [{'bucket': k, 'columns': [bounds[k], bounds[k + 1]], 'mbytes': 4e-6 * bank.G * (bounds[k + 1] - bounds[k]),
                             'allreduce_us_avg': sum(v) / len(v), 'launched_after': {2: 'decoder half of backward', 1: 'deep-encoder '
                             'weight gradients', 0: 'last backward launch'}[k]} for k, v in sorted(bt.items(), reverse=True)],
                'exposed_comm_us_per_step': sum(exposed) / max(1, len(exposed)),
                'note': 'allreduce_us = in-place collective on the communication stream (HIP events); exposed = main-stream '
                        'time from the end of the backward pass to the arrival of the last sums'}
        trainer.buckets.timing = None
        trainer.comm_timing = None
    # forward-only rate (north_star: ">= 50 % of the MFMA roofline on the UNet forward at batch 256"): cube gather + train-mode
    # forward (BatchNorm batch statistics, loss + per-cube scores), outside the timed region
    fwd_ms = None
    if measure_forward:
        fwd_n = 10
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bufs, nbt = bank.bufs.clone(), bank.nbt.clone()      # forward(train=True) moves the BatchNorm running statistics
        f0.record()
        for it in range(fwd_n):
            bank.set_input_cubes(raw, flow, perm[it % perm.shape[0]], B)
            bank.forward(ws, True)
        f1.record()
        torch.cuda.synchronize()
        bank.bufs.copy_(bufs)
        bank.nbt.copy_(nbt)
        fwd_ms = f0.elapsed_time(f1) / fwd_n
    l_raw, l_of = bank.losses(ws)
    per = {}
    for label, a, b in ev:
        per.setdefault(label, []).append(a.elapsed_time(b) * 1e-3)
    if breakdown and rank == 0:
        tot = sum(sum(v) for v in per.values())
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            extra = '  %.1f TF/s' % (fl[k] / (sum(v) / len(v)) / 1e12) if k in fl else ''
            sys.stderr.write('%-22s n=%3d avg %8.1f us  %5.1f%%%s\n' % (k, len(v), 1e6 * sum(v) / len(v), 100 * sum(v) / tot, extra))
        sys.stderr.write('sum of launches %.3f ms / step ; wall %.3f ms / step\n' % (1e3 * tot / steps, 1e3 * dt / steps))
    fwd_flop, train_flop = FLOP[model]
    value = B * world * steps / dt
    peak = FP32_MFMA_PEAK if precision == 'fp32' else BF16_MFMA_PEAK
    tag = 'fp32' if precision == 'fp32' else 'bf16'
    # the committed PMC passes: default workload (net4, B=256) and BASELINE config 4 (full, B=512, bf16)
    pname = None
    if model == 'net4' and B == 256:
        pname = 'r06_pmc_hbm_traffic%s.json' % ('' if precision == 'fp32' else '_bf16')
    elif model == 'full' and B == 512 and precision == 'bf16':
        pname = 'r06_pmc_hbm_traffic_bf16_full_b512.json'
    traffic = pmc_traffic(pname) if pname else (None, None)
    rec = {'value': value, 'unit': 'cubes/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
           'dtype': 'f32' if precision == 'fp32' else 'bf16 operands, f32 accumulate',
           'config': {'workload': {'net4': 'UCSDped2-shaped 5raw+1of UNet bank (SelfCompleteNet4, nf=32, padding=False) train step: '
                                           'cube gather + forward + backward + Adam(eps=1e-7)',
                                   'full': 'ShanghaiTech-shaped 5raw+5of UNet bank (SelfCompleteNetFull, nf=32, padding=False) train '
                                           'step: cube gather + forward + backward + Adam(eps=1e-7)'}[model],
                      'batch_per_gpu': B, 'global_batch': B * world, 'cube': '32x32x5 RGB uint8 + %d flow map(s) fp32' % tot_of,
                      'parallelism': 'dp%d' % world, 'train_tflops_per_gpu': value / world * train_flop / 1e12,
                      'frac_of_%s_mfma_peak_whole_step_algorithmic' % tag: value / world * train_flop / peak,
                      'loss_raw': float(l_raw), 'loss_of': float(l_of) if l_of is not None else 0.0},
           'roofline': conv_roofline(bank, B, per, precision, overlap, traffic)}
    if rec['roofline'] is not None:
        rec['roofline'].update(family_rooflines(bank, B, per, precision))
    if rec['roofline'] is not None and pname and _STEP_BYTES.get(pname):
        # whole train step, same counter passes: HBM bytes per step and the rate that implies at this run's step time
        rec['roofline']['step_traffic_bytes'] = _STEP_BYTES[pname]
        rec['roofline']['step_hbm_gbytes_per_s'] = _STEP_BYTES[pname] / (dt / steps) / 1e9
    if rec['roofline'] is not None and graph:
        f44 = lambda label: precision == 'fp32' and bank.wino and bank._w44(B, bank.lay.convs[int(label.lstrip('convdgrad'))], label.startswith('dgrad'))
        n_fam = sum(1 for label, _, _ in ev[:n_in] if label in fl and not f44(label))
        rec['roofline']['launches_timed_where'] = ('%d inside the timed region (its first %d step(s) run the eager loop), %d in %d further '
                                                   'eager event step(s) right after it' % (n_fam, n_ev, rec['roofline']['launches_timed'] - n_fam, ev_extra))
    if precision == 'bf16':
        rec['config']['precision'] = 'mixed bf16 (BASELINE config 4): conv / transposed-conv operands and stored activations bf16, ' \
                                     'parameters / BatchNorm statistics / loss / Adam fp32'
    if fwd_ms is not None:
        rec['config'].update({'forward_ms': fwd_ms, 'forward_tflops_per_gpu_algorithmic': B * fwd_flop / (fwd_ms * 1e-3) / 1e12,
                              'forward_frac_of_%s_mfma_peak_algorithmic' % tag: B * fwd_flop / (fwd_ms * 1e-3) / peak})
        if precision == 'fp32' and bank.wino:
            # executed share of the forward: the 14 Winograd conv launches execute 16/36, everything else (transposed convs) 1:1
            fa = conv_flops(bank.lay, B, bank.Ga)
            fx = conv_exec_flops(bank.lay, B, bank.Ga, True, lambda l, d: bank._w44(B, l, d))
            conv_a = sum(v for k, v in fa.items() if k.startswith('conv'))
            conv_x = sum(v for k, v in fx.items() if k.startswith('conv'))
            exe = B * fwd_flop - conv_a + conv_x
            rec['config']['forward_frac_of_fp32_mfma_peak_executed'] = exe / (fwd_ms * 1e-3) / peak
    if comm is not None:
        if dist is not None:
            comm['per_rank_cubes_per_s'] = {'min': min(per_rank), 'max': max(per_rank)}
        rec['comm'] = comm
    cap = next((c for k, c in trainer._graphs.items() if k[0] == 'train' and c != 'warm'), None)
    rec['execution'] = {'mode': ('hipGraph replay of the captured step (%d launches in %d segment(s)); timed steps 0..%d ran the eager '
                                 'launch loop with HIP events' % (cap.launches, len(cap.segments), n_ev - 1)) if (graph and cap is not None)
                        else 'eager launch loop (ctypes), HIP events around the conv-family launches in every timed step',
                        'launches_per_step': cap.launches if cap is not None else len(ws.fwd[True].calls) + len(getattr(ws, "bwd_cur", ws.bwd).calls) + 3,
                        'host_enqueue_ms_per_step': 1e3 * t_host / steps}
    if graph and cap is not None:
        # 'free': the captured backward pass has two branches (weight gradients beside the data-gradient / BatchNorm chain), so
        # kernels of the replayed steps overlap; the roofline durations above come from the one-stream eager steps
        rec['execution']['backward_schedule'] = getattr(cap, 'schedule', 'one stream')
    if small_batch_diag and graph:
        # the same workload on the eager launch loop, for the launch-overhead comparison (outside the timed region)
        trainer.use_graph = False
        n = min(steps, 30)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for it in range(n):
            trainer.step_cubes(raw, flow, perm[it])
        th = time.perf_counter() - t1
        torch.cuda.synchronize()
        rec['execution'].update({'eager_ms_per_step': 1e3 * (time.perf_counter() - t1) / n, 'eager_host_loop_ms_per_step': 1e3 * th / n})
    del trainer, net, raw, flow
    torch.cuda.empty_cache()
    return rec


def run_scoring(dev, B=2048, n=8192, reps=3):      # B = config.cfg's [mi355x] score_batch
    """Eval-mode scoring pass (test.py:312-345 / train.py:413-427): device-resident cubes -> per-cube raw / flow scores."""
    from vec_vad_amd.trainer import FusedTrainer
    net, tot_of = build_net('net4', 'fp32', dev)
    net.eval()
    tr = FusedTrainer(net)
    g = torch.Generator(device='cpu').manual_seed(7)
    raw = torch.randint(0, 256, (n, 5, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
    flow = (torch.randn((n, tot_of, 32, 32, 2), generator=g) * 2.0).to(dev)
    idx = [torch.arange(s, s + B, device=dev) for s in range(0, n, B)]
    for i in idx[:2]:
        tr.score_cubes(raw, flow, i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in idx:
            tr.score_cubes(raw, flow, i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * len(idx))
    fwd_flop = FLOP['net4'][0]
    rec = {'value': B / (ms * 1e-3), 'unit': 'cubes/s', 'ms_per_launch_of_%d_cubes' % B: ms, 'dtype': 'f32',
           'workload': 'SelfCompleteNet4 eval-mode scoring (running-statistics BatchNorm, per-cube squared-error sums), %d cubes per launch' % B,
           'path': getattr(tr.bank, 'eval_path', 'train-mode kernel family with running statistics'),
           'algorithmic_tflops': B * fwd_flop / (ms * 1e-3) / 1e12,
           'frac_of_fp32_mfma_peak_algorithmic': B * fwd_flop / (ms * 1e-3) / FP32_MFMA_PEAK}
    # one accounting with the headline's roofline: the multiply-adds the matrix cores EXECUTE (Winograd 3x3 layers x16/36, K padding
    # counted; transposed convs and the 1x1 output conv as they are)
    bank = tr.bank
    w44 = lambda l, d: bank._w44(B, l, d, evalm=True)
    exe = sum(v for k, v in conv_exec_flops(bank.lay, B, bank.Ga, bank.wino, w44).items() if k.startswith('conv'))
    rec['winograd_f4x4_layers'] = [l.idx for l in bank.lay.convs if w44(l, False)]
    exe += sum(2.0 * B * H * H * 9 * ci * co * bank.Ga for (_, H, ci, co) in bank.lay.convT)
    exe += 2.0 * B * 32 * 32 * bank.nf * sum(u.out_c for u in bank.units[bank.g0:bank.g0 + bank.Ga])      # 1x1 output convs (VALU)
    rec['executed_tflops'] = exe / (ms * 1e-3) / 1e12
    rec['frac_executed'] = exe / (ms * 1e-3) / FP32_MFMA_PEAK
    rec['roofline'] = {'bound': 'mfma', 'achieved': rec['executed_tflops'], 'peak': FP32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                       'frac': rec['frac_executed'], 'accounting': 'executed multiply-adds (as the headline roofline)'}
    del tr, net
    torch.cuda.empty_cache()
    return rec


def _fn2_traffic():
    for n in ('r06_pmc_hbm_traffic_flownet2.json', 'r05_pmc_hbm_traffic_flownet2.json', 'r04_pmc_hbm_traffic_flownet2.json', 'r03_pmc_hbm_traffic_flownet2.json'):
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', n)))
            return d['total_hbm_bytes_per_run_corrected'], _profile_tag(n, d)
        except Exception:
            continue
    return None, None


def run_flownet2(dev, reps=10):
    """BASELINE config 5: FlowNet2 forward on one 1024x436 pair zero-padded to 1024x448 (the reference itself fails at 436),
    xavier weights (no checkpoint offline), hipGraph replay; per-kernel-family timings from one eager pass with HIP events."""
    from vec_vad_amd.flownet2 import FlowNet2
    torch.manual_seed(0)
    net = FlowNet2().to(dev).eval()
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(1, 3, 2, 448, 1024, generator=g) * 255)
    x[:, :, :, 436:] = 0
    x = x.to(dev)
    for _ in range(2):
        out = net.forward_graphed(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        out = net.forward_graphed(x)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms = e0.elapsed_time(e1) / reps
    # per-family launch timings (eager; events bracket single launches)
    fam = {}
    net._runner.hook = lambda label, flop, a, b: fam.setdefault(label, []).append((flop, a, b))
    overlap = os.environ.get('VV_FN2_OVERLAP')
    os.environ['VV_FN2_OVERLAP'] = '0'          # one stream: a launch's events then bracket that launch alone
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    net._runner.hook = None
    if overlap is None:
        del os.environ['VV_FN2_OVERLAP']
    else:
        os.environ['VV_FN2_OVERLAP'] = overlap
    fams = {}
    exe_gflop = 0.0                     # multiply-adds the matrix cores execute per forward: Winograd launches x16/36
    for k, v in fam.items():
        t = sum(a.elapsed_time(b) for _, a, b in v) * 1e-3
        f = sum(fl for fl, _, _ in v)
        fx = f * (16.0 / 36.0 if k.endswith('_wino') else 1.0)
        exe_gflop += fx / 2 / 1e9
        fams[k] = {'launches_per_forward': len(v) // 2, 'ms_per_forward': 1e3 * t / 2, 'tflops': f / t / 1e12 if t > 0 else None,
                   'tflops_executed': fx / t / 1e12 if t > 0 else None}
    dom = max((k for k in fams if not k.endswith('_n2')), key=lambda k: fams[k]['ms_per_forward'])
    # throughput form: 4 pairs per launch (calc_optical_flow.py's default) -- at one pair the H/16 ... H/64 levels leave most CUs idle
    net._graphs.clear()
    xb = x.expand(4, -1, -1, -1, -1).contiguous()
    for _ in range(2):
        net.forward_graphed(xb)
    torch.cuda.synchronize()
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(5):
        net.forward_graphed(xb)
    b1.record()
    torch.cuda.synchronize()
    ms4 = b0.elapsed_time(b1) / 5 / 4
    rec = {'value': 1e3 / ms, 'unit': 'pairs/s', 'ms_per_pair': ms, 'ms_per_pair_wall': wall * 1e3, 'dtype': 'f32',
           'workload': 'FlowNet2 forward, one 1024x436 pair zero-padded to 1024x448, xavier weights, hipGraph replay',
           'algorithmic_gflop': FLOWNET2_GFLOP, 'finite': bool(torch.isfinite(out).all()),
           'executed_gflop': exe_gflop,
           'roofline': {'bound': 'mfma', 'kernel': 'conv2d_mfma_kernel + conv2d_wino_kernel (whole forward: 464.2 GFLOP of conv / deconv work, '
                                                   '%.1f GFLOP executed: the Winograd layers run 16/36 of their multiply-adds)' % exe_gflop,
                        'achieved': exe_gflop / ms, 'peak': FP32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                        'frac': exe_gflop / ms / (FP32_MFMA_PEAK / 1e12),
                        'accounting': 'achieved / frac = EXECUTED multiply-adds, like the headline roofline (frac_algorithmic beside it)',
                        'algorithmic_tflops': FLOWNET2_GFLOP / ms, 'frac_algorithmic': FLOWNET2_GFLOP / ms / (FP32_MFMA_PEAK / 1e12),
                        'traffic': _fn2_traffic()[0],
                        'traffic_source': _fn2_traffic()[1], 'traffic_unit': 'HBM bytes per forward pass (all kernels)',
                        'dominant_family': dom, 'dominant_family_frac': fams[dom]['tflops_executed'] / (FP32_MFMA_PEAK / 1e12),
                        'dominant_family_frac_algorithmic': fams[dom]['tflops'] / (FP32_MFMA_PEAK / 1e12)},
           'four_pairs_per_launch': {'ms_per_pair': ms4, 'pairs_per_s': 1e3 / ms4, 'tflops': FLOWNET2_GFLOP / ms4,
                                     'frac': exe_gflop / ms4 / (FP32_MFMA_PEAK / 1e12),
                                     'frac_algorithmic': FLOWNET2_GFLOP / ms4 / (FP32_MFMA_PEAK / 1e12)},
           'families_eager': fams}
    del net
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='cubes per GPU per step (32 with --gpus 8 = DataParallel split of 256)')
    ap.add_argument('--dp-split-of', type=int, default=256, help='N > 1: global batch of the second, DataParallel-faithful record (one batch of '
                    'this size split over the ranks, train.py:373-375); 0 = headline only')
    ap.add_argument('--pool', type=int, default=4096, help='device-resident synthetic cubes per GPU')
    ap.add_argument('--model', default='net4', choices=['net4', 'full'])
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="bf16 = BASELINE config 4's mixed precision; the headline number is fp32, like the reference")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-child', type=int, default=0, help=argparse.SUPPRESS)          # one leg of the CPU baseline (internal)
    ap.add_argument('--cpu-child-limit', type=float, default=10.0, help=argparse.SUPPRESS)
    ap.add_argument('--no-graph', action='store_true', help='eager launch loop in every timed step (default: hipGraph replay)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the config-4 / config-5 / scoring records')
    ap.add_argument('--no-forward-timing', action='store_true', help='skip the forward-only passes (counter profiles: the process '
                    'then runs train steps alone, so whole-process HBM bytes / Adam launches = bytes per step)')
    ap.add_argument('--breakdown', action='store_true', help='print a per-launch time table to stderr')
    ap.add_argument('--overlap', nargs='?', const='free', default='none', choices=('none', 'free', 'paired'),
                    help="side stream for the weight-gradient kernels: 'free' = under everything that follows (conv launches "
                         "are then contended), 'paired' = only under the next layer's BatchNorm backward (conv launches run alone)")
    args = ap.parse_args()
    if args.cpu_child:
        cpu_child(args.cpu_child, args.batch if args.batch != 256 else 32, 3, 10, args.cpu_child_limit)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    local_rank_dev = 0 if os.environ.get('VV_SINGLE_DEVICE') else local_rank      # bring-up: all ranks on GPU 0 (gloo only)
    torch.cuda.set_device(local_rank_dev)
    dev = torch.device('cuda', local_rank_dev)
    dist = None
    if world > 1 or os.environ.get('VV_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('VV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; 'gloo' only for single-GPU bring-up tests
        if world == 1:                     # VV_FORCE_DIST without a launcher: a one-rank rendezvous of its own
            os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit('process group has %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus))
    rec = run_unet(args.model, args.precision, args.batch, args.steps, args.warmup, dev, rank, world, dist, args.overlap,
                   args.breakdown, args.pool, graph=not args.no_graph, measure_forward=not args.no_forward_timing,
                   ev_extra=0 if args.no_forward_timing else 4)
    # N > 1: the second record SURVEY 8(d) asks for beside the weak-scaling headline -- the reference's own split (train.py:373-375:
    # nn.DataParallel scatters ONE 256-cube batch over the GPUs => 256 / N cubes per rank and step, per-rank BatchNorm statistics)
    rec_dp = None
    dpg = args.dp_split_of
    if world > 1 and not args.no_secondary and dpg > 0 and dpg % world == 0 and dpg // world != args.batch:
        try:
            rec_dp = run_unet(args.model, args.precision, dpg // world, min(50, max(args.steps, 4)), 5, dev, rank, world, dist, args.overlap,
                              False, min(args.pool, 1024), graph=not args.no_graph, measure_forward=False, ev_extra=0)
        except Exception as e:              # must not take the headline down
            rec_dp = {'value': None, 'error': repr(e)}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    out = {'metric': 'spatio-temporal cubes/sec (train step)', 'value': rec['value'], 'unit': 'cubes/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': rec['ms_per_step'], 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': rec['dtype'], 'data': 'synthetic', 'config': rec['config'],
           'roofline': rec['roofline'], 'execution': rec['execution']}
    if 'comm' in rec:
        out['comm'] = rec['comm']
    out['config']['scaling_mode'] = 'weak: %d cubes per GPU and step, global batch %d' % (args.batch, args.batch * world)
    if rec_dp is not None:
        rec_dp['label'] = ('DataParallel-faithful strong split (train.py:373-375): ONE global batch of %d cubes over %d ranks = %d cubes per '
                           'rank and step, per-rank BatchNorm statistics; value = whole-job cubes/s' % (dpg, world, dpg // world))
        out.setdefault('configs', {})['dp_faithful_global_batch_%d' % dpg] = rec_dp
        out['config'].update({'dp_faithful_cubes_per_s': rec_dp.get('value'), 'dp_faithful_ms_per_step': rec_dp.get('ms_per_step'),
                              'dp_faithful_exposed_comm_us': (rec_dp.get('comm') or {}).get('exposed_comm_us_per_step'),
                              'weak_exposed_comm_us': (rec.get('comm') or {}).get('exposed_comm_us_per_step')})
    if world == 1 and not args.no_secondary:
        sec = {}
        for name, fn in (('full_b512_bf16', lambda: run_unet('full', 'bf16', 512, 20, 5, dev, 0, 1, None, 'none', False, 2048,
                                                             graph=not args.no_graph)),
                         # the per-rank workloads of the reference's DataParallel split (train.py:375): BASELINE configs[2]
                         # = 256 / 8 GPUs, config.cfg's batch_size = 128 / 8 GPUs -- 1-GPU proxies, hipGraph vs eager loop
                         ('net4_b32', lambda: run_unet('net4', 'fp32', 32, 50, 5, dev, 0, 1, None, 'none', False, 1024,
                                                       measure_forward=False, graph=not args.no_graph, small_batch_diag=True)),
                         ('net4_b16', lambda: run_unet('net4', 'fp32', 16, 50, 5, dev, 0, 1, None, 'none', False, 1024,
                                                       measure_forward=False, graph=not args.no_graph, small_batch_diag=True)),
                         ('flownet2_1024x448', lambda: run_flownet2(dev)),
                         ('net4_eval_scoring', lambda: run_scoring(dev))):
            try:
                sec[name] = fn()
            except Exception as e:          # a secondary record must not take the headline line down
                sec[name] = {'value': None, 'error': repr(e)}
        sec['full_b512_bf16']['baseline_config'] = 'configs[3]: ShanghaiTech 5raw+5of (context_of_num=4), batch 512, mixed bf16 -- ' \
                                                   'measured on 1 GPU (the 8-GPU run is the driver\'s)'
        for nm, per in (('net4_b32', 'BASELINE configs[2] (batch 256 over 8 GPUs)'), ('net4_b16', "config.cfg's batch_size = 128 over 8 GPUs")):
            sec[nm]['baseline_config'] = 'per-rank workload of %s, measured on 1 GPU without the gradient exchange' % per
        sec['flownet2_1024x448']['baseline_config'] = 'configs[4]: FlowNet2 correlation+conv forward on 1024x436 frame pairs, 1xMI355X'
        out.setdefault('configs', {}).update(sec)
        # scalar copies of the secondary headlines inside `config` (the driver's parsed record keeps top-level objects only)
        def _g(name, key):
            v = sec.get(name, {}).get(key)
            return None if v is None else float(v)
        out['config'].update({'cfg4_cubes_per_s': _g('full_b512_bf16', 'value'), 'cfg4_ms_per_step': _g('full_b512_bf16', 'ms_per_step'),
                              'cfg4_conv_family_us': (sec['full_b512_bf16'].get('roofline') or {}).get('avg_launch_us'),
                              'cfg4_conv_family_frac_hbm': (sec['full_b512_bf16'].get('roofline') or {}).get('frac'),
                              'net4_b32_ms': _g('net4_b32', 'ms_per_step'), 'net4_b32_cubes_per_s': _g('net4_b32', 'value'),
                              'net4_b16_ms': _g('net4_b16', 'ms_per_step'), 'net4_b16_cubes_per_s': _g('net4_b16', 'value'),
                              'flownet2_ms_per_pair': _g('flownet2_1024x448', 'ms_per_pair'),
                              'eval_scoring_cubes_per_s': _g('net4_eval_scoring', 'value')})
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
