"""Fused training / scoring driver for the UNet bank (the loop bodies of train.py:379-402 and train.py:413-427).

One process per GPU.  Data parallelism replaces the reference's single-process ``nn.DataParallel`` (train.py:375):
every rank runs the grouped HIP forward/backward on its own shard of the global batch with per-rank BatchNorm
statistics (what DataParallel does), gradients are summed over RCCL (``torch.distributed`` backend
"nccl") in three buckets -- the decoder and deep-encoder buckets are reduced on a communication stream while the rest of the
backward pass is still running -- and scaled by 1/world inside the fused Adam kernel.  No parameter broadcast per step:
parameters, BatchNorm buffers and step counters are broadcast from rank 0 ONCE when the trainer is built (DataParallel
replicates rank 0's module, train.py:375) and the ranks then apply identical updates.

Launch overhead: a train step is ~130 kernel launches through ctypes.  At the reference's per-replica batch sizes (train.py:375 splits
``batch_size`` over the GPUs: 256 / 8 = 32 cubes, config.cfg's 128 / 8 = 16) the kernels are short and the host loop would set the
pace, so a step -- cube gather, forward, backward, Adam -- is captured ONCE per (batch size, cube store) into a hipGraph and
replayed (VV_GRAPH=0 keeps the eager loop).  Everything that varies between steps lives in device memory: the cube indices in
a static buffer, Adam's step counter and bias corrections in ``bank._adam_sc`` (vv_adam_tick).  With a process group the step
is captured as segments split where the gradient buckets become final; the collectives are launched eagerly between them.
"""
import os

import torch

from . import _lib as L


def shard_batch(indices, rank, world):
    """Rank r takes a contiguous chunk of the global batch (DataParallel's scatter; train.py:375).
    The global batch must split evenly so that the mean of per-rank mean losses equals the global mean."""
    n = indices.shape[0]
    if n % world:
        raise ValueError('global batch %d does not split evenly over %d ranks' % (n, world))
    per = n // world
    return indices[rank * per:(rank + 1) * per]


class GradBuckets:
    """Sum-all-reduce of the bank's BUCKET-MAJOR gradient buffer, one in-place collective per bucket.

    ``grads`` is the flat buffer of UNetBank (bucket k = columns [bounds[k], bounds[k+1]) of every UNet, contiguous as
    [G][width_k] at float offset G*bounds[k]), so bucket k is the contiguous range ``grads[G*bounds[k] : G*bounds[k+1]]`` and
    the collective runs on it directly -- no staging copies.  ``launch(k)`` starts the collective on ``comm_stream`` (after
    everything already queued on the current stream); ``finish()`` makes the current stream wait for all of them.  Works with
    any backend (RCCL on the GPU, gloo on CPU tensors for the tests)."""

    def __init__(self, grads, G, bounds, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.grads, self.G, self.bounds, self.group = grads, int(G), list(bounds), group
        self.cuda = grads.is_cuda
        flat = grads.view(-1)
        assert flat.numel() == self.G * self.bounds[-1] and self.bounds[0] == 0
        self.views = [flat[self.G * a:self.G * b] for a, b in zip(self.bounds[:-1], self.bounds[1:])]
        self.comm_stream = torch.cuda.Stream(device=grads.device) if self.cuda else None
        self.pending = []
        self.timing = None          # set to [] to collect (bucket, start event, end event) per collective (bench diagnostics)

    def launch(self, k):
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                if self.timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                work = self.dist.all_reduce(self.views[k], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.timing is not None:
                    work.wait()                 # stream-level wait (comm stream): the end event follows the collective
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.comm_stream)
                    self.timing.append((k, e0, e1))
        else:
            work = self.dist.all_reduce(self.views[k], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((k, work))

    def finish(self):
        for k, work in self.pending:
            if self.cuda:
                with torch.cuda.stream(self.comm_stream):
                    work.wait()
            else:
                work.wait()
        if self.cuda and self.pending:
            torch.cuda.current_stream(self.grads.device).wait_stream(self.comm_stream)
        self.pending = []


class FusedTrainer:
    """forward(train) -> backward -> [bucketed RCCL all-reduce] -> Adam, all asynchronous on the current stream."""

    def __init__(self, net, lr=1e-3, eps=1e-7, betas=(0.9, 0.999), lambda_raw=1.0, lambda_of=1.0, process_group=None,
                 reset_optimizer=True, overlap=False, always_bucket=False, sync_init=True):
        net.set_loss_weights(lambda_raw, lambda_of)
        self.net = net
        self.bank = net.bank()
        if reset_optimizer:      # the reference builds a fresh Adam for every block (train.py:376)
            self.bank.adam_m = self.bank.adam_v = None
            self.bank.adam_t = 0
        self.lr, self.eps, self.betas = lr, eps, betas
        self.group = process_group
        self.world = 1
        self.buckets = None
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        if self.world > 1 and sync_init:
            self.sync_from_rank0()
        # always_bucket: run the bucketed exchange even in a one-rank group (a sum over one rank is the identity): exercises
        # init_process_group('nccl') + GradBuckets on device tensors on a single-GPU box, bit-equal to the no-group path
        if self.world > 1 or (always_bucket and process_group is not None):
            lay = self.bank.lay
            # three buckets in the order the backward pass completes them: [c8.w, U) decoder convs + transposed convs + 1x1 out
            # (45 % of the parameters), [c4.w, c8.w) the deep encoder layers (51 %), [0, c4.w) the shallow encoder layers (3 %).
            # Each all-reduce is launched right after the last kernel that writes into its columns and overlaps everything that
            # follows; only the 3 % bucket is exposed at the end of the step.
            self.buckets = GradBuckets(self.bank.grads, self.bank.G, self.bank.gb, process_group)
            self.split_label = 'wgradT_reduce0'   # last launch of the decoder half of the backward plan
            self.split_label_mid = 'wgrad_reduce4'  # last launch that writes gradients of layers 4..7
        self.event_hook = None
        self.event_labels = None
        # overlap=True ('free'): weight gradients (MFMA-bound, one workgroup per CU) run on a side stream under the
        # BatchNorm-backward passes (HBM-bound) AND the data gradients of the following layers; measured +4 % cubes/s at B=256
        # but the conv launches are then contended.  overlap='paired': each weight gradient starts when its layer's data
        # gradient is done and the next data gradient waits for it, so the only thing sharing the chip with a weight
        # gradient is the next layer's HBM-bound BatchNorm backward; every conv_mfma launch still runs alone.
        self.overlap = overlap
        self.debug_delay = None          # (stream id, cycles): see _run_dual
        self.use_graph = os.environ.get('VV_GRAPH', '1') != '0'
        self._graphs = {}                # (kind, B, cube-store pointers) -> 'warm' | _Captured
        self._graph_pool = None
        self.comm_timing = None
        # True: the train step also stores the reconstructions (bank.outputs_nchw(ws) after a step); the reference's loop only uses the
        # losses (train.py:385-399), so by default the fused step skips that store
        self.keep_outputs = False
        self.side = torch.cuda.Stream(device=self.bank.device) if overlap else None

    def sync_from_rank0(self, params=True, buffers=True):
        """Every rank takes rank 0's parameters / BatchNorm running statistics / step counters.  nn.DataParallel (train.py:375)
        replicates device 0's module, so rank 0's state is THE model: at start-up the ranks must hold the same weights (they are
        built from unseeded initialisers), and before an eval-mode scoring pass or a save they must hold rank 0's running
        statistics (per-rank statistics in between are DataParallel's own semantics)."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        bank = self.bank
        ts = ([bank.params] if params else []) + ([bank.bufs, bank.nbt] if buffers else [])
        for t in ts:
            dist.broadcast(t, src=src, group=self.group)
        bank.mark_dirty()

    # ---- plan execution with optional per-launch HIP events and a mid-plan callback
    def _run(self, plan, stream, after=None):
        hook, labels = self.event_hook, self.event_labels
        for fn, args, label in plan.calls:
            timed = hook is not None and (labels is None or label in labels)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*args, stream)
            if rc:
                L.check(rc, label)
            if timed:
                e1.record()
                hook(label, e0, e1)
            if after is not None and label in after:
                after[label]()

    def _run_dual(self, plan, after=None, lo=0, hi=None):
        """Two-stream executor of a plan: meta = (stream id, events to wait for, event to record).  lo / hi: run calls [lo, hi) only
        (a segment of a captured multi-rank step; the side stream is joined at the end of every segment, so an event recorded in an
        earlier segment has completed by stream order and is not waited for)."""
        dev = self.bank.device
        main = torch.cuda.current_stream(dev)
        streams = (main, self.side)
        self.side.wait_stream(main)
        events = {}
        hook, labels = self.event_hook, self.event_labels
        paired = self.overlap == 'paired'
        for (fn, args, label), (sid, waits, rec, pwaits) in list(zip(plan.calls, plan.meta))[lo:hi]:
            st = streams[sid]
            for w in (pwaits if paired else waits):
                if w == '*main':
                    st.wait_stream(main)
                elif w == '*side':
                    st.wait_stream(self.side)
                elif w in events or lo == 0:
                    st.wait_event(events[w])
            timed = hook is not None and (labels is None or label in labels)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
            if self.debug_delay and self.debug_delay[0] == sid:      # test knob: stall one stream before each of its launches, so
                with torch.cuda.stream(st):                            # that a missing cross-stream dependency shows as wrong numbers
                    torch.cuda._sleep(self.debug_delay[1])
            rc = fn(*args, st.cuda_stream)
            if rc:
                L.check(rc, label)
            if timed:
                e1.record(st)
                hook(label, e0, e1)
            if rec is not None:
                ev = torch.cuda.Event()
                ev.record(st)
                events[rec] = ev
            if after is not None and label in after:
                after[label](events)
        main.wait_stream(self.side)

    def _forward_train(self, ws):
        """train-mode forward + its bookkeeping (BatchNorm's num_batches_tracked, folded eval model invalidated, backward plan built):
        the part every kind of step shares"""
        bank = self.bank
        fwd = (ws.fwd if self.keep_outputs else ws.fwdq)[True]
        self._run(fwd, bank._stream())
        ws.out4_valid = bool(self.keep_outputs)
        ws.fused_fwd = bool(getattr(fwd, 'fused_outconv', False))      # which backward plan matches this forward (UNetBank.backward)
        assert ws.fused_fwd == bool(not self.keep_outputs and bank.fuse_outconv)
        ws.bwd_cur = bank.backward_plan(ws, fused=ws.fused_fwd)
        bank.bump_nbt()
        bank.mark_dirty()

    def _adam(self):
        bank = self.bank
        bank.adam_step(lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=1.0 / self.world)

    def _step(self, ws):
        bank = self.bank
        stream = bank._stream()
        self._forward_train(ws)
        if self.overlap:
            if self.buckets is None:
                self._run_dual(ws.bwd_cur)
            else:
                def dec(events):     # decoder bucket: needs the side stream's decoder weight-grads too
                    torch.cuda.current_stream(bank.device).wait_event(events['sideT0'])
                    self.buckets.launch(2)

                def mid(events):     # deep-encoder bucket: its last weight-grad reduction ran on the side stream
                    torch.cuda.current_stream(bank.device).wait_stream(self.side)
                    self.buckets.launch(1)
                self._run_dual(ws.bwd_cur, after={self.split_label: dec, self.split_label_mid: mid})
                self._finish_exchange()
        elif self.buckets is None:
            self._run(ws.bwd_cur, stream)
        else:
            self._run(ws.bwd_cur, stream, after={self.split_label: lambda: self.buckets.launch(2),
                                              self.split_label_mid: lambda: self.buckets.launch(1)})
            self._finish_exchange()
        if self.event_hook is not None and (self.event_labels is None or 'adam' in self.event_labels):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._adam()
            e1.record()
            self.event_hook('adam', e0, e1)
        else:
            self._adam()
        return ws

    def _finish_exchange(self):
        """Last (3 %) bucket + wait for all three.  With ``self.comm_timing`` set to a list, the time the main stream spends
        between the end of the backward pass and the arrival of the last sums (= the exposed communication) is recorded."""
        ct = getattr(self, 'comm_timing', None)
        if ct is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.buckets.launch(0)
        self.buckets.finish()
        if ct is not None:
            e1.record()
            ct.append((e0, e1))

    # ---- hipGraph capture of a whole step
    def _graph_ok(self):
        # (comm_timing does not prevent the replay: _finish_exchange runs eagerly BETWEEN the captured segments, so its event pair
        # is recorded outside every capture; the per-bucket timing of GradBuckets wraps the collectives themselves and needs the eager loop)
        return (self.use_graph and not self.overlap and self.event_hook is None
                and (self.buckets is None or self.buckets.timing is None) and self.bank.device.type == 'cuda')

    def _graph_dual(self, B):
        """'free' | 'paired' | None: schedule of the backward pass inside a captured one-rank train step.  Default 'free': the
        weight gradients (one workgroup per CU, MFMA-bound) are a parallel branch of the hipGraph beside the data-gradient /
        BatchNorm-backward chain -- measured 9.41 -> 9.22 ms at B = 256, 2.12 -> 1.95 ms at B = 32, 1.66 -> 1.52 ms at B = 16
        (fp32 Net4), bit-identical to the one-stream order; 'paired' measured slower than one stream.  VV_GRAPH_OVERLAP =
        0 | free | paired overrides.  With a gradient exchange (world > 1) every captured segment between two bucket launches forks
        and joins on its own."""
        v = os.environ.get('VV_GRAPH_OVERLAP', 'free')
        return v if v in ('free', 'paired') else None

    def _capture(self, segments):
        """segments: list of (list of thunks taking the raw stream handle, eager callable or None).  Returns [(graph, after)]."""
        out = []
        for thunks, after in segments:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode='thread_local'):      # RCCL's watchdog thread polls events
                st = torch.cuda.current_stream(self.bank.device).cuda_stream
                for t in thunks:
                    t(st)
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            out.append((g, after))
        return out

    @staticmethod
    def _thunk(fn, args, label):
        def run(st):
            rc = fn(*args, st)
            if rc:
                L.check(rc, label)
        return run

    def _capture_train(self, raw_u8, flow, B):
        bank, lib = self.bank, self.bank.lib
        ws = bank.workspace(B)
        bwd = bank.backward_plan(ws, fused=not self.keep_outputs and bank.fuse_outconv)
        if bank.adam_m is None:
            bank.adam_m, bank.adam_v = torch.zeros_like(bank.params), torch.zeros_like(bank.params)
        idx = torch.zeros(B, dtype=torch.long, device=bank.device)
        HWp = ws.cube.shape[1]
        keep = (raw_u8, flow)
        seg = [self._thunk(lib.vv_cube_gather, (B, bank.tot_raw, bank.tot_of, HWp, idx.data_ptr(), raw_u8.data_ptr(),
                                                flow.data_ptr() if flow is not None else None, ws.cube.data_ptr(),
                                                ws.flow.data_ptr()), 'cube_gather')]
        fwd = (ws.fwd if self.keep_outputs else ws.fwdq)[True]
        dual = self._graph_dual(B)

        def dual_range(plan, lo, hi, mode=dual):
            # two-stream execution of plan calls [lo, hi) inside the capture: launches the plan marks for the side stream (the
            # weight gradients of the backward pass, the packing of the later layers' weight panels in the forward pass) become a
            # parallel branch of the hipGraph (plan.meta: stream, events waited for / recorded), joined at the end of the range
            def run(st):
                if self.side is None:
                    self.side = torch.cuda.Stream(device=bank.device)
                keep_mode, self.overlap = self.overlap, mode
                try:
                    self._run_dual(plan, lo=lo, hi=hi)
                finally:
                    self.overlap = keep_mode
            return run
        if dual:
            seg.append(dual_range(fwd, 0, len(fwd.calls)))
        else:
            seg += [self._thunk(*c) for c in fwd.calls]

        seg.append(self._thunk(*bank.nbt_call()))      # BatchNorm's num_batches_tracked (views of bank.nbt)
        segments = []
        lo = 0
        for i, c in enumerate(bwd.calls):
            if not dual:
                seg.append(self._thunk(*c))
            if self.buckets is not None and c[2] in (self.split_label, self.split_label_mid):
                k = 2 if c[2] == self.split_label else 1
                if dual:
                    seg.append(dual_range(bwd, lo, i + 1))
                    lo = i + 1
                segments.append((seg, (lambda k=k: self.buckets.launch(k))))
                seg = []
        if dual:
            seg.append(dual_range(bwd, lo, len(bwd.calls)))
        if self.buckets is not None:
            segments.append((seg, self._finish_exchange))
            seg = []
        b1, b2 = self.betas

        def adam(st):
            L.check(lib.vv_adam_tick(bank._adam_t_dev.data_ptr(), self.lr, b1, b2, bank._adam_sc.data_ptr(), st), 'adam_tick')
            L.check(lib.vv_adam_bucketed(bank.G, bank.lay.U, len(bank.gb) - 1, bank._gb_c, bank.params.data_ptr(),
                                         bank.grads.data_ptr(), bank.adam_m.data_ptr(), bank.adam_v.data_ptr(),
                                         bank._adam_sc.data_ptr(), b1, b2, self.eps, 1.0 / self.world, st), 'adam')
        seg.append(adam)
        segments.append((seg, None))
        cap = type('Captured', (), {})()
        cap.ws, cap.idx, cap.keep = ws, idx, keep
        nseg_bwd = 1 + (2 if self.buckets is not None else 0)
        cap.launches = sum(len(t) for t, _ in segments) + (len(bwd.calls) - nseg_bwd + len(fwd.calls) - 1 if dual else 0)
        cap.schedule = dual or 'one stream'
        cap.segments = self._capture(segments)
        return cap

    def release_graphs(self, keep_store=None):
        """Drop every captured step / scoring graph (and 'warm' marker) except those of cube store ``keep_store`` = (raw pointer,
        flow pointer).  A captured graph pins the cube store it gathers from (cap.keep) and its workspace: a caller that replaces
        its store (train.py's per-segment stores of ShanghaiTech) calls this so that the old store's memory is freed with it."""
        if keep_store is None:
            self._graphs = {}
            self._store_lru = []
        else:
            self._graphs = {k: v for k, v in self._graphs.items() if k[2:4] == keep_store}
            self._store_lru = [s for s in getattr(self, '_store_lru', []) if s == keep_store]

    MAX_STORES = 2          # cube stores whose graphs are kept at the same time (a training store + a validation / scoring store)

    def _graph_lookup(self, key):
        """Cache entry of ``key`` = (kind, B, raw pointer, flow pointer, ...).  Entries are pinned per cube STORE and the
        MAX_STORES most recently used stores are kept: a key of a further store evicts every entry (captured or 'warm') of the
        least recently used one, so a trainer fed a fresh store per data segment holds two stores' graphs, not one set per
        segment it has ever seen, while a loop that alternates a training and a validation store keeps replaying both
        (train.py frees a replaced store at once with release_graphs()).  Within the cache the distinct
        (kind, B, store, hyper-parameter) keys are bounded at 8."""
        store = key[2:4]
        lru = self._store_lru = [s for s in getattr(self, '_store_lru', []) if s != store] + [store]
        cap = self._graphs.get(key)
        if cap is None:
            while len(lru) > self.MAX_STORES:
                old = lru.pop(0)
                self._graphs = {k: v for k, v in self._graphs.items() if k[2:4] != old}
            if len(self._graphs) >= 8:
                self._graphs = {k: v for k, v in self._graphs.items() if k[2:4] == store}
                if len(self._graphs) >= 8:
                    self._graphs = {}
        return cap

    def _step_graphed(self, raw_u8, flow, idx):
        bank = self.bank
        B = int(idx.numel())
        key = ('train', B, raw_u8.data_ptr(), flow.data_ptr() if flow is not None else 0, self.lr, self.eps, self.betas, self.keep_outputs)
        cap = self._graph_lookup(key)
        if cap is None or cap == 'warm':
            # the first step of a (batch size, cube store) runs the eager loop: it builds the workspace, the backward plan and the
            # Adam moments and loads every kernel; the second one captures
            ws = self._step(bank.set_input_cubes(raw_u8, flow, idx))
            if cap is None:
                self._graphs[key] = 'warm'
                return ws
            torch.cuda.current_stream(bank.device).synchronize()
            self._graphs[key] = self._capture_train(raw_u8, flow, B)
            return ws
        cap.idx.copy_(idx)
        for g, after in cap.segments:
            g.replay()
            if after is not None:
                after()
        bank._adam_t += 1            # host mirror of the device step counter (vv_adam_tick advanced it inside the graph)
        bank.mark_dirty()
        cap.ws.out4_valid = bool(self.keep_outputs)      # part of the cache key: the captured plan stores them or not
        cap.ws.fused_fwd = bool(not self.keep_outputs and bank.fuse_outconv)
        return cap.ws

    def step_cubes(self, raw_u8, flow, idx):
        """One optimisation step on cubes ``idx`` of a device-resident cube store (uint8 [N,5,32,32,3], fp32 [N,Tf,32,32,2])."""
        if self._graph_ok() and idx is not None and idx.numel() > 0:
            return self._step_graphed(raw_u8, flow, idx)
        return self._step(self.bank.set_input_cubes(raw_u8, flow, idx))

    def step_cubes_uneven(self, raw_u8, flow, idx, n_global):
        """A global batch that does not split evenly over the ranks (the last batch of an epoch; the reference keeps it,
        train.py:373, and DataParallel scatters it in chunks of ceil(n / world)): this rank holds ``idx`` (possibly empty) of
        ``n_global`` cubes.  The loss is the mean over the GLOBAL batch, so the local mean-loss gradient is weighted by
        B_local * world / n_global before the sum (Adam then divides by world); a rank without cubes contributes zeros.  One
        plain all-reduce of the whole buffer -- no bucket overlap on this rare step.  Returns the workspace, or None for an
        empty shard."""
        import torch.distributed as dist
        bank = self.bank
        b = int(idx.numel())
        ws = None
        # (the bucketed exchange of the previous step has been joined by its finish(): nothing of it is pending on the comm stream)
        assert self.buckets is None or not self.buckets.pending
        if b:
            ws = bank.set_input_cubes(raw_u8, flow, idx)
            self._forward_train(ws)
            self._run(ws.bwd_cur, bank._stream())
            bank.grads.mul_(b * self.world / float(n_global))
        else:
            bank.grads.zero_()
        if self.world > 1:
            dist.all_reduce(bank.grads, op=dist.ReduceOp.SUM, group=self.group)
        self._adam()
        return ws

    def step_nchw(self, x, x_of):
        """One optimisation step on the reference's DataLoader tensors (train.py:380-383)."""
        return self._step(self.bank.set_input_nchw(x, x_of))

    def losses(self, ws):
        return self.bank.losses(ws)

    # ---- eval-mode scoring (train.py:413-427, test.py:319-335)
    @torch.no_grad()
    def score_cubes(self, raw_u8, flow, idx=None, batch=None):
        bank = self.bank
        if self._graph_ok():
            return self._score_graphed(raw_u8, flow, idx, batch)
        ws = bank.set_input_cubes(raw_u8, flow, idx, batch)
        bank.forward(ws, False, outputs=False)
        return bank.cube_scores(ws)

    def _score_graphed(self, raw_u8, flow, idx, batch):
        """Eval-mode scoring of one batch replayed from a hipGraph: cube gather + folded-model forward + the per-cube score sums.
        The folded model is rebuilt OUTSIDE the graph when the parameters changed (bank.prepare_eval)."""
        bank, lib = self.bank, self.bank.lib
        B = int(idx.numel()) if idx is not None else (batch if batch is not None else raw_u8.shape[0])
        key = ('eval', B, raw_u8.data_ptr(), flow.data_ptr() if flow is not None else 0)
        cap = self._graph_lookup(key)
        if cap is None or cap == 'warm':
            ws = bank.set_input_cubes(raw_u8, flow, idx, batch)
            bank.forward(ws, False, outputs=False)
            out = bank.cube_scores(ws)
            if cap is None:
                self._graphs[key] = 'warm'
                return out
            torch.cuda.current_stream(bank.device).synchronize()
            sidx = torch.arange(B, dtype=torch.long, device=bank.device)
            HWp = ws.cube.shape[1]
            seg = [self._thunk(lib.vv_cube_gather, (B, bank.tot_raw, bank.tot_of, HWp, sidx.data_ptr(), raw_u8.data_ptr(),
                                                    flow.data_ptr() if flow is not None else None, ws.cube.data_ptr(),
                                                    ws.flow.data_ptr()), 'cube_gather')]
            seg += [self._thunk(*c) for c in ws.fwdq[False].calls]
            cap = type('Captured', (), {})()
            cap.ws, cap.idx, cap.keep = ws, sidx, (raw_u8, flow)
            cap.launches = len(seg)
            cap.segments = self._capture([(seg, None)])
            self._graphs[key] = cap
            return out
        if bank.eval_fold:
            bank.prepare_eval()
        if idx is None:
            if not getattr(cap, 'identity', False):
                cap.idx.copy_(torch.arange(B, dtype=torch.long, device=bank.device))
                cap.identity = True
        else:
            cap.idx.copy_(idx)
            cap.identity = False
        cap.segments[0][0].replay()
        cap.ws.out4_valid = False
        return bank.cube_scores(cap.ws)

    @torch.no_grad()
    def score_nchw(self, x, x_of):
        bank = self.bank
        ws = bank.set_input_nchw(x, x_of)
        bank.forward(ws, False, outputs=False)
        return bank.cube_scores(ws)
