"""Data-parallel gradient exchange for the G+D step: one process per GPU, RCCL over xGMI.

Reference behaviour (run_network.py:66, ``accelerator="ddp"``): torch DDP averages the gradients of
the network being optimised across ranks inside ``loss.backward()``; BatchNorm statistics stay
per-rank (no sync_batchnorm); rank 0's buffers are what a checkpoint / evaluation sees.

MI355X-first shape of the same exchange:
  * each network's gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), laid out in the order
    backward produces them (last layer first) and cut into a few large contiguous buckets (<= 3 for the DCGAN
    generator: 10.5 / 33.5 / 6.5 MB) -- large messages are what the point-to-point xGMI links want;
  * a bucket's all-reduce is issued from the autograd thread the moment its last gradient has been accumulated
    (post-accumulate-grad hooks), so it runs on RCCL's stream underneath the rest of that network's backward: for
    the generator only the last bucket (block1, 6.5 MB) is still in flight when backward ends;
  * nothing waits for the exchange until the network is next USED (a forward-pre hook on the module): the
    discriminator's exchange + optimizer step therefore also overlap the generator forward that opens the next
    training_step, which does not read the discriminator;
  * the optimizer step itself is deferred to that same point and takes the 1/world factor as an argument, so the
    result is bit-identical to the "all-reduce(mean), then step" order of DDP.

Works on any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist

BUCKET_BYTES = int(os.environ.get("GZ_DDP_BUCKET_MB", "16")) << 20
# CUs the convolution planner leaves to RCCL's channel kernels while the exchange overlaps backward (gz_set_cu_budget).
# Default 0 = plans sized for the whole chip: the single-GPU rehearsal (profiles/r04_contention.json, tools/
# contention_rehearsal.py) shows that a co-running channel kernel costs the step 7-11 % WHILE IT RUNS whatever the
# planner assumes -- budgeted plans were within +-1 % of the unbudgeted ones at R = 8..32 and worse at R = 64 -- so the
# loss is a straggler effect of the CUs that share their vector-memory path with a channel, not a slot-count effect.
# The exposure is the exchange window itself (~0.5 ms of a 6.5 ms pair at bs 128).  The knob stays for real 8-GPU runs.
CU_RESERVE = int(os.environ.get("GZ_DDP_CU_RESERVE", "0"))


class _FlatGrads:
    def __init__(self, params, bucket_bytes=BUCKET_BYTES):
        self.params = [p for p in params][::-1]        # backward order: the last layer's gradient lands first
        # every view starts on a 16-byte boundary (the sink kernels read and write float4; a 1-element bias -- HoloGAN's
        # logit head -- would otherwise misalign everything behind it); the <= 3 padding floats per parameter stay zero
        self.offsets = []
        off = 0
        for p in self.params:
            off = (off + 3) & ~3
            self.offsets.append(off)
            off += p.numel()
        ref = self.params[0]
        self.flat = torch.zeros(off, device=ref.device, dtype=ref.dtype)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        # contiguous buckets: close one when the next parameter would push it over the cap
        self.buckets = []            # [start, end, first param index, last param index + 1]
        start = first = 0
        for i, p in enumerate(self.params):
            end = self.offsets[i] + p.numel()
            nxt = self.params[i + 1].numel() if i + 1 < len(self.params) else 0
            if i + 1 == len(self.params) or (end - start + nxt) * 4 > bucket_bytes:
                self.buckets.append((start, end, first, i + 1))
                start, first = end, i + 1
        self.bucket_of = {}
        for b, (_, _, lo, hi) in enumerate(self.buckets):
            for i in range(lo, hi):
                self.bucket_of[id(self.params[i])] = b

    def rebind(self):
        """Re-attach the views if something replaced / dropped ``p.grad``."""
        for p, off in zip(self.params, self.offsets):
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view


class GradSync:
    """Plugs into harness.Trainer: ``before_step``, ``after_backward``, ``flush``."""

    def __init__(self, module, process_group=None, overlap=True, bucket_bytes=BUCKET_BYTES, broadcast_buffers=False):
        self.module = module
        # torch DDP (the reference's accelerator="ddp", run_network.py:66) broadcasts rank 0's buffers -- the BatchNorm
        # running statistics, num_batches_tracked, spectral norm's u / v -- to every rank at the top of EVERY forward
        # (broadcast_buffers=True is its default).  Training-mode BatchNorm never reads the running statistics, so for
        # the standard networks the losses and parameters do not depend on it; what does is (a) in-training evaluation
        # on ranks > 0 and (b) HoloGAN's spectral-norm power iteration, which continues from u / v.  Default here:
        # per-rank buffers during training, rank 0's at every checkpoint (sync_buffers) -- one collective per
        # checkpoint instead of one per step; ``broadcast_buffers=True`` reproduces DDP's per-step broadcast (one
        # coalesced broadcast per dtype at the top of each training_step).  tests/test_ddp_gloo.py states both.
        self.broadcast_buffers = broadcast_buffers
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        # test hook: issue the collective even on a single rank (exercises the RCCL call path)
        self.always_reduce = bool(os.environ.get("GZ_DDP_ALWAYS_REDUCE")) and dist.is_initialized()
        self.nets = [module.discriminator, module.generator]     # optimizer_idx order
        self.flats = [_FlatGrads(list(n.parameters()), bucket_bytes) for n in self.nets]
        self.pending = [None, None]   # (works, optimizer)
        self.active = None            # network whose backward is running
        self.remaining = [None, None]   # per bucket: gradients still to arrive in this backward
        self.works = [[], []]
        self.issued = [set(), set()]
        self.stats = {"buckets_from_hooks": 0, "buckets_after_backward": 0}
        # GZ_DDP_MEASURE=1 (bench.py --gpus N sets it): bracket every wait for a bucket with events on the compute
        # stream (host clock on CPU tensors), so that a multi-GPU run reports how much of the exchange was EXPOSED --
        # the time the compute stream sat behind a collective that had not finished
        self.measure = bool(os.environ.get("GZ_DDP_MEASURE"))
        self._waits = []              # (optimizer_idx, start event, end event) or (optimizer_idx, seconds)
        # the tile / split plans are sized to whole rounds of workgroup slots: with the exchange's channel kernels on the
        # chip, tell the planner how many CUs it can count on (include/gz_ops.h: gz_set_cu_budget)
        self.cu_budget = 256
        if self._reduces() and self.flats[0].flat.is_cuda and self.overlap and CU_RESERVE > 0:
            from ._lib import lib
            self.cu_budget = lib.gz_set_cu_budget(256 - CU_RESERVE)
        self.hooks = [n.register_forward_pre_hook(self._make_hook(i)) for i, n in enumerate(self.nets)]
        self.net_of = {}              # id(param) -> optimizer_idx
        self.reported = set()         # parameters whose gradient of the running backward pass is complete
        for idx, fg in enumerate(self.flats):
            for p in fg.params:
                self.net_of[id(p)] = idx
                self.hooks.append(p.register_post_accumulate_grad_hook(self._make_grad_hook(idx)))

    def _make_hook(self, idx):
        def hook(_module, _inputs):
            self.finalize(idx)
        return hook

    def _make_grad_hook(self, idx):
        def hook(p):
            self._param_ready(idx, p)
        return hook

    def sink_listener(self, p):
        """functional's gradient sinks (harness.Trainer turns them on): every weight-gradient launch of this backward
        pass that feeds ``p`` has been queued -- the parameter counts as delivered; its slabs are summed into the flat
        buffer (one gz_reduce_multi launch per bucket) right before the bucket's all-reduce is issued."""
        idx = self.net_of.get(id(p))
        if idx is not None:
            self._param_ready(idx, p)

    def _param_ready(self, idx, p):
        if self.active != idx or not self.overlap or id(p) in self.reported:
            return
        self.reported.add(id(p))
        fg = self.flats[idx]
        b = fg.bucket_of[id(p)]
        self.remaining[idx][b] -= 1
        if self.remaining[idx][b] == 0:
            self._flush_sinks(fg.params[fg.buckets[b][2]:fg.buckets[b][3]])
            self._issue(idx, b)
            self.stats["buckets_from_hooks"] += 1

    @staticmethod
    def _flush_sinks(params=None):
        from . import functional as F
        if params is not None and not params[0].is_cuda:
            return
        F.flush_grad_sinks(params)

    def _reduces(self):
        return self.world > 1 or self.always_reduce

    def _issue(self, idx, b):
        if b in self.issued[idx]:
            return
        self.issued[idx].add(b)
        if not self._reduces():
            return
        fg = self.flats[idx]
        start, end = fg.buckets[b][:2]
        self.works[idx].append(dist.all_reduce(fg.flat[start:end], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))

    @torch.no_grad()
    def _broadcast_all_buffers(self, src=0):
        """DDP's ``_sync_buffers`` at the top of a forward: every buffer of the module takes rank ``src``'s value."""
        by_dtype = {}
        for b in self.module.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n

    def before_step(self, optimizer_idx):
        if self.broadcast_buffers and self.world > 1:
            self._broadcast_all_buffers()
        if getattr(self.module, "mutates_discriminator_before_forward", False):
            self.finalize(0)     # WGAN clamps D's weights at the top of training_step
        self.finalize(optimizer_idx)     # a pending step of the SAME network must land before its next backward
        fg = self.flats[optimizer_idx]
        fg.rebind()
        self.active = optimizer_idx
        self.remaining[optimizer_idx] = [sum(1 for i in range(lo, hi) if fg.params[i].requires_grad)
                                         for (_, _, lo, hi) in fg.buckets]
        self.works[optimizer_idx] = []
        self.issued[optimizer_idx] = set()
        self.reported = set()

    def after_backward(self, optimizer_idx, optimizer):
        fg = self.flats[optimizer_idx]
        if fg.flat.is_cuda:
            self._flush_sinks()          # whatever no complete bucket has claimed yet
        fg.rebind()
        self.active = None
        for b in range(len(fg.buckets)):       # whatever the hooks did not cover (unused parameters, overlap off)
            if b not in self.issued[optimizer_idx]:
                self._issue(optimizer_idx, b)
                self.stats["buckets_after_backward"] += 1
        self.pending[optimizer_idx] = (self.works[optimizer_idx], optimizer)
        self.works[optimizer_idx] = []
        if not self.overlap:
            self.finalize(optimizer_idx)

    def finalize(self, idx):
        item = self.pending[idx]
        if item is None:
            return
        self.pending[idx] = None
        works, optimizer = item
        fg = self.flats[idx]
        scale = 1.0
        if works and self.measure:
            self._timed_wait(idx, works, fg.flat)
        else:
            for work in works:
                work.wait()
        if works:
            scale = 1.0 / self.world
        if scale != 1.0 and getattr(optimizer, "accepts_grad_scale", False):
            optimizer.step(grad_scale=scale)      # the fused optimizers fold the 1/world into their single pass
        else:
            if scale != 1.0:
                fg.flat.mul_(scale)
            optimizer.step()
        fg.flat.zero_()          # == optimizer.zero_grad(set_to_none=False), one memset

    def _timed_wait(self, idx, works, flat):
        if flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for work in works:
                work.wait()              # NCCL / RCCL: makes the current stream wait, the host does not block
            e1.record()
            self._waits.append((idx, e0, e1))
        else:
            import time
            t0 = time.perf_counter()
            for work in works:
                work.wait()
            self._waits.append((idx, time.perf_counter() - t0))

    def exposed_wait_ms(self):
        """{'discriminator': ms, 'generator': ms, 'waits': n}: total time the compute stream (or the host, for CPU
        tensors) spent waiting for gradient buckets since the last call.  Synchronises the device."""
        out = {"discriminator": 0.0, "generator": 0.0, "waits": len(self._waits)}
        if any(len(w) == 3 for w in self._waits):
            torch.cuda.synchronize()
        for w in self._waits:
            ms = w[1].elapsed_time(w[2]) if len(w) == 3 else w[1] * 1e3
            out["discriminator" if w[0] == 0 else "generator"] += ms
        self._waits = []
        return out

    def flush(self):
        for i in range(len(self.nets)):
            self.finalize(i)

    @torch.no_grad()
    def sync_buffers(self, src=0):
        """Broadcast rank ``src``'s norm buffers (what DDP's broadcast_buffers leaves on every rank).  A collective:
        EVERY rank must call it (before evaluation or checkpointing)."""
        if self.world <= 1:
            return
        for net in self.nets:
            for b in net.buffers():
                dist.broadcast(b, src=src, group=self.group)

    def close(self):
        self.flush()
        for h in self.hooks:
            h.remove()
        if self.cu_budget != 256:
            from ._lib import lib
            lib.gz_set_cu_budget(256)
            self.cu_budget = 256
