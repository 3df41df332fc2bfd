"""Data-parallel gradient exchange for the G+D step: one process per GPU, RCCL over xGMI.

Reference behaviour (run_network.py:66, ``accelerator="ddp"``): torch DDP averages the gradients of
the network being optimised across ranks inside ``loss.backward()``; BatchNorm statistics stay
per-rank (no sync_batchnorm); rank 0's buffers are what a checkpoint / evaluation sees.

MI355X-first shape of the same exchange:
  * each network's gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), laid out in the order
    backward produces them (last layer first) and cut into a few large contiguous buckets -- large messages are what
    the point-to-point xGMI links want;
  * a bucket's all-reduce is issued from the autograd thread the moment its last gradient has been accumulated
    (post-accumulate-grad hooks: autograd runs a parameter's AccumulateGrad node once per pass, after EVERY node that
    feeds it -- its own dependency count, valid for double-backward graphs too), so it runs on RCCL's stream underneath
    the rest of that network's backward;
  * nothing waits for the exchange until a LAYER that reads the bucket's parameters runs again (round 5: the modules
    announce the parameters a layer is about to read, ``functional.ready``): buckets are waited for, stepped (one fused
    optimizer launch per bucket, 1/world and the re-zeroing of the flat buffer folded in) and re-packed one by one, in
    the order they were issued.  Networks whose modules do not announce their parameters (``gates_parameters``) are
    finalized as a whole at the top of their forward, as before;
  * the DEFERRED TAIL (round 5): the weight-gradient launches of the layers the next forward needs LAST (the DCGAN
    generator's block3 / block4 / output layer: 10.5 of 50.6 MB) are postponed to the end of backward and exchanged as
    the last bucket.  Backward order would otherwise make the FIRST layer's gradient (block1) the last message, with
    nothing left to hide it behind -- the next step opens with that very layer, since the stacked discriminator pass
    needs G(z) first.  Now block2's and block1's buckets travel underneath the postponed launches (0.5 ms of MFMA work
    at bs 128), and the tail bucket itself underneath block1's and block2's next forward.  Same kernels on the same
    operands in another order: bit-identical results;
  * the optimizer step takes the 1/world factor as an argument, so the result is bit-identical to the
    "all-reduce(mean), then step" order of DDP.

``exchange=False`` passes (gradient accumulation, harness.Trainer(accumulate_grad_batches=k)) add into the flat buffer
and exchange nothing -- Lightning's ``block_ddp_sync_behaviour``; the all-reduce runs on the stepping batch only.

Works on any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist

# Bucket cap.  Every message costs the compute stream two cross-stream hand-overs of ~20 us each (issue + wait) whatever its
# size, and every exchange of this workload hides behind milliseconds of compute, so messages are few and large: 32 MB
# (round 6; 16 MB before) makes HoloGAN's critic one message (21.5 MB at 64x64, 34 MB at 128x128).
BUCKET_BYTES = int(os.environ.get("GZ_DDP_BUCKET_MB", "32")) << 20
MIN_BUCKET_BYTES = 1 << 20       # a bucket is not closed below this size just because a large parameter follows
# CUs the convolution planner leaves to RCCL's channel kernels while the exchange overlaps backward (gz_set_cu_budget).
# Default 0 = plans sized for the whole chip: the single-GPU rehearsal (profiles/r04_contention.json, tools/
# contention_rehearsal.py) shows that a co-running channel kernel costs the step 7-11 % WHILE IT RUNS whatever the
# planner assumes -- budgeted plans were within +-1 % of the unbudgeted ones at R = 8..32 and worse at R = 64 -- so the
# loss is a straggler effect of the CUs that share their vector-memory path with a channel, not a slot-count effect.
# The exposure is the exchange window itself (~0.5 ms of a 6.5 ms pair at bs 128).  The knob stays for real 8-GPU runs.
CU_RESERVE = int(os.environ.get("GZ_DDP_CU_RESERVE", "0"))
DEFER_TAIL = os.environ.get("GZ_DDP_DEFER_TAIL", "1") != "0"


class _FlatGrads:
    def __init__(self, params, bucket_bytes=BUCKET_BYTES, tail=(), arrival=None):
        """``params``: the network's parameters in definition order; ``arrival``: the same parameters in the order their
        gradients COMPLETE during backward when that is not simply the reverse (a network says so with
        ``grad_arrival_order()``: HoloGAN's five ZMapping layers are evaluated by one launch at the top of the forward,
        so their gradients land last whichever block they belong to)."""
        tail_ids = {id(p) for p in tail}
        back = [p for p in params][::-1] if arrival is None else list(arrival)
        if arrival is not None and sorted(map(id, back)) != sorted(id(p) for p in params):
            raise ValueError("grad_arrival_order() must be a permutation of the network's parameters")
        self.params = [p for p in back if id(p) not in tail_ids] + [p for p in back if id(p) in tail_ids]
        self.n_main = sum(1 for p in back if id(p) not in tail_ids)      # params[n_main:] = the deferred tail
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
        # the views are made ONCE: re-creating them per step (two tensor ops per parameter, twice per step) was 0.2 ms
        # of host time per DCGAN pair
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        for p, v in zip(self.params, self.views):
            p.grad = v
        # contiguous buckets: close one when the next parameter would push it over the cap -- unless it is still tiny (a
        # few KB of norm parameters in front of a 33 MB weight ride with it) or what is left of its section is small (the
        # DCGAN generator's block1, 6.5 MB, comes right behind block2's 33.5 MB and only ~30 us of backward separate
        # them: one message instead of two; every message costs two cross-stream hand-overs of ~20 us each) -- and
        # where the deferred tail begins
        self.buckets = []            # [start, end, first param index, last param index + 1]
        start = first = 0
        floor = min(MIN_BUCKET_BYTES, bucket_bytes)
        sizes = [p.numel() * 4 for p in self.params]
        for i, p in enumerate(self.params):
            end = self.offsets[i] + p.numel()
            last = i + 1 == len(self.params)
            nxt = self.params[i + 1].numel() if not last else 0
            in_tail = i >= self.n_main
            section_end = len(self.params) if in_tail else self.n_main
            rest = sum(sizes[i + 1:section_end])
            # the deferred tail is waited for at the gates of ITS layers only, which all sit late in the next forward:
            # cutting it lands nothing earlier and costs a second pair of cross-stream hand-overs -- one message up to
            # twice the cap (HoloGAN's block3 + block4: 17.8 MB)
            cap = 2 * bucket_bytes if in_tail else bucket_bytes
            over = (end - start + nxt) * 4 > cap and (end - start) * 4 >= floor and rest > bucket_bytes // 2
            if last or over or i + 1 == self.n_main:
                self.buckets.append((start, end, first, i + 1))
                start, first = end, i + 1
        self.bucket_of = {}
        for b, (_, _, lo, hi) in enumerate(self.buckets):
            for i in range(lo, hi):
                self.bucket_of[id(self.params[i])] = b
        self.tail_buckets = {self.bucket_of[id(p)] for p in self.params[self.n_main:]}

    def bucket_params(self, b):
        return self.params[self.buckets[b][2]:self.buckets[b][3]]

    def rebind(self):
        """Re-attach the views if something replaced / dropped ``p.grad``."""
        for p, view in zip(self.params, self.views):
            g = p.grad
            if g is view:
                continue
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
            p.grad = view


class _Pending:
    """The exchange of one backward pass: buckets in the order their all-reduce was issued, the optimizer that will
    consume them, how many have been waited for + stepped so far."""
    __slots__ = ("order", "pos", "works", "optimizer", "done")

    def __init__(self, order, works, optimizer):
        self.order, self.works, self.optimizer, self.done = list(order), dict(works), optimizer, 0
        self.pos = {b: i for i, b in enumerate(self.order)}


def pick_tail(net, bucket_bytes=BUCKET_BYTES, min_bytes=MIN_BUCKET_BYTES):
    """The deferred tail of a network: the convolution weights of its LAST layers (definition order = forward order for
    the standard networks) while they fit min(bucket cap, a quarter of the network's gradient bytes); at least 1 MB,
    else no tail (a tiny last message hides nothing and costs a collective).  A network that knows better declares
    ``deferred_tail_parameters()`` (HoloGAN's generator: its schedule runs two generator steps back to back, so the
    tail is sized to what the NEXT generator forward can hide)."""
    own = getattr(net, "deferred_tail_parameters", None)
    if own is not None:
        tail = [p for p in own() if p.dim() >= 4 and not (p.numel() & 3)]
        return tail if sum(p.numel() for p in tail) * 4 >= min_bytes else []
    params = list(net.parameters())
    total = sum(p.numel() for p in params) * 4
    budget = min(bucket_bytes, total // 4)
    tail, used = [], 0
    for p in reversed(params):
        if p.dim() != 4 or (p.numel() & 3):
            continue                     # norm gains / biases: written in place by their backward kernel, never deferred
        if used + p.numel() * 4 > budget:
            break
        tail.append(p)
        used += p.numel() * 4
    return tail if used >= min_bytes else []


class GradSync:
    """Plugs into harness.Trainer: ``before_step``, ``after_backward``, ``flush``."""

    def __init__(self, module, process_group=None, overlap=True, bucket_bytes=BUCKET_BYTES, broadcast_buffers=None,
                 defer_tail=None, tail_min_bytes=MIN_BUCKET_BYTES):
        self.module = module
        # torch DDP (the reference's accelerator="ddp", run_network.py:66) broadcasts rank 0's buffers -- the BatchNorm
        # running statistics, num_batches_tracked, spectral norm's u / v -- to every rank at the top of EVERY forward
        # (broadcast_buffers=True is its default).  Training-mode BatchNorm never reads the running statistics, so for
        # the standard networks the losses and parameters do not depend on it; what does is (a) in-training evaluation
        # on ranks > 0 and (b) HoloGAN's spectral-norm power iteration, which continues from u / v.  Default here:
        # per-rank buffers during training, rank 0's at every checkpoint (sync_buffers) -- one collective per
        # checkpoint instead of one per step; ``broadcast_buffers=True`` reproduces DDP's per-step broadcast (one
        # coalesced broadcast per dtype at the top of each training_step).  tests/test_ddp_gloo.py states both.
        # Round 6, the default ``None`` = "what training READS": the buffers that feed the training arithmetic -- the
        # spectral-norm power-iteration vectors ``weight_u`` / ``weight_v`` (HoloGAN's critic, a few KB) -- take rank
        # 0's value at the top of every training_step exactly as under DDP, so that every rank normalises with the same
        # sigma the reference's ranks see; the BatchNorm running statistics, which training-mode BatchNorm never
        # reads, stay per rank until a checkpoint / evaluation (sync_buffers).
        self.broadcast_buffers = broadcast_buffers
        self._trained_buffers = [b for n, b in module.named_buffers() if n.endswith(("weight_u", "weight_v"))]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        # test hook: issue the collective even on a single rank (exercises the RCCL call path)
        self.always_reduce = bool(os.environ.get("GZ_DDP_ALWAYS_REDUCE")) and dist.is_initialized()
        self.nets = [module.discriminator, module.generator]     # optimizer_idx order
        # per-layer gates: only for networks whose modules announce their parameters (functional.ready)
        self.lazy = [bool(overlap and getattr(n, "gates_parameters", False)) for n in self.nets]
        on_gpu = all(p.is_cuda for n in self.nets for p in n.parameters())
        # the deferred tail postpones kernel launches of the HIP path (functional's gradient sinks), so by default it
        # exists for GPU networks only; defer_tail=True on CPU tensors (tests) still ORDERS the tail's bucket last --
        # its parameters are reported after backward instead of from their hooks -- which is all the landing logic sees
        if defer_tail is None:
            defer_tail = DEFER_TAIL and on_gpu
        tails = [pick_tail(n, bucket_bytes, tail_min_bytes) if (defer_tail and lazy) else []
                 for n, lazy in zip(self.nets, self.lazy)]
        arrivals = [getattr(n, "grad_arrival_order", None) for n in self.nets]
        self.flats = [_FlatGrads(list(n.parameters()), bucket_bytes, t, None if a is None else a())
                      for n, t, a in zip(self.nets, tails, arrivals)]
        self.tails = tails
        self.pending = [None, None]   # _Pending per network
        self.active = None            # network whose backward is running (and exchanging)
        self.remaining = [None, None]   # per bucket: gradients still to arrive in this backward
        self.works = [{}, {}]
        self.order = [[], []]         # buckets in issue order during the running pass
        self.stats = {"buckets_from_hooks": 0, "buckets_after_backward": 0, "buckets_deferred_tail": 0}
        # GZ_DDP_MEASURE=1 (bench.py sets it): bracket every wait for a bucket with events on the compute
        # stream (host clock on CPU tensors), so that a run reports how much of the exchange was EXPOSED --
        # the time the compute stream sat behind a collective that had not finished
        self.measure = bool(os.environ.get("GZ_DDP_MEASURE"))
        self._waits = []              # (optimizer_idx, start event, end event) or (optimizer_idx, seconds)
        self.trace = None             # tests: a list that receives ("issue" | "wait" | "step" | "gate", idx, bucket)
        # the tile / split plans are sized to whole rounds of workgroup slots: with the exchange's channel kernels on the
        # chip, tell the planner how many CUs it can count on (include/gz_ops.h: gz_set_cu_budget)
        self.cu_budget = 256
        if self._reduces() and self.flats[0].flat.is_cuda and self.overlap and CU_RESERVE > 0:
            from ._lib import lib
            self.cu_budget = lib.gz_set_cu_budget(256 - CU_RESERVE)
        self.hooks = []
        for i, n in enumerate(self.nets):
            if self.lazy[i]:
                self.hooks.append(n.register_forward_hook(self._make_post_hook(i)))
                # plain torch modules (the CPU oracle in the gloo tests) call their parameter-owning children: gate there
                for sub in n.modules():
                    own = list(sub.parameters(recurse=False))
                    if own:
                        self.hooks.append(sub.register_forward_pre_hook(self._make_gate_hook(own)))
            else:
                self.hooks.append(n.register_forward_pre_hook(self._make_hook(i)))
        self.net_of = {}              # id(param) -> optimizer_idx
        self.tail_ids = set()
        self.reported = set()         # parameters whose gradient of the running backward pass is complete
        for idx, fg in enumerate(self.flats):
            for p in fg.params:
                self.net_of[id(p)] = idx
                self.hooks.append(p.register_post_accumulate_grad_hook(self._make_grad_hook(idx)))
            self.tail_ids.update(id(p) for p in fg.params[fg.n_main:])
        self._F = None
        if on_gpu:
            from . import functional as F
            self._F = F
        self._prev_gate = None
        if any(self.lazy):
            from . import functional as Fn
            self._gate_owner = Fn
            self._prev_gate = Fn.set_param_gate(self.gate)
        else:
            self._gate_owner = None

    # ---- hooks ---------------------------------------------------------------------------------------------------
    def _make_hook(self, idx):
        def hook(_module, _inputs):
            self.finalize(idx)
        return hook

    def _make_gate_hook(self, params):
        def hook(_module, _inputs):
            self.gate(params)
        return hook

    def _make_post_hook(self, idx):
        def hook(_module, _inputs, _output):
            if self.pending[idx] is not None:
                raise RuntimeError("ddp.GradSync: %s declares gates_parameters but its forward left gradient buckets "
                                   "un-finalized -- a layer read its parameters without functional.ready()"
                                   % type(self.nets[idx]).__name__)
        return hook

    def _make_grad_hook(self, idx):
        def hook(p):
            self._param_ready(idx, p)
        return hook

    def gate(self, params):
        """A layer is about to read ``params``: the exchange + optimizer step of their buckets must have landed."""
        for p in params:
            idx = self.net_of.get(id(p))
            if idx is None:
                continue
            item = self.pending[idx]
            if item is None:
                continue
            b = self.flats[idx].bucket_of[id(p)]
            if item.pos[b] >= item.done:
                if self.trace is not None:
                    self.trace.append(("gate", idx, b))
                self.finalize(idx, upto=b)

    # ---- the running pass ----------------------------------------------------------------------------------------
    def _param_ready(self, idx, p, deferred=False):
        if self.active != idx or not self.overlap or id(p) in self.reported:
            return
        if id(p) in self.tail_ids and not deferred:
            return                    # its launch has been postponed: reported by after_backward
        self.reported.add(id(p))
        fg = self.flats[idx]
        b = fg.bucket_of[id(p)]
        self.remaining[idx][b] -= 1
        if self.remaining[idx][b] == 0:
            self._flush_sinks(fg.bucket_params(b))
            self._issue(idx, b)
            self.stats["buckets_deferred_tail" if deferred else "buckets_from_hooks"] += 1

    def _flush_sinks(self, params=None):
        if self._F is not None:
            self._F.flush_grad_sinks(params)

    def _reduces(self):
        return self.world > 1 or self.always_reduce

    def _issue(self, idx, b):
        if b in self.works[idx]:
            return
        work = None
        if self._reduces():
            fg = self.flats[idx]
            start, end = fg.buckets[b][:2]
            work = dist.all_reduce(fg.flat[start:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works[idx][b] = work
        self.order[idx].append(b)
        if self.trace is not None:
            self.trace.append(("issue", idx, b))

    @torch.no_grad()
    def _broadcast_all_buffers(self, src=0, only=None):
        """DDP's ``_sync_buffers`` at the top of a forward: every buffer of the module (or the buffers ``only``) takes
        rank ``src``'s value."""
        by_dtype = {}
        for b in (self.module.buffers() if only is None else only):
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n

    def before_step(self, optimizer_idx, exchange=True):
        if self.world > 1:
            if self.broadcast_buffers:
                self._broadcast_all_buffers()
            elif self.broadcast_buffers is None and self._trained_buffers:
                self._broadcast_all_buffers(only=self._trained_buffers)
        if getattr(self.module, "mutates_discriminator_before_forward", False):
            self.finalize(0)     # WGAN clamps D's weights at the top of training_step
        # A pending step of the SAME network must land before its next backward.  A network with per-layer gates lands
        # it by itself, bucket by bucket, during the forward that precedes that backward (a forward that leaves a
        # bucket un-finalized raises, _make_post_hook) -- HoloGAN's D, G, G schedule hands a generator pass straight to
        # the next generator step, whose early layers then run underneath the previous pass's tail bucket.
        if not self.lazy[optimizer_idx]:
            self.finalize(optimizer_idx)
        fg = self.flats[optimizer_idx]
        fg.rebind()
        self.active = optimizer_idx if exchange else None
        self.remaining[optimizer_idx] = [sum(1 for i in range(lo, hi) if fg.params[i].requires_grad)
                                         for (_, _, lo, hi) in fg.buckets]
        self.works[optimizer_idx] = {}
        self.order[optimizer_idx] = []
        self.reported = set()
        if self._F is not None:
            self._F.set_deferred_wgrads(self.tails[optimizer_idx] if (exchange and self.overlap) else ())

    def after_backward(self, optimizer_idx, optimizer, exchange=True):
        """Called by harness.Trainer right after ``loss.backward()``, while the gradient sinks are still on."""
        fg = self.flats[optimizer_idx]
        if not exchange:                 # an accumulating pass: the gradients stay in the flat buffer, nothing travels
            self._flush_sinks()
            fg.rebind()
            self.active = None
            return
        # (1) whatever no hook has claimed among the main buckets (unused parameters, overlap off) goes out first ...
        main = [b for b in range(len(fg.buckets)) if b not in fg.tail_buckets]
        tail = sorted(fg.tail_buckets)
        for b in main:
            if b not in self.works[optimizer_idx]:
                self._flush_sinks(fg.bucket_params(b))
                self._issue(optimizer_idx, b)
                self.stats["buckets_after_backward"] += 1
        # (2) ... then the postponed weight-gradient launches run, UNDER the all-reduces issued so far ...
        if self._F is not None:
            n = self._F.run_deferred_wgrads()
            self._F.set_deferred_wgrads(())
            if self.trace is not None and tail:
                self.trace.append(("deferred", optimizer_idx, n))
        # (3) ... and their bucket is the last message of the pass
        if tail and self.overlap:
            for p in fg.params[fg.n_main:]:
                if p.requires_grad:
                    self._param_ready(optimizer_idx, p, deferred=True)
        self._flush_sinks()              # (anything left: parameters outside every bucket rule)
        for b in tail:
            if b not in self.works[optimizer_idx]:
                self._issue(optimizer_idx, b)
                self.stats["buckets_after_backward"] += 1
        fg.rebind()
        self.active = None
        self.pending[optimizer_idx] = _Pending(self.order[optimizer_idx], self.works[optimizer_idx], optimizer)
        self.works[optimizer_idx] = {}
        self.order[optimizer_idx] = []
        if not self.overlap:
            self.finalize(optimizer_idx)

    def abort_pass(self, optimizer_idx):
        """A training_step / backward of network ``optimizer_idx`` raised half-way (harness.Trainer's except branch):
        wait for the all-reduces this rank has already issued from the backward hooks (a collective that is never waited
        for keeps its buffer busy and leaves the ranks with different numbers of matched calls), drop them, clear the
        running pass's bookkeeping and the postponed launches, and zero what the pass had accumulated in the flat
        buffer.  Earlier passes that are still pending (``self.pending``) are untouched."""
        for work in self.works[optimizer_idx].values():
            if work is not None:
                work.wait()
        self.works[optimizer_idx] = {}
        self.order[optimizer_idx] = []
        self.reported = set()
        self.active = None
        if self._F is not None:
            self._F.discard_grad_sinks()
            self._F.set_deferred_wgrads(())
        if self.pending[optimizer_idx] is None:       # (else the flat buffer still holds the pending pass's gradients)
            self.flats[optimizer_idx].flat.zero_()

    # ---- landing a pass ------------------------------------------------------------------------------------------
    def finalize(self, idx, upto=None):
        """Wait for + step the pending buckets of network ``idx`` in issue order: all of them, or up to bucket ``upto``."""
        item = self.pending[idx]
        if item is None:
            return
        fg = self.flats[idx]
        stop = len(item.order) if upto is None else item.pos[upto] + 1
        whole = item.done == 0 and stop == len(item.order)
        reduced = False
        first = item.done
        while item.done < stop:
            b = item.order[item.done]
            work = item.works.get(b)
            if work is not None:
                reduced = True
                if self.trace is not None:
                    self.trace.append(("wait", idx, b))
                if self.measure:
                    self._timed_wait(idx, work, fg.flat)
                else:
                    work.wait()
            item.done += 1
            if not whole:
                self._step(idx, item.optimizer, fg, [b], work is not None)
        if whole:                     # nothing was stepped bucket-wise: one optimizer launch for the network
            self._step(idx, item.optimizer, fg, item.order[first:stop], reduced)
        if item.done == len(item.order):
            self.pending[idx] = None

    def _step(self, idx, optimizer, fg, buckets, reduced):
        scale = 1.0 / self.world if reduced else 1.0
        if self.trace is not None:
            for b in buckets:
                self.trace.append(("step", idx, b))
        everything = len(buckets) == len(fg.buckets)
        lo = min(fg.buckets[b][0] for b in buckets)
        hi = max(fg.buckets[b][1] for b in buckets)
        contiguous = sum(fg.buckets[b][1] - fg.buckets[b][0] for b in buckets) == hi - lo
        if getattr(optimizer, "accepts_param_subset", False):
            # the fused optimizers: one pass, 1/world folded in, the gradients zeroed behind the read
            plist = None if everything else [p for b in buckets for p in fg.bucket_params(b)]
            optimizer.step(grad_scale=scale, params=plist, zero_grads=True)
            return
        # any other optimizer object (torch's own: `fused_optimizer: false`, the CPU tests)
        views = [fg.flat[lo:hi]] if contiguous else [fg.flat[fg.buckets[b][0]:fg.buckets[b][1]] for b in buckets]
        if scale != 1.0:
            for v in views:
                v.mul_(scale)
        if everything:
            optimizer.step()
        else:
            mine = {id(p) for b in buckets for p in fg.bucket_params(b)}
            hidden = [(p, p.grad) for p in fg.params if id(p) not in mine]
            for p, _ in hidden:
                p.grad = None            # torch's optimizers skip parameters without a gradient
            try:
                optimizer.step()
            finally:
                for p, g in hidden:
                    p.grad = g
        for v in views:
            v.zero_()                    # == optimizer.zero_grad(set_to_none=False)

    def _timed_wait(self, idx, work, flat):
        if flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()              # NCCL / RCCL: makes the current stream wait, the host does not block
            e1.record()
            self._waits.append((idx, e0, e1))
        else:
            import time
            t0 = time.perf_counter()
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
        if self._gate_owner is not None:
            self._gate_owner.set_param_gate(self._prev_gate)
            self._gate_owner = None
        if self._F is not None:
            self._F.set_deferred_wgrads(())
        if self.cu_budget != 256:
            from ._lib import lib
            lib.gz_set_cu_budget(256)
            self.cu_budget = 256
