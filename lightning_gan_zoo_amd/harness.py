"""Minimal training harness with the semantics of the Lightning generation the reference
targets: one optimizer is active per batch, cycling
D x disc_freq then G x gen_freq; ``toggle_optimizer`` freezes the other network's parameters
during ``training_step``; then ``loss.backward(); optimizer.step(); optimizer.zero_grad()``.

``LightningModule`` is pytorch_lightning's when that package is importable, otherwise a small
stand-in offering what the step classes and callbacks use (``log``, ``device``).
"""
import contextlib

import torch
from torch import nn

try:  # pragma: no cover - pytorch_lightning is not installed in the build image
    import pytorch_lightning as _pl
    _PLBase = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _PLBase = None
    HAVE_LIGHTNING = False


class _StandInLightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.logged = {}
        self.current_epoch = 0
        self.global_step = 0

    def log(self, key, value, *args, **kwargs):
        self.logged[key] = value.detach() if torch.is_tensor(value) else value

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")


LightningModule = _PLBase if HAVE_LIGHTNING else _StandInLightningModule


@contextlib.contextmanager
def few_host_threads(n=1):
    """Run tiny host-side tensor ops (noise draw, staging copy) on ``n`` intra-op threads.

    With the default thread count (= cores) every small CPU op wakes the whole OpenMP team, whose
    idle workers then spin; under a container CPU quota that gets the process throttled for tens of
    milliseconds every few steps -- the GPU queue starves although the host does almost nothing
    (measured on the MI355X box: 60-300 ms stalls per step, gone with one thread).  The values drawn
    do not depend on the thread count (CPU random kernels consume the generator serially)."""
    old = torch.get_num_threads()
    if old != n:
        torch.set_num_threads(n)
    try:
        yield
    finally:
        if old != n:
            torch.set_num_threads(old)


def draw_on_host(fn, device):
    """``fn()`` draws a small tensor on the host generator; returns it on ``device``."""
    with few_host_threads(1):
        t = fn()
        return _stager.to_device(t, device)


class HostStager:
    """Host -> device hand-off for small per-step tensors drawn on the HOST generator (latent noise,
    gradient-penalty alpha).  The reference does ``sample(...).to(device)`` from pageable memory
    (core/lightning_module.py:107-108); on ROCm a pageable hipMemcpy drains the stream and stalls
    for tens of ms every few calls (measured: 50-180 ms, i.e. more than the whole G+D pair), so the
    same values go through a small ring of pinned buffers and an asynchronous copy instead.  The
    host RNG stream, the values and the device they land on are unchanged."""

    def __init__(self, depth=16):     # deep enough that a buffer is only revisited after several steps (no host wait)
        self.depth = depth
        self.slots = {}     # (shape, dtype) -> [buffers, events, cursor]

    def to_device(self, t, device):
        device = torch.device(device)
        if device.type != "cuda" or t.is_cuda:
            return t.to(device)
        key = (tuple(t.shape), t.dtype)
        slot = self.slots.get(key)
        if slot is None:
            slot = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(self.depth)],
                    [None] * self.depth, 0]
            self.slots[key] = slot
        bufs, events, cur = slot
        if events[cur] is not None:
            events[cur].synchronize()          # the copy that last used this buffer has completed
        bufs[cur].copy_(t)
        out = self._device_read(bufs[cur], device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        events[cur] = ev
        slot[2] = (cur + 1) % self.depth
        return out


    @staticmethod
    def _device_read(pinned, device):
        """The device copies the words out of the pinned buffer itself (gz_copy_words: a kernel on the compute stream
        reading device-mapped host memory) -- hipMemcpyAsync's ENQUEUE cost the host 180 us per call while the stream
        was busy, 0.35 ms of host time per G+D pair.  Falls back to the asynchronous copy for odd sizes."""
        nbytes = pinned.numel() * pinned.element_size()
        if nbytes % 4 or not nbytes or HostStager.zero_copy is False:
            return pinned.to(device, non_blocking=True)
        from ._lib import check, lib
        import ctypes
        with torch.cuda.device(device):
            out = torch.empty(pinned.shape, dtype=pinned.dtype, device=device)
            check(lib.gz_copy_words(ctypes.c_void_p(pinned.data_ptr()), ctypes.c_void_p(out.data_ptr()), nbytes // 4,
                                    ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "copy_words")
        return out

    zero_copy = True


_stager = HostStager()


def host_to_device(t, device):
    return _stager.to_device(t, device)


def toggle_optimizer(module, optimizer_idx):
    """Only the active network's parameters require grad during a training_step."""
    for p in module.discriminator.parameters():
        p.requires_grad_(optimizer_idx == 0)
    for p in module.generator.parameters():
        p.requires_grad_(optimizer_idx == 1)


def optimizer_schedule(frequencies):
    """Lightning's 'frequency' rule: optimizer i is used for frequencies[i] consecutive batches."""
    order = []
    for idx, f in enumerate(frequencies):
        order.extend([idx] * int(f))
    return order


def accumulation_factor(schedule, epoch):
    """Lightning's ``accumulate_grad_batches`` argument as the reference builds it (run_network.py:61-68): the int 1, or
    ``{start_epoch: factor}`` -- GradientAccumulationScheduler semantics: the factor of the largest key <= epoch, 1 before
    the first key."""
    if schedule is None:
        return 1
    if isinstance(schedule, int):
        return max(1, schedule)
    factor = 1
    for start in sorted(int(k) for k in schedule):
        if epoch >= start:
            factor = int(schedule[start] if start in schedule else schedule[str(start)])
    return max(1, factor)


class Trainer:
    """Drives ``module.training_step`` over an iterable of batches with optimizer alternation.

    ``grad_sync`` (optional, see ddp.GradSync) averages the active network's gradients over the
    data-parallel ranks between backward and the optimizer step.

    ``accumulate_grad_batches`` (reference run_network.py:61-68; int or ``{start_epoch: factor}``) follows the training
    loop of the Lightning generation the reference targets: the optimizer of a batch is chosen by the running batch
    count and the ``frequency`` entries as always; the batch's loss is divided by the factor; the optimizer steps (and
    its gradients are cleared) only on batches with ``(batch index in the epoch + 1) % factor == 0`` or on the last
    batch of an epoch -- on every other batch the gradients just accumulate, and under data parallelism nothing is
    exchanged (Lightning's ``block_ddp_sync_behaviour``): the all-reduce runs on the stepping batch only."""

    def __init__(self, module, grad_sync=None, grad_sinks=True, accumulate_grad_batches=1):
        self.module = module
        self.optim = module.configure_optimizers()
        self.order = optimizer_schedule([o["frequency"] for o in self.optim])
        self.grad_sync = grad_sync
        self.batch_idx = 0                # running batch count (Lightning's total_batch_idx): selects the optimizer
        self.epoch_batch_idx = 0          # batch index inside the epoch: selects the stepping batches
        self.epoch = 0
        self.accumulate_grad_batches = accumulate_grad_batches
        self._ones = {}
        # weight gradients straight into p.grad (functional.set_grad_sinks): on for the duration of each step on the
        # GPU path -- ONE slab-reduction launch per backward pass (per gradient bucket under data parallelism) instead
        # of one per layer plus autograd's `grad += new` launches
        self._F = None
        if grad_sinks and any(p.is_cuda for p in module.parameters()):
            from . import functional as F
            self._F = F

    def active_optimizer(self, batch_idx=None):
        i = self.batch_idx if batch_idx is None else batch_idx
        return self.order[i % len(self.order)]

    def step(self, batch, last_in_epoch=False):
        idx = self.active_optimizer()
        m = self.module
        factor = accumulation_factor(self.accumulate_grad_batches, self.epoch)
        stepping = factor == 1 or (self.epoch_batch_idx + 1) % factor == 0 or last_in_epoch
        toggle_optimizer(m, idx)
        if self.grad_sync is not None:
            self.grad_sync.before_step(idx, exchange=stepping)
        F = self._F
        if F is not None:
            prev = F.set_grad_sinks(True)
        opt = self.optim[idx]["optimizer"]
        taken = None
        try:
            loss = m.training_step(batch, self.batch_idx, idx)
            gen = getattr(m, "generator", None)
            if F is not None and hasattr(gen, "prefetch_view") and torch.is_tensor(batch[0]):
                # HoloGAN: the NEXT forward's view matrices (0.4-0.5 ms of numpy on the host) now, with this step's whole
                # forward queued on the GPU.  (First placed at the END of the step; the kernel trace shows 0.2 ms idle at
                # every step boundary there -- under the profiler only: unprofiled the host is 7 ms per cycle ahead and
                # both placements measure 17.50 ms, tools/boundary_probe.py / tools/ab_expt.sh.  Kept here, where it
                # cannot land in front of an empty queue.)
                gen.prefetch_view(len(batch[0]))
            if factor == 1 and loss.dim() == 0 and loss.is_cuda:
                # d loss / d loss from a cached one: autograd otherwise fills a fresh scalar per backward (a launch each)
                one = self._ones.get((loss.device, loss.dtype))
                if one is None:
                    one = self._ones[(loss.device, loss.dtype)] = torch.ones((), device=loss.device, dtype=loss.dtype)
                loss.backward(gradient=one)
            else:
                (loss if factor == 1 else loss / factor).backward()
            if self.grad_sync is not None:
                # still inside the sink region: GradSync sums the slabs bucket by bucket, runs the weight-gradient
                # launches it postponed (deferred tail) and issues what the backward hooks have not issued yet
                self.grad_sync.after_backward(idx, opt, exchange=stepping)
            elif F is not None:
                if stepping and factor == 1 and getattr(opt, "accepts_sink_sources", False):
                    # the fused optimizer sums the weight-gradient slabs itself: no reduce launch, no gradient tensor
                    taken = F.take_grad_sinks([p for g in opt.param_groups for p in g["params"]])
                F.flush_grad_sinks()
        except BaseException:
            if F is not None:             # nothing half-built is reduced, nothing leaks into the next step's gradients
                F.discard_grad_sinks()
            if self.grad_sync is not None:
                self.grad_sync.abort_pass(idx)
            raise
        finally:
            if F is not None:
                F.set_grad_sinks(*prev)
        if self.grad_sync is None and stepping:
            if taken:
                opt.step(sink_sources=taken)
            else:
                opt.step()
            opt.zero_grad(set_to_none=True)
        self.batch_idx += 1
        self.epoch_batch_idx += 1
        return loss.detach(), idx

    def end_epoch(self):
        if self.grad_sync is not None:
            self.grad_sync.flush()        # a deferred optimizer step must use THIS epoch's learning rate
        import warnings
        for o in self.optim:
            sch = o.get("lr_scheduler")
            if sch is not None:
                with warnings.catch_warnings():
                    # Lightning steps EVERY scheduler at the epoch boundary, also that of an optimizer whose turn has
                    # not come yet (a one-batch epoch only runs the discriminator): torch's order check does not apply
                    warnings.filterwarnings("ignore", message="Detected call of `lr_scheduler.step\\(\\)` before")
                    sch.step()
        self.epoch += 1
        self.epoch_batch_idx = 0
        if hasattr(self.module, "current_epoch"):
            try:
                self.module.current_epoch += 1
            except AttributeError:
                pass

    def finish(self):
        if self.grad_sync is not None:
            self.grad_sync.flush()


class GraphedTrainer(Trainer):
    """Trainer whose per-optimizer step (training_step + backward + optimizer.step) is captured once in a HIP graph
    and replayed: the ~150-900 kernel launches of a step cost one graph launch on the host, which matters for the
    launch-heavy experiments (R1 ResNets: 13 ms of Python / launch time per 15 ms cycle) and on slow hosts.

    What stays outside the graph, in the reference's order, is exactly the host-side work: the latent noise (and
    WGAN-GP's alpha) is still drawn from the HOST generator each step and copied into static device buffers the
    captured step reads; the batch is copied into a static buffer unless it already is that buffer.
    HOLOGAN's per-step view matrices are host work too (numpy generator + 4x4 inverses): they are drawn in the
    reference's order and staged into a static buffer the captured generator reads (``Generator.staged_minv``).
    The learning rate is a by-value kernel argument of the captured optimizer launch, so a graph is keyed by the
    learning rates (and the batch shape) it was captured with: after ``end_epoch()`` has moved a scheduler, or when
    the data set's last batch is smaller, the step is captured again instead of replaying stale arguments.
    Restrictions (enforced): single process (no GradSync), fused optimizers (Adam with a device-side step counter,
    RMSprop)."""

    def __init__(self, module, warmup=2, grad_sync=None):
        if grad_sync is not None:
            raise RuntimeError("GraphedTrainer is single-process: the data-parallel gradient exchange is not captured")
        super().__init__(module)
        from . import functional as F
        self._F = F
        self.warmup = warmup
        self.graphs = {}        # (optimizer_idx, batch shape, learning rates) -> (graph, loss)
        self.seen = {}
        self.static_minv = None
        self.static_batch = None
        self.static_noise = None
        self.static_alpha = None
        for o in self.optim:
            if not hasattr(o["optimizer"], "make_capturable"):
                raise RuntimeError("GraphedTrainer needs the fused optimizers (lightning_gan_zoo_amd.optim)")

    def _stage(self, batch):
        m = self.module
        real, labels = batch
        if self.static_batch is None or self.static_batch[0].shape != real.shape:
            self.static_batch = (real.clone(), labels.clone() if torch.is_tensor(labels) else labels)
            self.static_noise = self.static_alpha = self.static_minv = None
        elif real.data_ptr() != self.static_batch[0].data_ptr():
            with torch.no_grad():         # R1 turns the batch into a leaf that requires grad
                self.static_batch[0].copy_(real, non_blocking=True)
        # host RNG draws in the reference's order: z (lightning_module.py:107-108), then alpha (utils.py:41)
        z = draw_on_host(lambda: m.noise_distn.sample((len(real), m.cfg.model.noise_dim)), real.device)
        if self.static_noise is None:
            self.static_noise = z.clone()
        else:
            self.static_noise.copy_(z, non_blocking=True)
        gen = getattr(m, "generator", None)
        if hasattr(gen, "sample_view") and hasattr(gen, "staged_minv"):        # HoloGAN: numpy draw + host inverses
            from .core.models.hologan_generator import view_inverse_matrices
            minv = draw_on_host(lambda: view_inverse_matrices(gen.sample_view(len(real))).reshape(len(real), 16)
                                .contiguous(), real.device)
            if self.static_minv is None:
                self.static_minv = minv.clone()
            else:
                self.static_minv.copy_(minv, non_blocking=True)
            gen.staged_minv = self.static_minv
        if hasattr(m, "gp_alpha") and self.active_optimizer() == 0:
            a = draw_on_host(lambda: torch.rand((len(real), 1, 1, 1)), real.device)
            if self.static_alpha is None:
                self.static_alpha = a.clone()
            else:
                self.static_alpha.copy_(a, non_blocking=True)

    def _body(self, idx):
        m = self.module
        prev = self._F.set_grad_sinks(True)
        try:
            loss = m.training_step(self.static_batch, self.batch_idx, idx)
            loss.backward()
            self._F.flush_grad_sinks()
        except BaseException:
            self._F.discard_grad_sinks()
            raise
        finally:
            self._F.set_grad_sinks(*prev)
        self.optim[idx]["optimizer"].step()
        return loss

    def step(self, batch):
        idx = self.active_optimizer()
        m = self.module
        toggle_optimizer(m, idx)
        self._stage(batch)
        m.sample_noise = lambda n: self.static_noise          # the captured step reads the static buffers
        if hasattr(m, "gp_alpha"):
            m.gp_alpha = self.static_alpha
        opt = self.optim[idx]["optimizer"]
        n = self.seen.get(idx, 0)
        self.seen[idx] = n + 1
        key = (idx, tuple(self.static_batch[0].shape), tuple(float(g["lr"]) for g in opt.param_groups))
        if n < self.warmup:                                    # eager warm-up steps (allocator, lazy state)
            loss = self._body(idx)
            opt.zero_grad(set_to_none=True)
        elif key not in self.graphs:
            opt.make_capturable()
            self._F.set_pack_cache(False)
            opt.zero_grad(set_to_none=True)
            if torch.is_tensor(self.static_batch[0]):
                self.static_batch[0].grad = None     # R1 asks for d/d(real): let the capture (re)create it
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # no cyclic-garbage collection while capturing: a collection that frees device tensors / events left
            # over from earlier work would call into the HIP runtime in the middle of the capture (abort)
            import gc
            was_enabled = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                with torch.cuda.graph(g):
                    loss = self._body(idx)
            finally:
                if was_enabled:
                    gc.enable()
            self.graphs[key] = (g, loss)
            # the capture itself does not execute: run it once so that this step takes effect
            g.replay()
        else:
            g, loss = self.graphs[key]
            g.replay()
        self.batch_idx += 1
        return loss.detach(), idx
