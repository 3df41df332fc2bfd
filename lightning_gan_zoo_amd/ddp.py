"""Data-parallel gradient exchange for the G+D step: one process per GPU, RCCL over xGMI.

Reference behaviour (run_network.py:66, ``accelerator="ddp"``): torch DDP averages the gradients of
the network being optimised across ranks inside ``loss.backward()``; BatchNorm statistics stay
per-rank (no sync_batchnorm); rank 0's buffers are what a checkpoint / evaluation sees.

MI355X-first shape of the same exchange:
  * each network's gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), so the
    exchange is a single all-reduce per step (D: 11 MB, G: 51 MB for the 64-feature nets) -- large
    messages are what the point-to-point xGMI links want, and there is no per-bucket launch cost;
  * the all-reduce is issued asynchronously right after backward and is only waited for when the
    network is next USED (a forward-pre hook on the module): the discriminator's exchange + optimizer
    step therefore overlap the generator forward that opens the next training_step, which does not
    read the discriminator;
  * the optimizer step itself is deferred to that same point, so the result is bit-identical to the
    "all-reduce, then step" order of DDP.

Works on any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


class _FlatGrads:
    def __init__(self, params):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, device=ref.device, dtype=ref.dtype)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def rebind(self):
        """Re-attach the views if something replaced / dropped ``p.grad``."""
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += p.numel()


class GradSync:
    """Plugs into harness.Trainer: ``before_step``, ``after_backward``, ``flush``."""

    def __init__(self, module, process_group=None, overlap=True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        # test hook: issue the collective even on a single rank (exercises the RCCL call path)
        self.always_reduce = bool(os.environ.get("GZ_DDP_ALWAYS_REDUCE")) and dist.is_initialized()
        self.nets = [module.discriminator, module.generator]     # optimizer_idx order
        self.flats = [_FlatGrads(list(n.parameters())) for n in self.nets]
        self.pending = [None, None]   # (work, optimizer)
        self.hooks = [n.register_forward_pre_hook(self._make_hook(i)) for i, n in enumerate(self.nets)]

    def _make_hook(self, idx):
        def hook(_module, _inputs):
            self.finalize(idx)
        return hook

    def before_step(self, optimizer_idx):
        if getattr(self.module, "mutates_discriminator_before_forward", False):
            self.finalize(0)     # WGAN clamps D's weights at the top of training_step
        self.flats[optimizer_idx].rebind()

    def after_backward(self, optimizer_idx, optimizer):
        fg = self.flats[optimizer_idx]
        fg.rebind()
        work = None
        if self.world > 1 or self.always_reduce:
            work = dist.all_reduce(fg.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending[optimizer_idx] = (work, optimizer)
        if not self.overlap:
            self.finalize(optimizer_idx)

    def finalize(self, idx):
        item = self.pending[idx]
        if item is None:
            return
        self.pending[idx] = None
        work, optimizer = item
        fg = self.flats[idx]
        scale = 1.0
        if work is not None:
            work.wait()
            scale = 1.0 / self.world
        if scale != 1.0 and getattr(optimizer, "accepts_grad_scale", False):
            optimizer.step(grad_scale=scale)      # the fused optimizers fold the 1/world into their single pass
        else:
            if scale != 1.0:
                fg.flat.mul_(scale)
            optimizer.step()
        fg.flat.zero_()          # == optimizer.zero_grad(set_to_none=False), one memset

    def flush(self):
        for i in range(len(self.nets)):
            self.finalize(i)

    @torch.no_grad()
    def sync_buffers(self, src=0):
        """Broadcast rank ``src``'s norm buffers (what DDP's broadcast_buffers leaves on every rank);
        call before evaluation or checkpointing."""
        if self.world <= 1:
            return
        for net in self.nets:
            for b in net.buffers():
                dist.broadcast(b, src=src, group=self.group)

    def close(self):
        self.flush()
        for h in self.hooks:
            h.remove()
