"""Fused multi-tensor Adam / RMSprop on the HIP kernels (csrc/gz_optim.hip), drop-in for the
``torch.optim.Adam`` / ``torch.optim.RMSprop`` objects the reference instantiates from its ``optimiser``
config nodes (conf/expt/dc_gan.yaml:12-15, wgan.yaml:24-26, wgan_gp.yaml:26-29, hologan.yaml:21-24).

Same constructor arguments, same ``param_groups`` keys and the same per-parameter state entries
(``step``, ``exp_avg``, ``exp_avg_sq`` / ``square_avg``), so optimizer ``state_dict``s interchange with
torch's and LR schedulers work unchanged.  Only the configurations the reference uses are supported
(no weight decay, amsgrad, momentum or centering); anything else raises.
"""
import ctypes

import torch

from . import functional as F
from ._lib import check, lib

MAX_TENSORS = 24


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _check_tensor(p):
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
        raise RuntimeError("fused optimizers need contiguous float32 GPU parameters")


class Adam(torch.optim.Optimizer):
    accepts_grad_scale = True
    accepts_param_subset = True
    accepts_sink_sources = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unused):
        if weight_decay or amsgrad:
            raise NotImplementedError("fused Adam: weight_decay / amsgrad are outside the reference's configs")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._step_py = {}          # id(param) -> step count as a Python int (mirror of state[p]["step"])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_py = {}          # re-read from the loaded tensors at the next step

    def _step_from_slabs(self, sources, grad_scale):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        max_t, nb = lib.gz_adam_src_max_tensors(), lib.gz_adam_src_table_bytes()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            todo = [(p, sources[id(p)][1]) for p in group["params"] if id(p) in sources]
            if not todo:
                continue
            for p, _ in todo:
                st = self.state[p]
                if not st:
                    _check_tensor(p)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            torch._foreach_add_([self.state[p]["step"] for p, _ in todo], 1.0)
            by_step = {}
            for p, srcs in todo:
                k = self._step_py.get(id(p))
                k = int(self.state[p]["step"].item()) if k is None else k + 1
                self._step_py[id(p)] = k
                by_step.setdefault(k, []).append((p, srcs))
            for step, items in by_step.items():
                # most slabs first (see functional.flush_grad_sinks): the edge layers' long slab chains start with the bulk
                items.sort(key=lambda it: -sum(nz for (_, nz, _) in it[1]))
                for i in range(0, len(items), max_t):
                    table = (ctypes.c_char * nb)()
                    for p, srcs in items[i:i + max_t]:
                        st = self.state[p]
                        for (slabs, nz, stride) in srcs:
                            check(lib.gz_adam_src_add(table, ctypes.c_void_p(p.data_ptr()),
                                                      ctypes.c_void_p(st["exp_avg"].data_ptr()),
                                                      ctypes.c_void_p(st["exp_avg_sq"].data_ptr()), p.numel(),
                                                      ctypes.c_void_p(slabs.data_ptr()), nz, stride), "adam_src_add")
                    check(lib.gz_adam_step_from_slabs(table, float(group["lr"]), float(beta1), float(beta2),
                                                      float(group["eps"]), step, float(grad_scale), stream),
                          "adam_step_from_slabs")
                for p, _ in items:
                    F.invalidate(p)

    def make_capturable(self):
        """Move the step counter to the device so that ``step()`` can be captured in a HIP graph and replayed
        (harness.GraphedTrainer): one float[3] per param group {step, 1 - beta1^step, sqrt(1 - beta2^step)} advanced
        by gz_adam_tick; the per-parameter ``state['step']`` entries become views of its first element (what
        torch.optim.Adam(capturable=True) keeps as 0-dim device tensors).  From here on the optimizer steps whole param
        groups from ``p.grad`` only: callers that look at ``accepts_param_subset`` / ``accepts_sink_sources``
        (ddp.GradSync, harness.Trainer) take their generic paths."""
        self.accepts_param_subset = self.accepts_sink_sources = False
        for group in self.param_groups:
            params = [p for p in group["params"]]
            if not params:
                continue
            steps = {float(self.state[p]["step"]) for p in params if self.state.get(p)}
            if len(steps) > 1:
                raise RuntimeError("capturable fused Adam needs one step count per param group")
            step = steps.pop() if steps else 0.0
            beta1, beta2 = group["betas"]
            tick = torch.tensor([step, 1.0 - beta1 ** step if step else 0.0, (1.0 - beta2 ** step) ** 0.5 if step else 0.0],
                                dtype=torch.float32, device=params[0].device)
            group["capturable"] = True
            group["_tick"] = tick
            for p in params:
                _check_tensor(p)
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = tick[0]

    def _step_capturable(self, group, grad_scale, stream, only=None, zero_grads=False):
        beta1, beta2 = group["betas"]
        tick = group["_tick"]
        plist = [p for p in group["params"] if p.grad is not None and (only is None or id(p) in only)]
        if not plist:
            return
        check(lib.gz_adam_tick(ctypes.c_void_p(tick.data_ptr()), float(beta1), float(beta2), stream), "adam_tick")
        for i in range(0, len(plist), MAX_TENSORS):
            chunk = plist[i:i + MAX_TENSORS]
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
            numel = (ctypes.c_longlong * len(chunk))(*[p.numel() for p in chunk])
            check(lib.gz_adam_step_dev(len(chunk), _ptr_array(chunk), _ptr_array(grads),
                                       _ptr_array([self.state[p]["exp_avg"] for p in chunk]),
                                       _ptr_array([self.state[p]["exp_avg_sq"] for p in chunk]), numel,
                                       float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                       ctypes.c_void_p(tick.data_ptr()), float(grad_scale), int(zero_grads), stream),
                  "adam_step_dev")
        for p in plist:
            F.invalidate(p)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, params=None, zero_grads=False, sink_sources=None):
        """``params``: step only these (ddp.GradSync steps one gradient bucket at a time, as its all-reduce lands);
        ``zero_grads``: the kernel also overwrites the gradients it has read with 0 (the flat exchange buffer);
        ``sink_sources`` (functional.take_grad_sinks): parameters whose gradient exists only as unreduced slabs -- the
        kernel sums them itself (gz_adam_step_from_slabs; same bits as reduce-then-step)."""
        loss = closure() if closure is not None else None
        if (params is not None or sink_sources) and not self.accepts_sink_sources:
            # a capturable group shares ONE device step counter, ticked once per step(): a per-bucket step would advance
            # it several times per optimizer step, and the slab path reads the step count on the host
            raise RuntimeError("capturable fused Adam steps whole param groups from p.grad only (no params= subset, "
                               "no sink_sources=)")
        if sink_sources:
            self._step_from_slabs(sink_sources, grad_scale)
        if getattr(self, "_pack_group", None) is None:      # the conv weights this optimizer rewrites re-pack together
            self._pack_group = F.register_pack_group([p for g in self.param_groups for p in g["params"]])
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        only = None if params is None else {id(p) for p in params}
        for group in self.param_groups:
            if group.get("_tick") is not None:
                self._step_capturable(group, grad_scale, stream, only, zero_grads)
                continue
            beta1, beta2 = group["betas"]
            by_step = {}
            todo = []
            for p in group["params"]:
                if p.grad is None or (only is not None and id(p) not in only):
                    continue
                st = self.state[p]
                if not st:
                    _check_tensor(p)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                todo.append((p, st))
            if not todo:
                continue
            # the per-parameter ``state['step']`` tensors (torch.optim.Adam's layout: checkpoints interchange) advance
            # with ONE foreach call; their values are mirrored as Python ints so that no .item() is needed per step
            torch._foreach_add_([st["step"] for _, st in todo], 1.0)
            for p, st in todo:
                k = self._step_py.get(id(p))
                k = int(st["step"].item()) if k is None else k + 1
                self._step_py[id(p)] = k
                by_step.setdefault(k, []).append(p)
            for step, plist in by_step.items():
                for i in range(0, len(plist), MAX_TENSORS):
                    chunk = plist[i:i + MAX_TENSORS]
                    grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                    numel = (ctypes.c_longlong * len(chunk))(*[p.numel() for p in chunk])
                    check(lib.gz_adam_step(len(chunk), _ptr_array(chunk), _ptr_array(grads),
                                           _ptr_array([self.state[p]["exp_avg"] for p in chunk]),
                                           _ptr_array([self.state[p]["exp_avg_sq"] for p in chunk]), numel,
                                           float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), step,
                                           float(grad_scale), int(zero_grads), stream), "adam_step")
                for p in plist:
                    F.invalidate(p)         # packed conv weights of p are stale now
        return loss


class RMSprop(torch.optim.Optimizer):
    accepts_grad_scale = True
    accepts_param_subset = True

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False, **unused):
        if weight_decay or momentum or centered:
            raise NotImplementedError("fused RMSprop: weight_decay / momentum / centered are outside the "
                                      "reference's configs")
        defaults = dict(lr=lr, momentum=0, alpha=alpha, eps=eps, centered=False, weight_decay=0, capturable=False,
                        foreach=None, maximize=False, differentiable=False)
        super().__init__(params, defaults)

    def make_capturable(self):
        """RMSprop's update does not depend on the step count: the kernel launch is replayable as it is (the
        ``state['step']`` counters then count captures, not replays)."""
        for group in self.param_groups:
            group["capturable"] = True

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, params=None, zero_grads=False):
        loss = closure() if closure is not None else None
        if getattr(self, "_pack_group", None) is None:      # the conv weights this optimizer rewrites re-pack together
            self._pack_group = F.register_pack_group([p for g in self.param_groups for p in g["params"]])
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        only = None if params is None else {id(p) for p in params}
        for group in self.param_groups:
            plist = []
            for p in group["params"]:
                if p.grad is None or (only is not None and id(p) not in only):
                    continue
                _check_tensor(p)
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                plist.append(p)
            for i in range(0, len(plist), MAX_TENSORS):
                chunk = plist[i:i + MAX_TENSORS]
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                numel = (ctypes.c_longlong * len(chunk))(*[p.numel() for p in chunk])
                check(lib.gz_rmsprop_step(len(chunk), _ptr_array(chunk), _ptr_array(grads),
                                          _ptr_array([self.state[p]["square_avg"] for p in chunk]), numel,
                                          float(group["lr"]), float(group["alpha"]), float(group["eps"]),
                                          float(grad_scale), int(zero_grads), stream), "rmsprop_step")
            for p in plist:
                F.invalidate(p)
        return loss


FUSED_TARGETS = {"torch.optim.Adam": "lightning_gan_zoo_amd.optim.Adam",
                 "torch.optim.RMSprop": "lightning_gan_zoo_amd.optim.RMSprop",
                 "torch.optim.adam.Adam": "lightning_gan_zoo_amd.optim.Adam",
                 "torch.optim.rmsprop.RMSprop": "lightning_gan_zoo_amd.optim.RMSprop"}


def fused_node(node, params):
    """Map an ``optimiser`` config node that targets torch's Adam / RMSprop onto the fused classes when
    every parameter lives on the GPU and the node uses only supported options; otherwise return it as is."""
    target = node.get("_target_")
    if target not in FUSED_TARGETS or not all(p.is_cuda for p in params):
        return node
    if any(node.get(k) for k in ("weight_decay", "amsgrad", "momentum", "centered")):
        return node
    out = dict(node)
    out["_target_"] = FUSED_TARGETS[target]
    return out
