"""lightning_gan_zoo_amd.functional, part 1: constants, C-ABI call helpers, the per-launch timer, packed-weight caches,
raw (non-differentiable) convolution launchers, gradient sinks and the parameter gate.  See the package docstring."""
import ctypes
import os
import weakref
from collections import namedtuple

import torch

from .._lib import check, lib

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3

Geom = namedtuple("Geom", "kh kw stride pad")
K4S2P1 = Geom(4, 4, 2, 1)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # the launchers take the caller's current stream explicitly; the raw query is ~10x cheaper than building a
    # torch.cuda.Stream object per launch (0.6 ms per step at ~300 launches)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("lightning_gan_zoo_amd: %s must live on the GPU (got %s); the HIP path has no CPU "
                           "fallback" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("lightning_gan_zoo_amd: %s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def out_size(n, g):
    return (n + 2 * g.pad - g.kh) // g.stride + 1


# ---------------------------------------------------------------------------
# optional per-launch timing (bench.py): HIP events recorded on the stream the kernel is launched
# on, around each implicit-GEMM launch.  Off by default; costs nothing when off.
# ---------------------------------------------------------------------------
class KernelTimer:
    def __init__(self, detail=False):
        self.records = []      # (label, flops, start_event, end_event)
        self.detail = detail   # per-shape labels, and the 3-D convolutions / plain GEMMs are timed too
        self.enabled = True    # bench.py samples every few cycles: two event records per launch cost ~5 % of a step

    def summary(self):
        """{label: (launches, total_ms, total_flops)} -- call after a device synchronize."""
        agg = {}
        for label, flops, s, e in self.records:
            n, ms, fl = agg.get(label, (0, 0.0, 0.0))
            agg[label] = (n + 1, ms + s.elapsed_time(e), fl + flops)
        return agg


_timer = None
_TILES = {0: "128x128", 1: "128x64", 2: "128x32", 3: "64x64", 4: "256x256", 5: "256x128", 6: "512x64", 7: "128x256", 8: "256x64",
          9: "256x(4x32)"}


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def _timed(op, shape, geom, flops, launch):
    if _timer is None or not _timer.enabled:
        return launch()
    N, C, H, W, K, OH, OW = shape
    tile = lib.gz_conv2d_tile(op, N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride)
    label = "igemm<%s,%s>" % (("F", "Dg", "Wg")[op], _TILES.get(tile, "?"))
    if _timer.detail:
        label += " k%ds%dp%d N%d C%d H%d K%d OH%d" % (geom.kh, geom.stride, geom.pad, N, C, H, K, OH)
    return _timed_as(label, flops, launch)


def _timed_as(label, flops, launch):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = launch()
    e.record()
    _timer.records.append((label, flops, s, e))
    return r


def _timed_detail(label_fn, flops, launch):
    """Ops outside the 2-D convolution family: timed only by a detail timer (tools/layer_times.py)."""
    if _timer is None or not _timer.detail or not _timer.enabled:
        return launch()
    return _timed_as(label_fn(), flops, launch)


# ---------------------------------------------------------------------------
# packed weights (GEMM-B images).  Cached per Parameter object and version so the three
# discriminator passes of one step share one pack.
# ---------------------------------------------------------------------------
_pack_cache = {}
_pack_cache_enabled = True     # harness.GraphedTrainer turns it off: a captured step must contain its pack kernels


def set_pack_cache(enabled):
    global _pack_cache_enabled
    _pack_cache_enabled = bool(enabled)
    clear_pack_cache()
    _pack3_cache.clear()


# Pack groups: the conv weights one optimizer updates.  After its step every packed image of the group is stale at
# once, and the first miss re-packs them ALL in one launch (gz_conv2d_pack_multi) into their existing buffers instead
# of one launch per image as the step walks through the layers (15 launches per DCGAN pair; they were 0.5 % of the
# bs 512 pair and 1.4 % of the bs 128 pair).  Images keep their buffers; the job table lives on the device and is
# rebuilt only when the set of images changes (the first steps).
_pack_group_of = {}       # parameter data_ptr -> group id
_group_keys = {}          # group id -> set of (data_ptr, kind)
_group_table = {}         # group id -> (signature, device table, number of jobs, total workgroups)
_NO_PACK_GROUPS = bool(os.environ.get("GZ_NO_PACK_GROUPS"))      # experiment: one launch per image, as in round 1


class _PackEntry:
    __slots__ = ("ref", "version", "shape", "wp", "geom", "stale")

    def __init__(self, w, wp, geom):
        self.ref, self.version, self.shape, self.wp, self.geom, self.stale = weakref.ref(w), w._version, tuple(w.shape), wp, geom, False


def register_pack_group(params):
    """Called by the fused optimizers with the parameters they update; returns the group id."""
    gid = len(_group_keys) + 1
    _group_keys[gid] = set()
    for p in params:
        if p.dim() == 4 and p.is_cuda:
            _pack_group_of[p.data_ptr()] = gid
    return gid


def _pack_one(w, wp, kind, geom):
    K, C, KH, KW = w.shape
    if kind == "f":
        check(lib.gz_conv2d_pack_fwd(_p(w), _p(wp), K, C, KH, KW, _stream()), "conv2d_pack_fwd")
    else:
        check(lib.gz_conv2d_pack_dgrad(_p(w), _p(wp), K, C, KH, KW, geom.stride, geom.pad, _stream()),
              "conv2d_pack_dgrad")


def _repack_group(gid):
    """Re-pack every STALE image of the group in one launch.  Normally that is the whole group (the optimizer stepped
    all of its parameters); under ddp.GradSync the parameters are stepped bucket by bucket as their all-reduce lands,
    so a miss re-packs what has been stepped so far and the rest follows with its own bucket -- a job table is kept
    per distinct stale set (two for the DCGAN generator: main buckets, deferred tail)."""
    live = []
    for key in sorted(_group_keys[gid]):
        e = _pack_cache.get(key)
        w = e.ref() if e is not None else None
        if w is None or w.data_ptr() != key[0] or tuple(w.shape) != e.shape:
            _group_keys[gid].discard(key)
            _pack_cache.pop(key, None)
            continue
        if e.stale or e.version != w._version:
            live.append((key, e, w))
    if not live:
        return
    sig = tuple((key, e.wp.data_ptr()) for key, e, _ in live)
    tabs = _group_table.setdefault(gid, {})
    tab = tabs.get(sig)
    if tab is None:
        if len(tabs) > 8:
            tabs.clear()
        nb = lib.gz_conv2d_pack_job_bytes()
        host = (ctypes.c_char * (nb * len(live)))()
        block0 = 0
        for i, (key, e, w) in enumerate(live):
            K, C, KH, KW = e.shape
            n = lib.gz_conv2d_pack_job(ctypes.c_void_p(ctypes.addressof(host) + i * nb), _p(w), _p(e.wp),
                                       0 if key[1] == "f" else 1, K, C, KH, KW,
                                       e.geom.stride if e.geom is not None else 1, e.geom.pad if e.geom is not None else 0,
                                       block0)
            check(min(n, 0), "conv2d_pack_job")
            block0 += n
        dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(live[0][2].device)
        tab = (dev, len(live), block0)
        tabs[sig] = tab
    check(lib.gz_conv2d_pack_multi(_p(tab[0]), tab[1], tab[2], _stream()), "conv2d_pack_multi")
    for _, e, w in live:
        e.version, e.stale = w._version, False


def _packed(w, kind, geom):
    pk = getattr(w, "_gz_packs", None)          # spectral_normalize_multi wrote both images next to w itself
    if pk is not None and pk["version"] == w._version and (kind == "f" or pk["geom"] == geom):
        return pk[kind]
    key = (w.data_ptr(), kind)
    cacheable = _pack_cache_enabled and isinstance(w, torch.nn.Parameter)
    e = _pack_cache.get(key) if cacheable else None
    if e is not None and (e.ref() is not w or e.shape != tuple(w.shape)):
        _pack_cache.pop(key, None)
        e = None
    if e is not None and e.version == w._version and not e.stale:
        return e.wp
    gid = None if (_NO_PACK_GROUPS or not cacheable) else _pack_group_of.get(key[0])
    if e is not None and gid is not None and key in _group_keys[gid]:
        _repack_group(gid)
        return e.wp
    K, C, KH, KW = w.shape
    if kind == "f":
        n = lib.gz_conv2d_pack_fwd_elems(K, C, KH, KW)
    else:
        n = lib.gz_conv2d_pack_dgrad_elems(K, C, KH, KW, geom.stride)
    wp = torch.empty(n, device=w.device, dtype=torch.float32)
    _pack_one(w, wp, kind, geom)
    if cacheable:
        _pack_cache[key] = _PackEntry(w, wp, geom)
        if gid is not None:
            _group_keys[gid].add(key)
    return wp


def clear_pack_cache():
    _pack_cache.clear()
    _group_table.clear()
    for keys in _group_keys.values():
        keys.clear()


def invalidate(w):
    """The packed images of `w` are stale: its memory was rewritten without a version bump (raw in-place kernels:
    clamp_, the fused optimizers).  Images that belong to a pack group keep their buffers and are re-packed together
    at the next use; the others are dropped."""
    ptr = w.data_ptr()
    grouped = not _NO_PACK_GROUPS and ptr in _pack_group_of
    for kind in ("f", "d"):
        e = _pack_cache.get((ptr, kind))
        if e is None:
            continue
        if grouped:
            e.stale = True
        else:
            _pack_cache.pop((ptr, kind), None)
    _pack3_cache.pop((ptr, "f"), None)
    _pack3_cache.pop((ptr, "d"), None)


# ---------------------------------------------------------------------------
# raw (non-differentiable) launchers
# ---------------------------------------------------------------------------
# GZ_POISON_SCRATCH=1 (tests): every scratch / workspace buffer starts as NaN, so a kernel that reads a part of it that
# this launch has not written shows up in the results instead of depending on what the allocator handed back
_POISON = bool(os.environ.get("GZ_POISON_SCRATCH"))


def _ws(nfloats, device):
    if _POISON:
        return torch.full((nfloats,), float("nan"), device=device, dtype=torch.float32)
    return torch.empty(nfloats, device=device, dtype=torch.float32)


def _scratch(nbytes, device):
    """Split-K scratch of one launch (include/gz_ops.h: gz_*_workspace_bytes); None when the op is not split."""
    if not nbytes:
        return None, 0
    return _ws(nbytes // 4, device), nbytes


def _conv_fwd_raw(x, w, bias, geom, act, slope):
    N, C, H, W = x.shape
    K = w.shape[0]
    OH, OW = out_size(H, geom), out_size(W, geom)
    y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32)
    wp = _packed(w, "f", geom)
    ws, nbytes = _scratch(lib.gz_conv2d_fwd_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                            geom.pad), x.device)
    _timed(0, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_fwd(_p(x), _p(wp), _p(bias), _p(y), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh, geom.kw,
                          geom.stride, geom.pad, act, slope, _stream()), "conv2d_fwd"))
    return y


_NO_BN_FUSE = bool(os.environ.get("GZ_NO_BN_FUSE"))      # experiment: statistics by a separate pass, as in round 1
_NORM_UNFUSED = bool(os.environ.get("GZ_BN_FINALIZE_LAUNCH"))    # experiment: the separate bn_finalize launch (round 4)


def _conv_fwd_stats_raw(x, w, geom):
    """(y, stats) -- the convolution and, from the same launch, the per-tile BatchNorm partial sums of y
    (gz_conv2d_fwd_stats); stats is None when the launch cannot carry them (split-K)."""
    N, C, H, W = x.shape
    K = w.shape[0]
    OH, OW = out_size(H, geom), out_size(W, geom)
    rows = 0 if _NO_BN_FUSE else lib.gz_conv2d_fwd_stats_rows(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                              geom.pad)
    if x.data_ptr() & 15:       # an offset view: the fused launch needs 16-byte rows, the plain one re-plans for itself
        rows = 0
    if rows <= 0:
        return _conv_fwd_raw(x, w, None, geom, ACT_NONE, 0.0), None
    y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32)
    stats = torch.empty((rows, K, 2), device=x.device, dtype=torch.float32)
    wp = _packed(w, "f", geom)
    ws, nbytes = _scratch(lib.gz_conv2d_fwd_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                            geom.pad), x.device)
    _timed(0, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_fwd_stats_ws(_p(x), _p(wp), _p(y), _p(stats), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh,
                                   geom.kw, geom.stride, geom.pad, _stream()), "conv2d_fwd_stats"))
    return y, stats


def _conv_dgrad_stats_raw(g, w, geom, hw):
    N, K, OH, OW = g.shape
    C = w.shape[1]
    H, W = hw
    rows = 0 if _NO_BN_FUSE else lib.gz_conv2d_dgrad_stats_rows(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                                geom.pad)
    if g.data_ptr() & 15:
        rows = 0
    if rows <= 0:
        return _conv_dgrad_raw(g, w, None, geom, hw, ACT_NONE, 0.0), None
    x = torch.empty((N, C, H, W), device=g.device, dtype=torch.float32)
    stats = torch.empty((rows, C, 2), device=g.device, dtype=torch.float32)
    wp = _packed(w, "d", geom)
    ws, nbytes = _scratch(lib.gz_conv2d_dgrad_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                              geom.pad), g.device)
    _timed(1, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_dgrad_stats_ws(_p(g), _p(wp), _p(x), _p(stats), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh,
                                     geom.kw, geom.stride, geom.pad, _stream()), "conv2d_dgrad_stats"))
    return x, stats


def _conv_dgrad_raw(g, w, bias, geom, hw, act, slope):
    N, K, OH, OW = g.shape
    C = w.shape[1]
    H, W = hw
    x = torch.empty((N, C, H, W), device=g.device, dtype=torch.float32)
    wp = _packed(w, "d", geom)
    ws, nbytes = _scratch(lib.gz_conv2d_dgrad_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride,
                                                              geom.pad), g.device)
    _timed(1, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_dgrad(_p(g), _p(wp), _p(bias), _p(x), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh, geom.kw,
                            geom.stride, geom.pad, act, slope, _stream()), "conv2d_dgrad"))
    return x


def _conv_dgrad_act_raw(g, y, act, slope, w, geom, hw):
    """Input gradient of ``act(conv(x, w))`` from the gradient ``g`` w.r.t. the activation's output and the saved output
    ``y``, the mask formed on load (gz_conv2d_dgrad_act); None when the shape does not take the fused kernel."""
    N, K, OH, OW = g.shape
    C = w.shape[1]
    H, W = hw
    shape = (N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride, geom.pad)
    if not lib.gz_conv2d_dgrad_act_fuses(*shape, act):
        return None
    g, y = _req(g), _req(y)
    wp = _packed(w, "d", geom)
    x = torch.empty((N, C, H, W), device=g.device, dtype=torch.float32)
    if (g.data_ptr() | y.data_ptr() | x.data_ptr() | wp.data_ptr()) & 15:
        return None
    _timed(1, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_dgrad_act(_p(g), _p(y), act, float(slope), _p(wp), _p(x), *shape, _stream()), "conv2d_dgrad_act"))
    return x


def _channel_sum_raw(g):
    """g.sum over every dimension but the channel (the bias gradient of a convolution), no autograd."""
    g = _req(g)
    N, C = g.shape[0], g.shape[1]
    inner = g.numel() // (N * C)
    if inner % 4:
        return g.sum([d for d in range(g.dim()) if d != 1])
    out = torch.empty(C, device=g.device, dtype=torch.float32)
    ws = _ws(max(lib.gz_norm_workspace_bytes(N, C) // 4, 1), g.device)
    check(lib.gz_channel_sum(_p(g), _p(out), _p(ws), N, C, inner, _stream()), "channel_sum")
    return out


class _ChannelSum(torch.autograd.Function):
    """Bias gradient while a graph is being recorded (double-backward branches): the same kernel, differentiable
    (its adjoint broadcasts the incoming vector back over n and the map)."""

    @staticmethod
    def forward(ctx, g):
        ctx.shape = tuple(g.shape)
        return _channel_sum_raw(g)

    @staticmethod
    def backward(ctx, v):
        view = (1, -1) + (1,) * (len(ctx.shape) - 2)
        return v.reshape(view).expand(ctx.shape)


def _conv_wgrad_raw(x, g, geom, with_bias=False):
    """dw (and, with_bias, the bias gradient db = g.sum((0, 2, 3)): in the same launch where the kernel reads all
    of g anyway -- gz_conv2d_wgrad_fuses_bias -- otherwise by a separate reduction)."""
    N, C, H, W = x.shape
    _, K, OH, OW = g.shape
    dw = torch.empty((K, C, geom.kh, geom.kw), device=x.device, dtype=torch.float32)
    nbytes = lib.gz_conv2d_wgrad_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw)
    ws = _ws(max(nbytes // 4, 1), x.device)
    db = None
    if with_bias and lib.gz_conv2d_wgrad_fuses_bias(N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride, geom.pad):
        db = torch.empty(K, device=x.device, dtype=torch.float32)
    _timed(2, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_wgrad(_p(x), _p(g), _p(dw), _p(db), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh, geom.kw,
                            geom.stride, geom.pad, _stream()), "conv2d_wgrad"))
    if with_bias:
        return dw, (db if db is not None else _channel_sum_raw(g))
    return dw


# ---------------------------------------------------------------------------
# gradient sinks (round 4): weight gradients go straight into ``p.grad``
# ---------------------------------------------------------------------------
# torch's AccumulateGrad adds every contribution to an existing ``p.grad`` with a launch of its own (the discriminator
# is applied twice per D step: 11 `add_` per DCGAN pair; under data parallelism ``p.grad`` is a view of the flat
# exchange buffer, so EVERY gradient pays one: 25 per pair), and every split weight-gradient launch is followed by its
# own slab reduction.  With sinks on (harness.Trainer / ddp.GradSync turn them on; the Lightning drop-in route and
# plain ``loss.backward()`` users keep autograd's behaviour), a first-order backward
#   * leaves the slabs of a split weight-gradient launch unreduced (gz_conv2d_wgrad_partial) and returns None to
#     autograd for that parameter,
#   * ``flush_grad_sinks()`` -- called once after backward, or per gradient bucket by GradSync -- sums the slabs of all
#     pending parameters in ONE launch (gz_reduce_multi), writing a fresh ``p.grad`` (beta 0) or accumulating into the
#     existing one (beta 1: gradient accumulation, the flat exchange buffer).
# Double-backward graphs (create_graph=True) never take this path.
class _SinkState:
    enabled = False
    pending = {}              # id(param) -> [param, [(slabs, nz, stride), ...]]
    deferred = frozenset()    # id(param): the weight-gradient LAUNCH itself is postponed to run_deferred_wgrads()
    deferred_jobs = []        # (param, x, g, geom) in arrival order


_sinks = _SinkState()


def set_grad_sinks(enabled):
    """Turn the direct-to-``p.grad`` weight-gradient path on / off; returns the previous state as a 1-tuple
    (``set_grad_sinks(*prev)`` restores it).  Whatever is pending is flushed first."""
    old = (_sinks.enabled,)
    flush_grad_sinks()
    _sinks.enabled = bool(enabled)
    return old


def grad_sinks_enabled():
    return _sinks.enabled


def discard_grad_sinks():
    """Drop every pending contribution WITHOUT launching anything (a step that raised half-way: its slabs must neither
    be reduced from half-built state nor leak into the next step's gradients)."""
    _sinks.pending.clear()
    _sinks.deferred_jobs.clear()


def set_deferred_wgrads(params=()):
    """ddp.GradSync: the convolution weight gradients of these parameters are not launched where backward reaches them
    but by ``run_deferred_wgrads()`` at the end of the pass.  They are the layers the NEXT forward needs last, so
    their bucket can be exchanged last -- and their launches then run behind the all-reduce of everything else, which
    the next forward needs first.  Same kernels on the same operands: results are bit-identical."""
    _sinks.deferred = frozenset(id(p) for p in params)
    if _sinks.deferred_jobs:
        run_deferred_wgrads()


def run_deferred_wgrads():
    """Launch the postponed weight gradients (in arrival order) into their parameters' sinks."""
    jobs, _sinks.deferred_jobs = _sinks.deferred_jobs, []
    keep, _sinks.deferred = _sinks.deferred, frozenset()
    try:
        for (w, x, g, geom) in jobs:
            if not _sink_conv_wgrad(w, x, g, geom):
                _sink_fail("deferred weight gradient")
    finally:
        _sinks.deferred = keep
    return len(jobs)


# Parameter gate (ddp.GradSync): the modules announce the parameters a layer is about to read -- ``ready(w, gamma, ..)``
# -- so that the gradient exchange + optimizer step of the PREVIOUS pass only has to have landed bucket by bucket, at
# the first layer that reads a bucket, instead of for the whole network at the top of its forward.
_param_gate = None


def set_param_gate(fn):
    global _param_gate
    old, _param_gate = _param_gate, fn
    return old


def ready(*params):
    if _param_gate is not None:
        _param_gate(params)


# WHEN a sunk parameter's gradient is complete is autograd's knowledge, not counted here: the parameter's AccumulateGrad
# node runs once per graph task, after every node that feeds it has run -- also when those nodes returned None for it,
# and also for nodes that a double backward (WGAN-GP, R1) created -- so a ``register_post_accumulate_grad_hook`` hook
# is the "every contribution of this pass has been queued" signal (ddp.GradSync flushes and issues a bucket from it;
# tests/test_runner_cpu.py::test_post_accumulate_hook_fires_once_for_none_gradients pins the torch behaviour).
# Rounds 4's own use counter ran inside Function.forward, where grad mode is always off, and never counted anything.


def _sink_conv_wgrad(w, x, g, geom):
    """The weight gradient of a convolution into the sink of parameter ``w``; False = not taken (the caller computes
    it the ordinary way).  x / g are the operands gz_conv2d_wgrad takes."""
    if not _sinks.enabled or not isinstance(w, torch.nn.Parameter) or (w.numel() & 3):
        return False
    if w.grad is not None and (w.grad.data_ptr() & 15 or not w.grad.is_contiguous() or w.grad.dtype != torch.float32):
        return False          # a foreign p.grad layout (not 16-byte aligned): autograd's accumulation takes it
    N, C, H, W = x.shape
    _, K, OH, OW = g.shape
    nbytes = lib.gz_conv2d_wgrad_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw)
    if not nbytes:
        return False
    if id(w) in _sinks.deferred:
        _sinks.deferred_jobs.append((w, x, g, geom))
        return True
    ws = _ws(nbytes // 4, x.device)
    dw = torch.empty((K, C, geom.kh, geom.kw), device=x.device, dtype=torch.float32)   # unsplit launches write here
    nz, stride = ctypes.c_int(0), ctypes.c_longlong(0)
    _timed(2, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_wgrad_partial(_p(x), _p(g), _p(dw), _p(ws), nbytes, N, C, H, W, K, OH, OW, geom.kh, geom.kw,
                                    geom.stride, geom.pad, ctypes.byref(nz), ctypes.byref(stride), _stream()),
        "conv2d_wgrad_partial"))
    src = (ws, nz.value, stride.value) if nz.value > 1 else (dw, 1, w.numel())
    if src[2] & 3:
        return False if nz.value <= 1 else _sink_fail("slab stride")
    _sinks.pending.setdefault(id(w), [w, []])[1].append(src)
    return True


def _sink_conv_wgrad_act(w, b, x, g, y, geom, act, slope):
    """First-order backward of ``act(conv(x, w) + b)`` for a layer whose input needs no gradient (the critics' first
    convolution, reference standard_networks.py:62-66), with sinks on: ONE launch (gz_conv2d_wgrad_act_partial) masks
    ``g`` with the saved output ``y`` on load and leaves weight- and bias-gradient slabs for the flush -- instead of
    act_bwd + wgrad + slab reduction + channel_sum (+ autograd's accumulation).  ``b`` is the bias Parameter or None.
    False = not taken."""
    if not _sinks.enabled or act not in (ACT_RELU, ACT_LRELU) or id(w) in _sinks.deferred:
        return False
    for p_ in (w, b):
        if p_ is None:
            continue
        if not isinstance(p_, torch.nn.Parameter) or (p_.numel() & 3):
            return False
        if p_.grad is not None and (p_.grad.data_ptr() & 15 or not p_.grad.is_contiguous() or p_.grad.dtype != torch.float32):
            return False
    N, C, H, W = x.shape
    _, K, OH, OW = g.shape
    shape = (N, C, H, W, K, OH, OW, geom.kh, geom.kw, geom.stride, geom.pad)
    if not lib.gz_conv2d_wgrad_act_fuses(*shape, act):
        return False
    g, y = _req(g), _req(y)
    nbytes = lib.gz_conv2d_wgrad_workspace_bytes(N, C, H, W, K, OH, OW, geom.kh, geom.kw)
    ws = _ws(nbytes // 4, x.device)
    nz, stride, boff = ctypes.c_int(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
    _timed(2, (N, C, H, W, K, OH, OW), geom, 2.0 * N * OH * OW * K * C * geom.kh * geom.kw, lambda: check(
        lib.gz_conv2d_wgrad_act_partial(_p(x), _p(g), _p(y), act, float(slope), _p(ws), nbytes, *shape, ctypes.byref(nz),
                                        ctypes.byref(stride), ctypes.byref(boff), _stream()),
        "conv2d_wgrad_act_partial"))
    _sinks.pending.setdefault(id(w), [w, []])[1].append((ws, nz.value, stride.value))
    if b is not None:
        _sinks.pending.setdefault(id(b), [b, []])[1].append((ws[boff.value:], nz.value, stride.value))
    return True


def _sink_grad(p, g):
    """A COMPLETE gradient contribution ``g`` of parameter ``p`` (a bias, a Linear weight, a spectral-norm weight_orig):
    queued as a one-slab source and summed with everything else pending by flush_grad_sinks' one launch -- into a fresh
    ``p.grad`` or onto the existing one (the flat exchange buffer, gradient accumulation).  Handing it to autograd
    instead is free only for a parameter with ONE contribution per pass; HoloGAN's critic runs twice per D step and
    autograd then spends an ``add`` launch (and a host-side gap) per parameter, 13 per cycle.  True = taken (the caller
    returns None to autograd)."""
    if g is None or not _sinks.enabled or not isinstance(p, torch.nn.Parameter) or (p.numel() & 3):
        return False
    if p.grad is not None and (p.grad.data_ptr() & 15 or not p.grad.is_contiguous() or p.grad.dtype != torch.float32):
        return False
    g = _req(g)
    if g.data_ptr() & 15 or g.numel() != p.numel():
        return False
    _sinks.pending.setdefault(id(p), [p, []])[1].append((g, 1, p.numel()))
    return True


def _sink_zero(p, shape, device):
    """The gradient of ``p`` from this use is EXACTLY zero (a convolution bias in front of a normalisation over its own
    plane).  With sinks on, the parameter joins the flush as a job without sources -- written as zeros by the launch
    that sums everything else, or left alone when ``p.grad`` already holds contributions -- and None goes back to
    autograd; otherwise a zero tensor (one fill launch per such bias: HoloGAN had 11 per optimizer cycle)."""
    if (_sinks.enabled and isinstance(p, torch.nn.Parameter) and not (p.numel() & 3)
            and (p.grad is None or (p.grad.is_contiguous() and p.grad.dtype == torch.float32 and not p.grad.data_ptr() & 15))):
        _sinks.pending.setdefault(id(p), [p, []])
        return None
    return torch.zeros(shape, device=device, dtype=torch.float32)


def _sink_or_return(p, g):
    """``g`` for autograd, or None when the sink took it."""
    if g is None or _sink_grad(p, g):
        return None
    return g


def _sink_fail(what):
    raise RuntimeError("lightning_gan_zoo_amd: gradient sink cannot take this launch (%s)" % what)


def take_grad_sinks(params):
    """Hand the pending slab sources of ``params`` to the caller INSTEAD of reducing them into ``p.grad`` (the fused
    optimizers sum the slabs themselves: optim.Adam.step(sink_sources=...)).  Only parameters whose gradient of this pass
    consists of the pending sources alone -- ``p.grad`` is None -- and that have at least one source are taken;
    everything else stays for flush_grad_sinks.  -> {id(p): (p, [(slabs, nz, stride), ...])}"""
    out = {}
    max_src = lib.gz_reduce_multi_max_sources()
    for p in params:
        item = _sinks.pending.get(id(p))
        if item is None or p.grad is not None or not item[1] or len(item[1]) > max_src or (p.numel() & 3):
            continue
        if p.data_ptr() & 15 or any((s[0].data_ptr() & 15) or (s[2] & 3) for s in item[1]):
            continue
        out[id(p)] = (p, _sinks.pending.pop(id(p))[1])
    return out


def flush_grad_sinks(params=None):
    """Sum the queued weight-gradient slabs into ``p.grad`` -- of ``params`` (an iterable) or of everything pending --
    with as few gz_reduce_multi launches as the table size allows."""
    if not _sinks.pending:
        return
    if params is None:
        keys = list(_sinks.pending)
    else:
        keys = [id(p) for p in params if id(p) in _sinks.pending]
    if not keys:
        return
    # the gradients with the most slabs first: a 3-channel edge layer's 6144 values come as hundreds of slabs -- a long
    # chain of dependent loads for a handful of workgroups, which hides behind the bulk only if it starts with it
    keys.sort(key=lambda k: -sum(nz for (_, nz, _) in _sinks.pending[k][1]))
    max_jobs, max_src = lib.gz_reduce_multi_max_jobs(), lib.gz_reduce_multi_max_sources()
    nb = lib.gz_reduce_multi_table_bytes()
    st = _stream()
    table, njobs = (ctypes.c_char * nb)(), 0
    keep = []        # the slabs stay allocated until every launch that reads them has been ENQUEUED: a block freed
                     # earlier could come back as the next parameter's fresh gradient tensor inside this very loop
    for k in keys:
        w, srcs = _sinks.pending.pop(k)
        keep.append(srcs)
        fresh = w.grad is None
        target = torch.empty_like(w, memory_format=torch.contiguous_format) if fresh else w.grad
        if not target.is_contiguous() or target.dtype != torch.float32:
            _sink_fail("p.grad is not a contiguous float32 tensor")
        if njobs >= max_jobs:
            check(lib.gz_reduce_multi(table, st), "reduce_multi")
            table, njobs = (ctypes.c_char * nb)(), 0
        if not srcs:
            if not fresh:
                continue          # an exact-zero contribution to a gradient that already exists: nothing to do
            check(lib.gz_reduce_multi_add(table, _p(target), w.numel(), 0, None, 0, 0), "reduce_multi_add(zero)")
        for (slabs, nz, stride) in srcs[:max_src]:
            check(lib.gz_reduce_multi_add(table, _p(target), w.numel(), 0 if fresh else 1, _p(slabs), nz, stride),
                  "reduce_multi_add")
        njobs += 1
        if len(srcs) > max_src:       # more launches contributed than one job holds (not on the benchmarked paths):
            check(lib.gz_reduce_multi(table, st), "reduce_multi")          # the rest accumulates in launches of its own
            for lo in range(max_src, len(srcs), max_src):
                table = (ctypes.c_char * nb)()
                for (slabs, nz, stride) in srcs[lo:lo + max_src]:
                    check(lib.gz_reduce_multi_add(table, _p(target), w.numel(), 1, _p(slabs), nz, stride),
                          "reduce_multi_add")
                check(lib.gz_reduce_multi(table, st), "reduce_multi")
            table, njobs = (ctypes.c_char * nb)(), 0
        if fresh:
            w.grad = target
    if njobs:
        check(lib.gz_reduce_multi(table, st), "reduce_multi")
    del keep         # (stream-ordered allocator: later allocations on this stream come after the launches above)


def _act_bwd_raw(g, out, act, slope):
    dx = torch.empty_like(g)
    check(lib.gz_act_bwd(_p(g), _p(out), _p(dx), g.numel(), act, slope, _stream()), "act_bwd")
    return dx


def gemm(a, b, bias=None, trans_a=False, trans_b=False, act=ACT_NONE, slope=0.0):
    """c = act(op(a) @ op(b) + bias); raw launcher (no autograd)."""
    a, b = _req(a, "a"), _req(b, "b")
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    c = torch.empty((M, N), device=a.device, dtype=torch.float32)
    ws, nbytes = _scratch(lib.gz_gemm_workspace_bytes(M, N, K), a.device)
    _timed_detail(lambda: "gemm %s%s M%d N%d K%d" % ("T" if trans_a else "N", "T" if trans_b else "N", M, N, K),
                  2.0 * M * N * K,
                  lambda: check(lib.gz_gemm(_p(a), _p(b), _p(bias), _p(c), _p(ws), nbytes, M, N, K, a.shape[1],
                                            b.shape[1], N, int(trans_a), int(trans_b), act, slope, _stream()), "gemm"))
    return c




# (the 3-D packed-weight cache lives here with the 2-D one: set_pack_cache / invalidate manage both)
_pack3_cache = {}


def _packed3(w, kind):
    key = (w.data_ptr(), kind)
    cacheable = _pack_cache_enabled and isinstance(w, torch.nn.Parameter)
    if cacheable:
        hit = _pack3_cache.get(key)
        if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == tuple(w.shape):
            return hit[3]
    K, C, KS = w.shape[0], w.shape[1], w.shape[2]
    if kind == "f":
        wp = torch.empty(lib.gz_conv3d_pack_fwd_elems(K, C, KS), device=w.device, dtype=torch.float32)
        check(lib.gz_conv3d_pack_fwd(_p(w), _p(wp), K, C, KS, _stream()), "conv3d_pack_fwd")
    else:
        wp = torch.empty(lib.gz_conv3d_pack_dgrad_elems(K, C, KS, 2), device=w.device, dtype=torch.float32)
        check(lib.gz_conv3d_pack_dgrad(_p(w), _p(wp), K, C, KS, 2, 1, _stream()), "conv3d_pack_dgrad")
    if cacheable:
        _pack3_cache[key] = (weakref.ref(w), w._version, tuple(w.shape), wp)
    return wp


__all__ = [n for n in list(globals()) if not n.startswith("__")]     # the flat namespace of the package (private helpers included)
