"""torch.autograd.Function wrappers over the C-ABI HIP kernels (include/gz_ops.h).

PyTorch is used for device memory, streams and the autograd tape only; every
arithmetic step of the hot path is a call into libgz_hip.so.  The convolution
family is closed under differentiation:

    F(x, w)  = conv2d            dF/dx^T g  = Dg(g, w)   dF/dw^T g  = Wg(x, g)
    Dg(g, w) = conv_transpose2d  dDg/dg^T v = F(v, w)    dDg/dw^T v = Wg(v, g)
    Wg(x, g) = weight gradient   dWg/dx^T v = Dg(g, v)   dWg/dg^T v = F(x, v)

so `torch.autograd.grad(..., create_graph=True)` (the WGAN-GP gradient penalty,
reference core/utils/utils.py:48-54) works through these ops to any order.
"""
import ctypes
import os
import weakref
from collections import namedtuple

import torch

from ._lib import check, lib

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
          9: "256x(2x64)"}


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


# ---------------------------------------------------------------------------
# activation backward as a differentiable op (linear in g; the mask is piecewise constant)
# ---------------------------------------------------------------------------
class _ActBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, out, act, slope):
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(g, out)
        return _act_bwd_raw(_req(g), out, act, slope)

    @staticmethod
    def backward(ctx, v):
        g, out = ctx.saved_tensors
        gg = _ActBwd.apply(v, out, ctx.act, ctx.slope) if ctx.needs_input_grad[0] else None
        go = None
        if ctx.needs_input_grad[1] and ctx.act == ACT_TANH:
            # d/d(out) of g*(1-out^2); only tanh has a non-constant mask
            if out.numel() % 4 == 0:
                v, g, o = _req(v), _req(g), _req(out)
                go = torch.empty_like(o)
                check(lib.gz_tanh_bwd2(_p(v), _p(g), _p(o), _p(go), o.numel(), _stream()), "tanh_bwd2")
            else:
                go = v * g * (-2.0 * out)
        return gg, go, None, None


# ---------------------------------------------------------------------------
# F / Dg / Wg
# ---------------------------------------------------------------------------
# In a first-order backward the weight gradient and the input gradient of a layer are independent and could share
# the GPU from two streams (one kernel's drain covered by the other).  Measured and left OFF (threshold 0): two large
# MFMA-bound kernels running together lose more to cache / LDS contention than their drains cost (dc_gan +8 %,
# hologan +9 % step time), and for the small R1 layers the two extra stream waits per layer make the already
# launch-heavy step host-bound (15.4 -> 19.0 ms).  GZ_WG_SIDE_STREAM_FLOPS=<flops> enables it for layers below
# that size.  When enabled, the main stream waits for the side stream before anything else is enqueued, so every
# consumer sees both results and no record_stream is needed.
_side_streams = {}
_WG_SIDE_FLOPS = float(os.environ.get("GZ_WG_SIDE_STREAM_FLOPS", "0"))


def _conv_flops(x, gy, w):
    return 2.0 * gy.shape[0] * gy.shape[2] * gy.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]


def _pair_wgrad_dgrad(wgrad, dgrad, flops):
    """Run wgrad() on the side stream and dgrad() on the current one; returns (dx, dw)."""
    if flops > _WG_SIDE_FLOPS or torch.is_grad_enabled():     # double backward: one stream (autograd tracks nothing across)
        dx = dgrad()
        return dx, wgrad()
    main = torch.cuda.current_stream()
    key = (main.device_index, main.cuda_stream)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=main.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = wgrad()
    dx = dgrad()
    main.wait_stream(side)
    return dx, dw


class _ConvF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, geom, act, slope, want_stats=False):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.geom, ctx.act, ctx.slope = geom, act, slope
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        if want_stats:          # BatchNorm follows: no bias, no activation; second output = partial statistics
            y, stats = _conv_fwd_stats_raw(x, w, geom)
            ctx.save_for_backward(x, w, None)
            if stats is None:
                stats = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)      # else every backward launches a zero fill for the statistics' "gradient"
            return y, stats
        y = _conv_fwd_raw(x, w, bias, geom, act, slope)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy, *_stats_grad):
        if gy is None:
            return (None,) * 7
        x, w, y = ctx.saved_tensors
        geom = ctx.geom
        if (ctx.act != ACT_NONE and not torch.is_grad_enabled() and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                and (not ctx.has_bias or not ctx.needs_input_grad[2] or ctx.bias_ref is not None)):
            # weights (and bias) only: activation backward, weight gradient and bias gradient in one launch where it exists
            b = ctx.bias_ref if ctx.has_bias and ctx.needs_input_grad[2] else None
            if _sink_conv_wgrad_act(w, b, x, gy, y, geom, ctx.act, ctx.slope):
                return (None,) * 7
        if (ctx.act in (ACT_RELU, ACT_LRELU) and not torch.is_grad_enabled() and ctx.needs_input_grad[0]
                and not ctx.needs_input_grad[1] and not (ctx.has_bias and ctx.needs_input_grad[2])):
            # input only (a generator step through the frozen critic's first layer): the mask is formed on load
            dx = _conv_dgrad_act_raw(gy, y, ctx.act, ctx.slope, w, geom, tuple(x.shape[2:]))
            if dx is not None:
                return (dx,) + (None,) * 6
        if ctx.act != ACT_NONE:
            gy = _ActBwd.apply(gy, y, ctx.act, ctx.slope)
        if not torch.is_grad_enabled() and _WG_SIDE_FLOPS <= 0:
            return _ConvF._first_order(ctx, gy, x, w, geom)
        dx = dw = db = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = _pair_wgrad_dgrad(lambda: _ConvWg.apply(x, gy, geom),
                                       lambda: _ConvDg.apply(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0),
                                       _conv_flops(x, gy, w))
        elif ctx.needs_input_grad[0]:
            dx = _ConvDg.apply(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        elif ctx.needs_input_grad[1]:
            dw = _ConvWg.apply(x, gy, geom)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _ChannelSum.apply(gy)
        return dx, dw, db, None, None, None, None

    @staticmethod
    def _first_order(ctx, gy, x, w, geom):
        """No graph is being recorded: raw launches, weight and bias gradient from one kernel where it can."""
        gy = _req(gy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad_raw(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                dw, db = _conv_wgrad_raw(x, gy, geom, with_bias=True)
                dw = _sink_or_return(w, dw)       # (complete gradient: one-slab source, no AccumulateGrad `add_`)
            elif not _sink_conv_wgrad(w, x, gy, geom):
                dw = _conv_wgrad_raw(x, gy, geom)
        elif want_b:
            db = _channel_sum_raw(gy)
        if want_b and ctx.bias_ref is not None:
            db = _sink_or_return(ctx.bias_ref, db)
        return dx, dw, db, None, None, None, None


class _ConvDg(torch.autograd.Function):
    """x = act(conv_transpose2d(g, w) + bias): ConvTranspose2d forward and Conv2d input gradient."""

    @staticmethod
    def forward(ctx, g, w, bias, geom, hw, act, slope, want_stats=False, bias_cancels=False):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.geom, ctx.act, ctx.slope = geom, act, slope
        ctx.has_bias = bias is not None
        ctx.bias_cancels = bias_cancels      # the caller normalises the output per (sample, channel): d/d bias == 0
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        if want_stats:
            x, stats = _conv_dgrad_stats_raw(g, w, geom, hw)
            ctx.save_for_backward(g, w, None)
            if stats is None:
                stats = torch.empty(0, device=g.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)
            return x, stats
        x = _conv_dgrad_raw(g, w, bias, geom, hw, act, slope)
        ctx.save_for_backward(g, w, x if act != ACT_NONE else None)
        return x

    @staticmethod
    def backward(ctx, v, *_stats_grad):
        if v is None:
            return (None,) * 9
        g, w, x = ctx.saved_tensors
        geom = ctx.geom
        if ctx.act != ACT_NONE:
            v = _ActBwd.apply(v, x, ctx.act, ctx.slope)
        if not torch.is_grad_enabled() and _WG_SIDE_FLOPS <= 0:      # no graph is being recorded: raw launches
            v = _req(v)
            dg = _conv_fwd_raw(v, w, None, geom, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
            dw = None
            if ctx.needs_input_grad[1] and not _sink_conv_wgrad(w, v, g, geom):
                dw = _conv_wgrad_raw(v, g, geom)
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if ctx.bias_cancels:
                    db = _sink_zero(ctx.bias_ref, (v.shape[1],), v.device)
                else:
                    db = _channel_sum_raw(v)
                    if ctx.bias_ref is not None:
                        db = _sink_or_return(ctx.bias_ref, db)
            return dg, dw, db, None, None, None, None, None, None
        dg = dw = db = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dg, dw = _pair_wgrad_dgrad(lambda: _ConvWg.apply(v, g, geom),
                                       lambda: _ConvF.apply(v, w, None, geom, ACT_NONE, 0.0), _conv_flops(v, g, w))
        elif ctx.needs_input_grad[0]:
            dg = _ConvF.apply(v, w, None, geom, ACT_NONE, 0.0)
        elif ctx.needs_input_grad[1]:
            dw = _ConvWg.apply(v, g, geom)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _ChannelSum.apply(v)
        return dg, dw, db, None, None, None, None, None, None


class _ConvWg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, geom):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.geom = geom
        ctx.save_for_backward(x, g)
        return _conv_wgrad_raw(x, g, geom)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        geom = ctx.geom
        v = _req(v)
        dx = dg = None
        if ctx.needs_input_grad[0]:
            dx = _ConvDg.apply(g, v, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        if ctx.needs_input_grad[1]:
            dg = _ConvF.apply(x, v, None, geom, ACT_NONE, 0.0)
        return dx, dg, None


def conv2d(x, w, bias=None, geom=K4S2P1, act=ACT_NONE, slope=0.0):
    return _ConvF.apply(x, w, bias, geom, act, slope)


def conv_transpose2d(x, w, bias=None, geom=K4S2P1, act=ACT_NONE, slope=0.0, bias_cancels=False):
    """w is the ConvTranspose2d weight [Cin, Cout, KH, KW]; output size (H-1)*S - 2P + KH.
    bias_cancels: the caller feeds the output straight into a normalisation over each (sample, channel) plane (AdaIN,
    InstanceNorm), which removes a per-channel constant -- the bias gradient is exactly zero and is returned as such
    (a first-order backward skips the two reduction launches; the reference's value is rounding noise)."""
    H, W = x.shape[2:]
    oh = (H - 1) * geom.stride - 2 * geom.pad + geom.kh
    ow = (W - 1) * geom.stride - 2 * geom.pad + geom.kw
    return _ConvDg.apply(x, w, bias, geom, (oh, ow), act, slope, False, bias_cancels)


def conv2d_with_stats(x, w, geom=K4S2P1):
    """(conv2d(x, w), stats) for a convolution that feeds a training-mode BatchNorm: the launch also produces the
    per-tile (sum, sum of squares) of its output; pass ``stats`` to batch_norm_act.  stats is an empty tensor when
    the launch could not carry them (batch_norm_act then reads the feature map itself)."""
    return _ConvF.apply(x, w, None, geom, ACT_NONE, 0.0, True)


def conv_transpose2d_with_stats(x, w, geom=K4S2P1):
    H, W = x.shape[2:]
    oh = (H - 1) * geom.stride - 2 * geom.pad + geom.kh
    ow = (W - 1) * geom.stride - 2 * geom.pad + geom.kw
    return _ConvDg.apply(x, w, None, geom, (oh, ow), ACT_NONE, 0.0, True)


# ---------------------------------------------------------------------------
# dense layers on the same GEMM core: the generator's 1x1 -> 4x4 ConvTranspose2d
# (reference standard_networks.py:60) and nn.Linear
# ---------------------------------------------------------------------------
class _MatMul(torch.autograd.Function):
    """c = a @ b for row-major a [M,K], b [K,N].  ``param``: the Parameter ``b`` is a view of (the DCGAN generator's
    first layer): a first-order backward hands its gradient to that parameter's sink instead of autograd."""

    @staticmethod
    def forward(ctx, a, b, param=None):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        ctx.param = param
        return gemm(a, b)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _MatMulNT.apply(gc, b)            # gc @ b^T
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                gb = gemm(a, gc, trans_a=True)
                if _sink_grad(ctx.param, gb.view_as(ctx.param)):
                    gb = None
            else:
                gb = _MatMulTN.apply(a, gc)        # a^T @ gc
        return ga, gb, None


class _MatMulNT(torch.autograd.Function):
    """c = a @ b^T for a [M,K], b [N,K]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        return gemm(a, b, trans_b=True)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = _MatMul.apply(gc, b) if ctx.needs_input_grad[0] else None       # gc @ b
        gb = _MatMulTN.apply(gc, a) if ctx.needs_input_grad[1] else None     # gc^T @ a
        return ga, gb


class _MatMulTN(torch.autograd.Function):
    """c = a^T @ b for a [K,M], b [K,N]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        return gemm(a, b, trans_a=True)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = _MatMulNT.apply(b, gc) if ctx.needs_input_grad[0] else None     # b @ gc^T
        gb = _MatMul.apply(a, gc) if ctx.needs_input_grad[1] else None       # a @ gc
        return ga, gb


def matmul(a, b, param=None):
    return _MatMul.apply(a, b, param)


def matmul_nt(a, b):
    return _MatMulNT.apply(a, b)


def linear(x, weight, bias=None):
    y = _MatMulNT.apply(x, weight)
    return y if bias is None else y + bias


# ---------------------------------------------------------------------------
# last discriminator layer: Conv2d(C, 1, k4, s2, p0) on a 4x4 map == per-sample dot
# ---------------------------------------------------------------------------
def _rowdot_raw(a, b, bcast):
    R, L = a.shape
    y = torch.empty(R, device=a.device, dtype=torch.float32)
    check(lib.gz_rowdot(_p(a), _p(b), _p(y), R, L, int(bcast), _stream()), "rowdot")
    return y


def _rowscale_raw(x, s, R, L, bcast):
    out = torch.empty((R, L), device=s.device, dtype=torch.float32)
    check(lib.gz_rowscale(_p(x), _p(s), None, None, _p(out), R, L, int(bcast), 0, _stream()), "rowscale")
    return out


def _coldot_raw(g, x):
    R, L = x.shape
    out = torch.empty(L, device=x.device, dtype=torch.float32)
    nbytes = lib.gz_coldot_workspace_bytes(R, L)
    ws = _ws(max(nbytes // 4, 1), x.device)
    check(lib.gz_coldot(_p(g), _p(x), _p(out), _p(ws), nbytes, R, L, _stream()), "coldot")
    return out


def _sink_coldot(param, g, x):
    """``param``'s gradient sum_r g[r] * x[r, :] into its sink as UNREDUCED row slices (gz_coldot_partial): the slab sum
    that would follow is done by the launch that sums everything else.  False = not taken."""
    if not _sinks.enabled or not isinstance(param, torch.nn.Parameter) or (param.numel() & 3):
        return False
    if param.grad is not None and (param.grad.data_ptr() & 15 or not param.grad.is_contiguous()
                                   or param.grad.dtype != torch.float32):
        return False
    R, L = x.shape
    if L != param.numel():
        return False
    nbytes = lib.gz_coldot_workspace_bytes(R, L)
    ws = _ws(max(nbytes // 4, 1), x.device)
    out = torch.empty(L, device=x.device, dtype=torch.float32)
    nz = ctypes.c_int(0)
    check(lib.gz_coldot_partial(_p(g), _p(x), _p(out), _p(ws), nbytes, R, L, ctypes.byref(nz), _stream()), "coldot_partial")
    src = (ws, nz.value, L) if nz.value > 1 else (out, 1, L)
    if src[0].data_ptr() & 15:
        _sink_fail("coldot slices")
    _sinks.pending.setdefault(id(param), [param, []])[1].append(src)
    return True


class _DotF(torch.autograd.Function):
    """y[r] = <x[r,:], w>"""

    @staticmethod
    def forward(ctx, x, w, param=None):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.save_for_backward(x, w)
        ctx.param = param          # the Parameter ``w`` is a view of: a first-order backward feeds its sink
        return _rowdot_raw(x, w, True)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = _DotDg.apply(g, w, ctx.param) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                if not _sink_coldot(ctx.param, _req(g), x):
                    dw = _coldot_raw(_req(g), x)
                    if _sink_grad(ctx.param, dw.view_as(ctx.param)):
                        dw = None
            else:
                dw = _DotWg.apply(x, g)
        return dx, dw, None


class _DotDg(torch.autograd.Function):
    """x[r,:] = g[r] * w"""

    @staticmethod
    def forward(ctx, g, w, param=None):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.save_for_backward(g, w)
        ctx.param = param
        return _rowscale_raw(w, g, g.numel(), w.numel(), True)

    @staticmethod
    def backward(ctx, v):
        g, w = ctx.saved_tensors
        dg = _DotF.apply(v, w, ctx.param) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                if not _sink_coldot(ctx.param, _req(g), _req(v)):     # == _DotWg(v, g)
                    dw = _coldot_raw(_req(g), _req(v))
                    if _sink_grad(ctx.param, dw.view_as(ctx.param)):
                        dw = None
            else:
                dw = _DotWg.apply(v, g)
        return dg, dw, None


class _DotWg(torch.autograd.Function):
    """dw = sum_r g[r] * x[r,:]"""

    @staticmethod
    def forward(ctx, x, g):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.save_for_backward(x, g)
        return _coldot_raw(g, x)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        dx = _DotDg.apply(g, v) if ctx.needs_input_grad[0] else None
        dg = _DotF.apply(x, v) if ctx.needs_input_grad[1] else None
        return dx, dg


def full_dot_conv(x, w):
    """Conv2d whose kernel covers the whole (unpadded) input: [N,C,H,W] x [1,C,H,W] -> [N,1,1,1]."""
    n = x.shape[0]
    y = _DotF.apply(x.reshape(n, -1), w.reshape(-1), w if isinstance(w, torch.nn.Parameter) else None)
    return y.reshape(n, 1, 1, 1)


# ---------------------------------------------------------------------------
# normalisation + activation
# ---------------------------------------------------------------------------
def _norm_ws(x, N, C):
    return _ws(max(lib.gz_norm_workspace_bytes(N, C) // 4, 1), x.device)


def _sink_small_pair(a, b):
    """(a.grad, b.grad, accumulate) for two small parameter gradients one kernel writes / adds in place (a norm's
    gamma and beta), or None when the sinks are off or the pair cannot be taken TOGETHER (nothing is touched then).
    Fresh gradients are created on the spot (accumulate 0); existing ones are accumulated into (1)."""
    if not _sinks.enabled:
        return None
    for p in (a, b):
        if not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
            return None
    if (a.grad is None) != (b.grad is None):
        return None
    if a.grad is None:
        a.grad = torch.empty_like(a, memory_format=torch.contiguous_format)
        b.grad = torch.empty_like(b, memory_format=torch.contiguous_format)
        return a.grad, b.grad, 0
    for p in (a, b):
        if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
            return None
    return a.grad, b.grad, 1


class _BatchNormAct(torch.autograd.Function):
    """act(BatchNorm(x)); training mode updates the running buffers in place exactly like
    nn.BatchNorm2d (momentum, unbiased running var, num_batches_tracked += 1).  ``groups`` > 1: the batch is that many
    independent statistics groups stacked along n (gz_batchnorm_finalize_g) -- one pass over [real; fake] is the
    reference's two discriminator calls."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act, slope, stats=None,
                groups=1):
        x = _req(x, "x")
        N, C = x.shape[:2]
        inner = x.numel() // (N * C)
        if groups > 1 and (not training or N % groups):
            raise RuntimeError("statistics groups need training mode and a batch that is a multiple of the group count")
        coef = torch.empty(4 * groups * C, device=x.device, dtype=torch.float32)
        st = _stream()
        if training and not _NORM_UNFUSED:
            # finalize + apply in one launch (the statistics come from the convolution's epilogue, or from a row-sum pass)
            out = torch.empty_like(x)
            fused = stats is not None and stats.numel()
            ws = None if fused else _norm_ws(x, N, C)
            check(lib.gz_batchnorm_act_fwd_fused(_p(x), _p(stats) if fused else None, stats.shape[0] if fused else 0,
                                                 (N // groups) * inner if fused else 0, _p(gamma), _p(beta), _p(coef),
                                                 _p(running_mean), _p(running_var), _p(nbt), _p(ws), _p(out), N, C, inner,
                                                 eps, momentum, groups, act, slope, st), "batchnorm_act_fwd_fused")
            ctx.save_for_backward(x, coef, gamma, beta)
            ctx.cfg = (N, C, inner, act, slope, training, groups)
            return out
        if training and stats is not None and stats.numel():
            # partial sums written by the producing convolution's epilogue (conv2d_with_stats)
            check(lib.gz_batchnorm_finalize_g(_p(stats), stats.shape[0], (N // groups) * inner, _p(gamma), _p(beta),
                                              _p(coef), _p(running_mean), _p(running_var), _p(nbt), C, eps, momentum,
                                              groups, st), "batchnorm_finalize")
        elif training:
            ws = _norm_ws(x, N, C)
            check(lib.gz_batchnorm_stats_g(_p(x), _p(gamma), _p(beta), _p(coef), _p(running_mean), _p(running_var),
                                           _p(nbt), _p(ws), N, C, inner, eps, momentum, groups, st), "batchnorm_stats")
        else:
            check(lib.gz_batchnorm_eval_coef(_p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(coef), C,
                                             eps, st), "batchnorm_eval_coef")
        out = torch.empty_like(x)
        check(lib.gz_norm_act_fwd_g(_p(x), _p(coef), _p(out), N, C, inner, 1, groups, act, slope, st), "norm_act_fwd")
        ctx.save_for_backward(x, coef, gamma, beta)
        ctx.cfg = (N, C, inner, act, slope, training, groups)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef, gamma, beta = ctx.saved_tensors
        N, C, inner, act, slope, training, groups = ctx.cfg
        gout = _req(gout)
        nones = (None,) * 10
        if not training:
            # eval mode: y = act(x * scale[c] + shift[c]) with constants from the running statistics -- a plain
            # per-channel affine, assembled from the row helpers (not on the training hot path)
            st = _stream()
            gp = gout
            if act != ACT_NONE:
                out = torch.empty_like(x)
                check(lib.gz_norm_act_fwd(_p(x), _p(coef), _p(out), N, C, inner, 1, act, slope, st), "norm_act_fwd")
                gp = _act_bwd_raw(gout, out, act, slope)
            scale, mean, rstd = coef[:C], coef[2 * C:3 * C], coef[3 * C:]
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _rowscale_raw(gp.view(N * C, inner), scale.repeat(N).contiguous(), N * C, inner, False).view_as(x)
            dbeta = _channel_sum_raw(gp)
            sgx = _rowdot_raw(gp.view(N * C, inner), x.view(N * C, inner), False).view(N, C).sum(0)
            dgamma = rstd * (sgx - mean * dbeta)
            return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None) + nones
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        # gradient sinks: the finalize kernel writes (or adds to) gamma.grad / beta.grad itself -- a discriminator applied
        # twice per step otherwise pays a framework `add_` launch per affine parameter
        sunk = _sink_small_pair(gamma, beta) if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
        if sunk is not None:
            dgamma, dbeta, accumulate = sunk
        else:
            dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
            dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
            accumulate = 0
        kbuf = torch.empty(2 * groups * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        check(lib.gz_batchnorm_act_bwd_g(_p(gout), _p(x), _p(coef), _p(dx), _p(dgamma), _p(dbeta), _p(ws), _p(kbuf), N,
                                         C, inner, act, slope, groups, accumulate, _stream()), "norm_act_bwd")
        if sunk is not None:
            return (dx, None, None) + nones
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None) + nones


def batch_norm_act(x, gamma, beta, running_mean, running_var, nbt, training, momentum=0.1, eps=1e-5,
                   act=ACT_NONE, slope=0.0, stats=None, groups=1):
    if groups > 1 and stats is not None and stats.numel():
        # the convolution's partial rows must not straddle two groups: rows per group integral, pixels per row too
        rows, M = stats.shape[0], x.shape[0] * (x.numel() // (x.shape[0] * x.shape[1]))
        if rows % groups or M % rows or (M // groups) % (M // rows):
            stats = None
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act, slope,
                               stats, groups)


class _RowNormAct(torch.autograd.Function):
    """act(InstanceNorm(x)) with per-channel affine (nn.InstanceNorm2d(affine=True), biased variance,
    always instance statistics).  Its backward is itself differentiable (_RowNormActBwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, slope):
        x = _req(x, "x")
        N, C = x.shape[:2]
        inner = x.numel() // (N * C)
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty_like(x)
        check(lib.gz_rownorm_act_fwd(_p(x), _p(gamma), _p(beta), _p(coef), _p(out), N, C, inner, eps, 0, 0, act, slope,
                                     _stream()), "rownorm_act_fwd")
        ctx.save_for_backward(x, gamma, coef, beta)
        ctx.cfg = (N, C, inner, act, slope)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, gamma, coef, beta = ctx.saved_tensors
        if (not torch.is_grad_enabled() and gamma is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]):
            # first-order backward under gradient sinks: the affine gradients are written / added in place
            sunk = _sink_small_pair(gamma, beta)
            if sunk is not None:
                N, C, inner, act, slope = ctx.cfg
                gout = _req(gout)
                dx = torch.empty_like(x)
                kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
                ws = _norm_ws(x, N, C)
                check(lib.gz_rownorm_act_bwd_acc(_p(gout), _p(x), _p(coef), _p(dx), _p(sunk[0]), _p(sunk[1]), _p(ws),
                                                 _p(kbuf), N, C, inner, act, slope, sunk[2], _stream()),
                      "rownorm_act_bwd")
                return (dx if ctx.needs_input_grad[0] else None), None, None, None, None, None
        dx, dgamma, dbeta = _RowNormActBwd.apply(gout, x, gamma, coef, ctx.cfg)
        return (dx if ctx.needs_input_grad[0] else None,
                dgamma if (gamma is not None and ctx.needs_input_grad[1]) else None,
                dbeta if (gamma is not None and ctx.needs_input_grad[2]) else None, None, None, None)


class _RowNormActBwd(torch.autograd.Function):
    """(dx, dgamma, dbeta) of _RowNormAct as a function of (gout, x, gamma); `coef` carries the
    statistics of x and is not an independent variable (the double backward formula accounts
    for the dependence of mean / rstd on x)."""

    @staticmethod
    def forward(ctx, gout, x, gamma, coef, cfg):
        N, C, inner, act, slope = cfg
        gout = _req(gout)
        dx = torch.empty_like(x)
        # affine=False (HoloGAN's discriminator): no per-channel reduction over the row sums at all
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32) if gamma is not None else None
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32) if gamma is not None else None
        kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        check(lib.gz_norm_act_bwd(_p(gout), _p(x), _p(coef), _p(dx), _p(dgamma), _p(dbeta), _p(ws), _p(kbuf), N, C,
                                  inner, 0, 0, 0, act, slope, _stream()), "norm_act_bwd")
        ctx.save_for_backward(gout, x, gamma, coef)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return dx, dgamma, dbeta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v, v_dgamma, v_dbeta):
        if v_dgamma is not None or v_dbeta is not None:
            raise RuntimeError("second-order terms through dgamma/dbeta are outside the hot path")
        gout, x, gamma, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        if v is None:
            return None, None, None, None, None
        v = _req(v)
        gg = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        need_gamma = gamma is not None and ctx.needs_input_grad[2]
        ggamma = torch.empty(C, device=x.device, dtype=torch.float32) if need_gamma else None
        ws = _norm_ws(x, N, C)
        check(lib.gz_rownorm_act_bwd2(_p(gout), _p(v), _p(x), _p(coef), _p(gg), _p(gx), _p(ggamma), _p(ws), N, C,
                                      inner, act, slope, _stream()), "rownorm_act_bwd2")
        if ggamma is not None:
            ggamma = _sink_or_return(gamma, ggamma)       # (second-order contribution: joins gamma's sink, no add launch)
        return gg, gx, ggamma, None, None


def instance_norm_act(x, gamma, beta, eps=1e-5, act=ACT_NONE, slope=0.0):
    return _RowNormAct.apply(x, gamma, beta, eps, act, slope)


# ---------------------------------------------------------------------------
# gradient-penalty tail (reference core/utils/utils.py:41-42, 55-57)
# ---------------------------------------------------------------------------
class _Lerp(torch.autograd.Function):
    """out[n] = alpha[n]*a[n] + (1-alpha[n])*b[n] for a, b [N, L], alpha [N]."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b, alpha = _req(a, "a"), _req(b, "b"), _req(alpha, "alpha")
        R, L = a.shape[0], a.numel() // a.shape[0]
        out = torch.empty_like(a)
        check(lib.gz_rowscale(_p(a), _p(alpha), _p(b), None, _p(out), R, L, 0, 1, _stream()), "rowscale(lerp)")
        ctx.save_for_backward(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = row_scale(g, alpha)
        if ctx.needs_input_grad[1]:
            gb = row_scale(g, 1.0 - alpha)
        return ga, gb, None


def lerp_rows(a, b, alpha):
    return _Lerp.apply(a, b, alpha.reshape(-1))


class _RowScale(torch.autograd.Function):
    """out[n] = s[n] * x[n]"""

    @staticmethod
    def forward(ctx, x, s):
        x, s = _req(x, "x"), _req(s, "s")
        R, L = x.shape[0], x.numel() // x.shape[0]
        out = torch.empty_like(x)
        check(lib.gz_rowscale(_p(x), _p(s), None, None, _p(out), R, L, 0, 0, _stream()), "rowscale")
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = _RowScale.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _RowDot.apply(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class _RowDot(torch.autograd.Function):
    """y[n] = <a[n], b[n]>"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        R = a.shape[0]
        return _rowdot_raw(a.reshape(R, -1), b.reshape(R, -1), False)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _RowScale.apply(b, g) if ctx.needs_input_grad[0] else None
        gb = _RowScale.apply(a, g) if ctx.needs_input_grad[1] else None
        return ga, gb


def row_scale(x, s):
    return _RowScale.apply(x, s.reshape(-1))


def row_dot(a, b):
    return _RowDot.apply(a, b)


def row_sumsq(x):
    """sum of squares per sample, [N, ...] -> [N]"""
    return _RowDot.apply(x, x)


class _GPPenalty(torch.autograd.Function):
    """mean((sqrt(sumsq) - 1)^2): the tail of the gradient penalty in one launch each way (gz_gp_penalty)."""

    @staticmethod
    def forward(ctx, sumsq):
        sumsq = _req(sumsq, "sumsq")
        out = torch.empty((), device=sumsq.device, dtype=torch.float32)
        check(lib.gz_gp_penalty(_p(sumsq), _p(out), sumsq.numel(), _stream()), "gp_penalty")
        ctx.save_for_backward(sumsq)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (sumsq,) = ctx.saved_tensors
        ds = torch.empty_like(sumsq)
        check(lib.gz_gp_penalty_bwd(_p(sumsq), _p(_req(g)), _p(ds), sumsq.numel(), _stream()), "gp_penalty_bwd")
        return ds


def gp_penalty(sumsq):
    """``torch.mean((norm - 1) ** 2)`` with ``norm = sqrt(sumsq)`` per sample (reference core/utils/utils.py:55-57);
    the subgradient at an exactly-zero gradient is 0, as torch.norm's."""
    return _GPPenalty.apply(sumsq.reshape(-1))


_ones = {}


def ones_like_const(t):
    """A read-only tensor of ones shaped like ``t`` (``grad_outputs=torch.ones_like(scores)``, reference utils.py:51):
    cached per shape and device instead of a fill launch per step.  Never written by anyone."""
    key = (tuple(t.shape), t.device, t.dtype)
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = torch.ones(t.shape, device=t.device, dtype=t.dtype)
    return o


@torch.no_grad()
def clamp_(t, lo, hi):
    """in-place clamp of a parameter tensor (WGAN weight clipping, lightning_module.py:160-162)."""
    if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
        raise RuntimeError("clamp_: expected a contiguous float32 GPU tensor")
    check(lib.gz_clamp_(_p(t), t.numel(), float(lo), float(hi), _stream()), "clamp_")
    invalidate(t)
    return t


# ---------------------------------------------------------------------------
# HoloGAN: 3-D transposed convolution family, AdaIN, dense layers with fused epilogue,
# rigid-body resampling, spectral normalisation
# ---------------------------------------------------------------------------
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


def _conv3d_fwd_raw(x, w, bias, act, slope):
    N, C, D, H, W = x.shape
    K, KS = w.shape[0], w.shape[2]
    OD, OH, OW = D // 2, H // 2, W // 2
    y = torch.empty((N, K, OD, OH, OW), device=x.device, dtype=torch.float32)
    wp = _packed3(w, "f")
    ws, nbytes = _scratch(lib.gz_conv3d_fwd_workspace_bytes(N, C, K, OD, OH, OW, KS), x.device)
    _timed_detail(lambda: "igemm3d<F> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * KS ** 3,
                  lambda: check(lib.gz_conv3d_fwd(_p(x), _p(wp), _p(bias), _p(y), _p(ws), nbytes, N, C, D, H, W, K, OD,
                                                  OH, OW, KS, 2, 1, act, slope, _stream()), "conv3d_fwd"))
    return y


def _conv3d_dgrad_raw(g, w, bias, act, slope):
    N, K, OD, OH, OW = g.shape
    C, KS = w.shape[1], w.shape[2]
    D, H, W = 2 * OD, 2 * OH, 2 * OW
    x = torch.empty((N, C, D, H, W), device=g.device, dtype=torch.float32)
    wp = _packed3(w, "d")
    ws, nbytes = _scratch(lib.gz_conv3d_dgrad_workspace_bytes(N, C, K, OD, OH, OW, KS), g.device)
    _timed_detail(lambda: "igemm3d<Dg> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * KS ** 3,
                  lambda: check(lib.gz_conv3d_dgrad(_p(g), _p(wp), _p(bias), _p(x), _p(ws), nbytes, N, C, D, H, W, K,
                                                    OD, OH, OW, KS, 2, 1, act, slope, _stream()), "conv3d_dgrad"))
    return x


def _conv3d_wgrad_raw(x, g, ks):
    N, C, D, H, W = x.shape
    _, K, OD, OH, OW = g.shape
    dw = torch.empty((K, C, ks, ks, ks), device=x.device, dtype=torch.float32)
    nbytes = lib.gz_conv3d_wgrad_workspace_bytes(N, C, K, OD, OH, OW, ks)
    ws = _ws(max(nbytes // 4, 1), x.device)
    _timed_detail(lambda: "igemm3d<Wg> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * ks ** 3,
                  lambda: check(lib.gz_conv3d_wgrad(_p(x), _p(g), _p(dw), _p(ws), nbytes, N, C, D, H, W, K, OD, OH, OW,
                                                    ks, 2, 1, _stream()), "conv3d_wgrad"))
    return dw


class _Conv3DDg(torch.autograd.Function):
    """x = conv_transpose3d(g, w) + bias   (k3, s2, p1, output_padding 1)"""

    @staticmethod
    def forward(ctx, g, w, bias, bias_cancels=False):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.save_for_backward(g, w)
        ctx.w_ref = w
        ctx.has_bias = bias is not None
        ctx.bias_cancels = bias_cancels
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        return _conv3d_dgrad_raw(g, w, bias, ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, v):
        g, w = ctx.saved_tensors
        v = _req(v)
        if torch.is_grad_enabled():
            dg = _Conv3DF.apply(v, w) if ctx.needs_input_grad[0] else None
            dw = _Conv3DWg.apply(v, g, w.shape[2]) if ctx.needs_input_grad[1] else None
        else:                                # no graph is being recorded: raw launches, the weight gradient into its sink
            dg = _conv3d_fwd_raw(v, w, None, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
            dw = _sink_or_return(ctx.w_ref, _conv3d_wgrad_raw(v, g, w.shape[2])) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if ctx.bias_cancels and not torch.is_grad_enabled():
                db = _sink_zero(ctx.bias_ref, (v.shape[1],), v.device)
            else:
                db = _ChannelSum.apply(v) if torch.is_grad_enabled() else _channel_sum_raw(v)
        return dg, dw, db, None


class _Conv3DF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.save_for_backward(x, w)
        return _conv3d_fwd_raw(x, w, None, ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _req(gy)
        dx = _Conv3DDg.apply(gy, w, None, False) if ctx.needs_input_grad[0] else None
        dw = _Conv3DWg.apply(x, gy, w.shape[2]) if ctx.needs_input_grad[1] else None
        return dx, dw


class _Conv3DWg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, ks):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.save_for_backward(x, g)
        return _conv3d_wgrad_raw(x, g, ks)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        v = _req(v)
        dx = _Conv3DDg.apply(g, v, None, False) if ctx.needs_input_grad[0] else None
        dg = _Conv3DF.apply(x, v) if ctx.needs_input_grad[1] else None
        return dx, dg, None


def conv_transpose3d(x, w, bias=None, bias_cancels=False):
    """nn.ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1); w [Cin, Cout, 3, 3, 3]; bias_cancels as in
    conv_transpose2d."""
    return _Conv3DDg.apply(x, w, bias, bias_cancels)


class _AdaINAct(torch.autograd.Function):
    """act(scale[n,c] * (x - mean) * rsqrt(var_unbiased + eps) + bias[n,c]); reference AdaIn
    (core/models/hologan_generator.py:333-345) followed by ReLU."""

    @staticmethod
    def forward(ctx, x, scale, bias, eps, act, slope):
        x, scale, bias = _req(x, "x"), _req(scale, "scale"), _req(bias, "bias")
        N, C = x.shape[:2]
        inner = x.numel() // (N * C)
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty_like(x)
        check(lib.gz_rownorm_act_fwd(_p(x), _p(scale), _p(bias), _p(coef), _p(out), N, C, inner, eps, 1, 1, act, slope,
                                     _stream()), "rownorm_act_fwd(adain)")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ds = torch.empty((N, C), device=x.device, dtype=torch.float32)
        db = torch.empty((N, C), device=x.device, dtype=torch.float32)
        kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        check(lib.gz_norm_act_bwd(_p(gout), _p(x), _p(coef), _p(dx), _p(ds), _p(db), _p(ws), _p(kbuf), N, C, inner, 0,
                                  1, 1, act, slope, _stream()), "norm_act_bwd(adain)")
        return dx, ds, db, None, None, None


def adain_act(x, scale, bias, eps=1e-8, act=ACT_RELU, slope=0.0):
    return _AdaINAct.apply(x, scale, bias, eps, act, slope)


class _AdaINActPacked(torch.autograd.Function):
    """adain_act with scale and shift given as the two halves of ONE [N, 2C] tensor (the ZMapping output,
    hologan_generator.py:15-18): no slicing / copying on the way in, and the gradient comes back as one [N, 2C]
    tensor written by the kernel (the framework spelling cost 2 copies forward and a zeros + 2 slice copies + add
    backward per block)."""

    @staticmethod
    def forward(ctx, x, sb, eps, act, slope):
        x, sb = _req(x, "x"), _req(sb, "scale|shift")
        N, C = x.shape[:2]
        if sb.shape != (N, 2 * C):
            raise RuntimeError("adain_act_packed: expected scale|shift of shape [N, 2C]")
        inner = x.numel() // (N * C)
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        sbp = sb.data_ptr()
        out = torch.empty_like(x)
        check(lib.gz_rownorm_act_fwd(_p(x), ctypes.c_void_p(sbp), ctypes.c_void_p(sbp + 4 * C), _p(coef), _p(out), N, C,
                                     inner, eps, 2, 1, act, slope, _stream()), "rownorm_act_fwd(adain, packed)")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dsb = torch.empty((N, 2 * C), device=x.device, dtype=torch.float32)
        kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        p0 = dsb.data_ptr()
        check(lib.gz_norm_act_bwd(_p(gout), _p(x), _p(coef), _p(dx), ctypes.c_void_p(p0), ctypes.c_void_p(p0 + 4 * C),
                                  _p(ws), _p(kbuf), N, C, inner, 0, 2, 1, act, slope, _stream()),
              "norm_act_bwd(adain, packed)")
        return dx, dsb, None, None, None


def adain_act_packed(x, sb, eps=1e-8, act=ACT_RELU, slope=0.0):
    return _AdaINActPacked.apply(x, sb, eps, act, slope)


class _AdaINConst(torch.autograd.Function):
    """adain_act_packed(x.repeat(N, ...), sb) for a constant x of shape [1, C, ...] without materialising the
    repeat or the per-sample input gradient (reference hologan_generator.py:141-142)."""

    @staticmethod
    def forward(ctx, x, sb, eps, act, slope):
        x, sb = _req(x, "x"), _req(sb, "scale|shift")
        N, C = sb.shape[0], x.shape[1]
        inner = x.numel() // C
        if x.shape[0] != 1 or sb.shape[1] != 2 * C:
            raise RuntimeError("adain_const: expected x [1, C, ...] and scale|shift [N, 2C]")
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty((N,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        check(lib.gz_adain_const_fwd(_p(x), _p(sb), _p(coef), _p(out), N, C, inner, eps, act, slope, _stream()),
              "adain_const_fwd")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x)
        dsb = torch.empty((N, 2 * C), device=x.device, dtype=torch.float32)
        check(lib.gz_adain_const_bwd(_p(gout), _p(x), _p(coef), _p(dx), _p(dsb), N, C, inner, act, slope, _stream()),
              "adain_const_bwd")
        return (_sink_or_return(ctx.x_ref, dx) if ctx.needs_input_grad[0] else None), dsb, None, None, None


def adain_const_act(x, sb, eps=1e-8, act=ACT_RELU, slope=0.0):
    if x.numel() // x.shape[1] > 1024:          # rows longer than the constant-input kernel keeps in registers
        return adain_act_packed(x.repeat(sb.shape[0], *([1] * (x.dim() - 1))), sb, eps, act, slope)
    return _AdaINConst.apply(x, sb, eps, act, slope)


class _LinearAct(torch.autograd.Function):
    """act(x @ W^T + b) with bias and activation fused in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x, weight = _req(x, "x"), _req(weight, "weight")
        out = gemm(x, weight, bias, trans_b=True, act=act, slope=slope)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, bias is not None
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = _req(g)
        if ctx.act != ACT_NONE:
            if g.numel() % 4 == 0:
                g = _act_bwd_raw(g, out, ctx.act, ctx.slope)
            else:   # odd tiny shapes
                d = {ACT_RELU: (out > 0).float(), ACT_TANH: 1 - out * out}.get(ctx.act)
                g = g * (d if d is not None else torch.where(out > 0, 1.0, ctx.slope))
        dx = gemm(g, weight) if ctx.needs_input_grad[0] else None
        dw = gemm(g, x, trans_a=True) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(g.shape[1], device=g.device, dtype=torch.float32)
            check(lib.gz_colsum(_p(g), _p(db), g.shape[0], g.shape[1], _stream()), "colsum")
            db = _sink_or_return(ctx.bias_ref, db) if ctx.bias_ref is not None else db
        if dw is not None:
            dw = _sink_or_return(weight, dw)
        return dx, dw, db, None, None


def linear_act(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    return _LinearAct.apply(x, weight, bias, act, slope)


class _LinearActMulti(torch.autograd.Function):
    """Several act(x @ W_j^T + b_j) over ONE x in one launch; one launch for every dW_j / db_j backward."""

    @staticmethod
    def forward(ctx, x, act, slope, *wb):
        x = _req(x, "x")
        ws = [_req(w, "weight") for w in wb[0::2]]
        bs = [None if b is None else _req(b, "bias") for b in wb[1::2]]
        N, K = x.shape
        if any(w.dim() != 2 or w.shape[1] != K for w in ws) or len(ws) > lib.gz_linear_multi_max_jobs():
            raise ValueError("linear_act_multi: weights must be [J, %d], at most %d of them"
                             % (K, lib.gz_linear_multi_max_jobs()))
        outs = tuple(torch.empty((N, w.shape[0]), device=x.device, dtype=torch.float32) for w in ws)
        table = (ctypes.c_char * lib.gz_linear_multi_table_bytes())()
        for w, b, o in zip(ws, bs, outs):
            check(lib.gz_linear_multi_add(table, _p(w), _p(b), _p(o), None, None, None, w.shape[0]), "linear_multi_add")
        check(lib.gz_linear_multi_fwd(table, _p(x), N, K, act, slope, _stream()), "linear_multi_fwd")
        ctx.save_for_backward(x, *ws, *outs)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, [b is not None for b in bs]
        ctx.params = tuple(wb)               # the Parameters themselves: their gradients join the sinks
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        n = len(ctx.has_bias)
        x, ws, outs = saved[0], saved[1:1 + n], saved[1 + n:]
        N, K = x.shape
        gs = [torch.zeros_like(o) if g is None else _req(g) for g, o in zip(gs, outs)]
        grads = []
        table = (ctypes.c_char * lib.gz_linear_multi_table_bytes())()
        for j, (w, o, g) in enumerate(zip(ws, outs, gs)):
            dw = torch.empty_like(w)
            db = torch.empty(w.shape[0], device=w.device, dtype=torch.float32) if ctx.has_bias[j] else None
            check(lib.gz_linear_multi_add(table, None, None, _p(o), _p(g), _p(dw), _p(db), w.shape[0]), "linear_multi_add")
            grads += [dw, db]
        check(lib.gz_linear_multi_bwd(table, _p(x), N, K, ctx.act, ctx.slope, _stream()), "linear_multi_bwd")
        dx = None
        if ctx.needs_input_grad[0]:          # (z is noise in the shipped models: not on their path)
            for w, o, g in zip(ws, outs, gs):
                gm = g if ctx.act == ACT_NONE else g * _act_derivative(o, ctx.act, ctx.slope)
                part = gemm(gm.contiguous(), w)
                dx = part if dx is None else dx + part
        # complete gradients: one-slab sources of the sink flush (under data parallelism p.grad is a view of the flat
        # exchange buffer and autograd would spend an `add_` launch per parameter: 10 per generator step)
        grads = [None if not ctx.needs_input_grad[3 + j] else _sink_or_return(p_, g_)
                 for j, (p_, g_) in enumerate(zip(ctx.params, grads))]
        return (dx, None, None, *grads)


def _act_derivative(out, act, slope):
    if act == ACT_RELU:
        return (out > 0).float()
    if act == ACT_TANH:
        return 1 - out * out
    return torch.where(out > 0, 1.0, slope)


def linear_act_multi(x, layers, act=ACT_NONE, slope=0.0):
    """``[linear_act(x, w, b, act) for (w, b) in layers]`` in ONE launch (and one launch for all the weight and bias
    gradients): HoloGAN's five ZMapping layers over the same z (reference core/models/hologan_generator.py:7-19)."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _LinearActMulti.apply(x, act, slope, *flat)


class _RigidResample(torch.autograd.Function):
    """[N,C,S,S,S] voxels + [N,16] inverse view matrices -> [N, C*S, S, S] projected feature map."""

    @staticmethod
    def forward(ctx, vox, minv):
        vox, minv = _req(vox, "vox"), _req(minv, "minv")
        N, C, S = vox.shape[0], vox.shape[1], vox.shape[2]
        out = torch.empty((N, C * S, S, S), device=vox.device, dtype=torch.float32)
        check(lib.gz_rigid_resample_fwd(_p(vox), _p(minv), _p(out), None, N, C, S, _stream()), "rigid_resample_fwd")
        ctx.save_for_backward(minv)
        ctx.shape = (N, C, S)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (minv,) = ctx.saved_tensors
        N, C, S = ctx.shape
        g = _req(g)
        gv = torch.empty((N, C, S, S, S), device=g.device, dtype=torch.float32)
        ws, nbytes = _scratch((lib.gz_rigid_resample_bwd_workspace_bytes(N, S) + 3) // 4 * 4, g.device)
        check(lib.gz_rigid_resample_bwd(_p(g), _p(minv), _p(gv), _p(ws), nbytes, N, C, S, _stream()),
              "rigid_resample_bwd")
        return gv, None


def rigid_resample(vox, minv):
    return _RigidResample.apply(vox, minv)


def rigid_resample_indices(vox, minv):
    """Debug / test hook: the int64 corner indices idx_a..idx_h the kernel uses, [8, N*S^3]."""
    N, C, S = vox.shape[0], vox.shape[1], vox.shape[2]
    out = torch.empty((N, C * S, S, S), device=vox.device, dtype=torch.float32)
    idx = torch.empty((8, N * S ** 3), device=vox.device, dtype=torch.int64)
    check(lib.gz_rigid_resample_fwd(_p(_req(vox)), _p(_req(minv)), _p(out), _p(idx), N, C, S, _stream()),
          "rigid_resample_fwd")
    return out, idx


class _SpectralNormWeight(torch.autograd.Function):
    """weight_orig / sigma with sigma = u^T W v after one power iteration (training: the module's u / v buffers are
    updated in place, like torch.nn.utils.spectral_norm); u, v enter sigma as constants.  Six launches forward, two
    backward (the framework spelling took ~16 + ~10)."""

    @staticmethod
    def forward(ctx, weight_orig, u, v, training, eps):
        W = _req(weight_orig, "weight_orig")
        R = W.shape[0]
        L = W.numel() // R
        st = _stream()
        dev = W.device
        us = torch.empty(R, device=dev, dtype=torch.float32)       # the copies this node keeps (the buffers move on)
        vs = torch.empty(L, device=dev, dtype=torch.float32)
        sigma = torch.empty(1, device=dev, dtype=torch.float32)
        Wm = W.view(R, L)
        if training:
            v_raw = _coldot_raw(u, Wm)                              # W^T u
            check(lib.gz_vec_normalize(_p(v_raw), _p(v), _p(vs), None, L, eps, st), "vec_normalize(v)")
            wv = _rowdot_raw(Wm, vs, True)                          # W v
            check(lib.gz_vec_normalize_dot(_p(wv), _p(u), _p(us), _p(sigma), R, eps, st), "vec_normalize(u), sigma")
        else:
            us.copy_(u)
            vs.copy_(v)
            wv = _rowdot_raw(Wm, vs, True)
            check(lib.gz_vec_dot(_p(us), _p(wv), _p(sigma), R, st), "vec_dot(sigma)")
        w = torch.empty_like(W)
        check(lib.gz_div_scalar(_p(W), _p(sigma), _p(w), W.numel(), st), "div_scalar")
        ctx.save_for_backward(w, us, vs, sigma)
        return w

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w, us, vs, sigma = ctx.saved_tensors
        g = _req(g)
        R = w.shape[0]
        L = w.numel() // R
        rowdots = _rowdot_raw(g.view(R, L), w.view(R, L), False)
        out = torch.empty_like(w)
        check(lib.gz_spectral_norm_bwd(_p(g), _p(rowdots), _p(us), _p(vs), _p(sigma), _p(out), R, L, _stream()),
              "spectral_norm_bwd")
        return out, None, None, None, None


def spectral_normalize(weight_orig, u, v, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's weight: one power iteration (training: u, v updated in place),
    then weight_orig / sigma with sigma = u^T W v differentiable w.r.t. weight_orig.  The two
    matrix-vector products per iteration run on gz_coldot / gz_rowdot, the rest on csrc/gz_loss.hip."""
    L = weight_orig.numel() // weight_orig.shape[0]
    if L % 4 == 0:
        return _SpectralNormWeight.apply(weight_orig, u, v, bool(training), float(eps))
    w_mat = weight_orig.reshape(weight_orig.shape[0], -1)           # odd row lengths: framework arithmetic
    if training:
        with torch.no_grad():
            wd = _req(w_mat.detach())
            v_new = _coldot_raw(u, wd)                       # W^T u
            v_new = v_new / v_new.norm().clamp_min(eps)
            u_new = _rowdot_raw(wd, v_new, True)             # W v
            u_new = u_new / u_new.norm().clamp_min(eps)
            v.copy_(v_new)
            u.copy_(u_new)
    uc, vc = u.clone(), v.clone()
    sigma = torch.dot(uc, _DotF.apply(w_mat, vc))
    return weight_orig / sigma


class _SpectralNormMulti(torch.autograd.Function):
    """spectral_normalize for SEVERAL conv weights of one discriminator call: the power iterations in four launches
    (gz_sn_power_iteration), w = weight_orig / sigma together with both packed images of every layer in a fifth
    (gz_conv2d_pack_table_launch) -- 8 launches per layer before.  The packed images ride on the returned tensors
    (``_gz_packs``) and are picked up by ``_packed``."""

    @staticmethod
    def forward(ctx, eps, geom, *wuv):
        ctx.params = wuv[0::3]
        Ws = [_req(w, "weight_orig") for w in wuv[0::3]]
        us_buf, vs_buf = wuv[1::3], wuv[2::3]
        dev, st = Ws[0].device, _stream()
        if len(Ws) > lib.gz_sn_max_jobs() or 3 * len(Ws) > lib.gz_conv2d_pack_table_max_jobs():
            raise ValueError("spectral_normalize_multi: at most %d layers" % lib.gz_sn_max_jobs())
        table = (ctypes.c_char * lib.gz_sn_table_bytes())()
        packs = (ctypes.c_char * lib.gz_conv2d_pack_table_bytes())()
        keep, saved, outs = [], [], []
        for W, u, v in zip(Ws, us_buf, vs_buf):
            K, C, KH, KW = W.shape
            R, L = K, C * KH * KW
            us = torch.empty(R, device=dev, dtype=torch.float32)
            vs = torch.empty(L, device=dev, dtype=torch.float32)
            sigma = torch.empty(1, device=dev, dtype=torch.float32)
            ws = _ws(lib.gz_sn_workspace_floats(R, L), dev)
            check(lib.gz_sn_add(table, _p(W), _p(u), _p(v), _p(us), _p(vs), _p(sigma), _p(ws), R, L), "sn_add")
            w = torch.empty_like(W)
            wf = torch.empty(lib.gz_conv2d_pack_fwd_elems(K, C, KH, KW), device=dev, dtype=torch.float32)
            wd = torch.empty(lib.gz_conv2d_pack_dgrad_elems(K, C, KH, KW, geom.stride), device=dev, dtype=torch.float32)
            for what, dst in ((2, w), (0, wf), (1, wd)):
                check(lib.gz_conv2d_pack_table_add(packs, _p(W), _p(dst), _p(sigma), what, K, C, KH, KW, geom.stride,
                                                   geom.pad), "pack_table_add")
            w._gz_packs = {"f": wf, "d": wd, "geom": geom, "version": w._version}
            keep.append(ws)
            saved += [w, us, vs, sigma]
            outs.append(w)
        check(lib.gz_sn_power_iteration(table, eps, st), "sn_power_iteration")
        check(lib.gz_conv2d_pack_table_launch(packs, st), "pack_table_launch")
        del keep            # (stream-ordered allocator: the workspaces may be reused by later launches of this stream)
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        grads = []
        for k, g in enumerate(gs):
            w, us, vs, sigma = saved[4 * k: 4 * k + 4]
            if g is None:
                grads += [None, None, None]
                continue
            g = _req(g)
            R = w.shape[0]
            L = w.numel() // R
            rowdots = _rowdot_raw(g.view(R, L), w.view(R, L), False)
            out = torch.empty_like(w)
            check(lib.gz_spectral_norm_bwd(_p(g), _p(rowdots), _p(us), _p(vs), _p(sigma), _p(out), R, L, _stream()),
                  "spectral_norm_bwd")
            grads += [_sink_or_return(ctx.params[k], out), None, None]
        return (None, None, *grads)


def spectral_normalize_multi(layers, training, geom, eps=1e-12):
    """``[spectral_normalize(W, u, v, training) for (W, u, v) in layers]`` for the conv weights of one discriminator call
    (all of geometry ``geom``); training mode on 4-D weights with C*KH*KW % 4 == 0 takes the five-launch path."""
    ok = training and all(W.dim() == 4 and (W.numel() // W.shape[0]) % 4 == 0 and W.is_cuda and W.is_contiguous()
                          and W.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 for W, _, v in layers)
    if not ok or len(layers) > lib.gz_sn_max_jobs():
        return [spectral_normalize(W, u, v, training, eps) for W, u, v in layers]
    flat = []
    for W, u, v in layers:
        flat += [W, u, v]
    return list(_SpectralNormMulti.apply(float(eps), geom, *flat))


def spectral_power_iterations(layers, calls=1, eps=1e-12):
    """``calls`` consecutive power iterations of torch.nn.utils.spectral_norm for each (weight_orig, u, v) -- what
    ``calls`` discriminator forward passes would run one after the other (the module buffers end up where the last call
    leaves them).  Returns per layer (sigma [calls], u [calls, R], v [calls, L]): the values each call works with.
    No autograd: sigma's dependence on the weight is handled by sn_conv_in_act's backward (u, v constants, as in torch)."""
    st = _stream()
    out = []
    with torch.no_grad():
        for W, _, _ in layers:
            R, L = W.shape[0], W.numel() // W.shape[0]
            out.append((torch.empty(calls, device=W.device, dtype=torch.float32),
                        torch.empty((calls, R), device=W.device, dtype=torch.float32),
                        torch.empty((calls, L), device=W.device, dtype=torch.float32)))
        keep = []
        for g in range(calls):
            table = (ctypes.c_char * lib.gz_sn_table_bytes())()
            for (W, u, v), (sig, us, vs) in zip(layers, out):
                W = _req(W.detach(), "weight_orig")
                R, L = W.shape[0], W.numel() // W.shape[0]
                ws = _ws(lib.gz_sn_workspace_floats(R, L), W.device)
                keep.append(ws)
                check(lib.gz_sn_add(table, _p(W), _p(u), _p(v), _p(us[g]), _p(vs[g]), _p(sig[g:g + 1]), _p(ws), R, L),
                      "sn_add")
            check(lib.gz_sn_power_iteration(table, eps, st), "sn_power_iteration")
        del keep
    return out


class _SNConvINAct(torch.autograd.Function):
    """act(InstanceNorm(conv2d(x, weight_orig / sigma) + bias)) for the spectral-normalised blocks of HoloGAN's critic
    (reference core/models/hologan_discriminator.py:28-38), computed as act(IN_{eps sigma^2}(conv2d(x, weight_orig))):
    the InstanceNorm removes the bias and any scale except through eps, so the convolution runs on weight_orig's own
    packed images (cached for the whole optimizer step instead of re-packed at every call) and ``groups`` discriminator
    calls with different sigma can share one pass over the stacked batch.  Backward: weight_orig's gradient is the
    convolution's weight gradient plus dL/dsigma_g u_g v_g^T (gz_sn_sigma_term: exactly torch's -(sum g w) u v^T / sigma);
    the bias gradient is exactly zero (the reference's is rounding noise)."""

    @staticmethod
    def forward(ctx, x, weight_orig, bias, sigma, us, vs, geom, in_eps, act, slope):
        x, W = _req(x, "x"), _req(weight_orig, "weight_orig")
        groups = sigma.numel()
        N = x.shape[0]
        y = _conv_fwd_raw(x, W, None, geom, ACT_NONE, 0.0)
        C, inner = y.shape[1], y.shape[2] * y.shape[3]
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty_like(y)
        check(lib.gz_rownorm_act_fwd_sigma(_p(y), _p(sigma), groups, _p(coef), _p(out), N, C, inner, in_eps, act, slope,
                                           _stream()), "rownorm_act_fwd_sigma")
        ctx.save_for_backward(x, W, y, coef, sigma, us, vs)
        ctx.cfg = (geom, in_eps, act, slope, groups, bias is not None)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.param = weight_orig
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, W, y, coef, sigma, us, vs = ctx.saved_tensors
        geom, in_eps, act, slope, groups, has_bias = ctx.cfg
        gout = _req(gout)
        N, C = y.shape[0], y.shape[1]
        inner = y.shape[2] * y.shape[3]
        st = _stream()
        g_raw = torch.empty_like(y)
        rowsums = torch.empty(2 * N * C, device=y.device, dtype=torch.float32)
        check(lib.gz_rownorm_act_bwd_rows(_p(gout), _p(y), _p(coef), _p(g_raw), _p(rowsums), N, C, inner, act, slope, st),
              "rownorm_act_bwd_rows")
        dx = _conv_dgrad_raw(g_raw, W, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            R, L = W.shape[0], W.numel() // W.shape[0]
            coefs = torch.empty(lib.gz_sn_sigma_coef_floats(groups), device=y.device, dtype=torch.float32)
            term = torch.empty_like(W)
            check(lib.gz_sn_sigma_term(_p(rowsums), _p(coef[3 * N * C:]), _p(sigma), _p(us), _p(vs), _p(coefs), _p(term),
                                       N * C, groups, R, L, in_eps, st), "sn_sigma_term")
            p = ctx.param
            # weight_orig is a leaf: the convolution's weight gradient and the sigma term join its sink as two sources
            sunk = isinstance(p, torch.nn.Parameter) and not (W.numel() & 3) and _sink_conv_wgrad(p, x, g_raw, geom)
            if sunk:     # (setdefault: a weight whose launch GradSync postponed has no entry yet)
                _sinks.pending.setdefault(id(p), [p, []])[1].append((term, 1, W.numel()))
            if not sunk:
                dW = _conv_wgrad_raw(x, g_raw, geom)
                dW.add_(term)
        db = _sink_zero(ctx.bias_ref, ctx.bias_shape, y.device) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dW, db, None, None, None, None, None, None, None


def sn_conv_in_act(x, weight_orig, bias, sigma, us, vs, geom, in_eps=1e-5, act=ACT_NONE, slope=0.0):
    """One spectral-normalised conv + InstanceNorm2d(affine=False) + activation block; sigma [groups], us [groups, R],
    vs [groups, L] from spectral_power_iterations (groups = discriminator calls stacked along the batch)."""
    return _SNConvINAct.apply(x, weight_orig, bias, sigma, us, vs, geom, float(in_eps), act, slope)


# ---------------------------------------------------------------------------
# loss heads (csrc/gz_loss.hip)
# ---------------------------------------------------------------------------
class _BCELogitsMean(torch.autograd.Function):
    """mean BCE-with-logits against a constant target: criterion(x, ones_like(x)) / zeros_like(x)."""

    @staticmethod
    def forward(ctx, x, target):
        x = _req(x, "logits").reshape(-1)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.gz_bce_logits_mean(_p(x), _p(loss), x.numel(), float(target), _stream()), "bce_logits_mean")
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _req(g).reshape(1)
        dx = torch.empty_like(x)
        check(lib.gz_bce_logits_mean_bwd(_p(x), _p(g), _p(dx), x.numel(), ctx.target, _stream()), "bce_logits_mean_bwd")
        return dx, None


def bce_logits_mean(logits, target):
    return _BCELogitsMean.apply(logits.reshape(-1), target)


class _PairLoss(torch.autograd.Function):
    """Loss head over the stacked logits [first half; second half] (gz_pair_loss): mode 0 = the mean of two BCE means
    against constants t0 / t1, mode 1 = t0 * mean(first) + t1 * mean(second)."""

    @staticmethod
    def forward(ctx, x, t0, t1, mode):
        x = _req(x, "logits").reshape(-1)
        if x.numel() % 2:
            raise RuntimeError("pair loss: an even number of logits (two stacked batches) is required")
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.gz_pair_loss(_p(x), _p(loss), x.numel() // 2, float(t0), float(t1), int(mode), _stream()), "pair_loss")
        ctx.save_for_backward(x)
        ctx.cfg = (float(t0), float(t1), int(mode))
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        t0, t1, mode = ctx.cfg
        g = _req(g).reshape(1)
        dx = torch.empty_like(x)
        check(lib.gz_pair_loss_bwd(_p(x), _p(g), _p(dx), x.numel() // 2, t0, t1, mode, _stream()), "pair_loss_bwd")
        return dx, None, None, None


def bce_logits_pair_mean(logits, t_first, t_second):
    """(BCE(logits[:n], t_first).mean() + BCE(logits[n:], t_second).mean()) / 2 in one launch."""
    return _PairLoss.apply(logits.reshape(-1), t_first, t_second, 0)


def weighted_half_means(logits, w_first, w_second):
    """w_first * logits[:n].mean() + w_second * logits[n:].mean() in one launch."""
    return _PairLoss.apply(logits.reshape(-1), w_first, w_second, 1)


class _MSEMean(torch.autograd.Function):
    """mean((a - b)^2) with b a constant (HoloGAN's q_loss: b is the latent the generator was fed)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib.gz_mse_mean(_p(a), _p(b), _p(loss), a.numel(), _stream()), "mse_mean")
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _req(g).reshape(1)
        da = torch.empty_like(a)
        check(lib.gz_mse_mean_bwd(_p(a), _p(b), _p(g), _p(da), a.numel(), _stream()), "mse_mean_bwd")
        return da, (-da if ctx.needs_input_grad[1] else None)


def mse_mean(a, b):
    return _MSEMean.apply(a, b)


# ---------------------------------------------------------------------------
# R1-regularised ResNet path (reference core/submodules/gan_stability/models/resnet.py).
# Every op here is linear or piecewise linear, so forward/adjoint pairs close under differentiation and
# compute_grad2's create_graph=True (core/utils/utils.py:60-69) works to any order.
# ---------------------------------------------------------------------------
K3S1P1 = Geom(3, 3, 1, 1)
K1S1P0 = Geom(1, 1, 1, 0)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _req(x, "x")
        y = torch.empty_like(x)
        check(lib.gz_act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream()), "act_fwd")
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _ActBwd.apply(g, y, ctx.act, ctx.slope), None, None


def activation(x, act=ACT_LRELU, slope=0.2):
    return _Act.apply(x, act, slope)


def _axpby_raw(a, alpha, b, beta, act_out=None, act=ACT_NONE, slope=0.0):
    out = torch.empty_like(a)
    check(lib.gz_axpby(_p(a), alpha, _p(b), beta, _p(out), _p(act_out), a.numel(), act, slope, _stream()), "axpby")
    return out


class _Scale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, alpha):
        ctx.alpha = alpha
        return _axpby_raw(_req(a, "a"), alpha, None, 0.0)

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.alpha), None


class _AddScaled(torch.autograd.Function):
    """out = a + beta*b (the residual tail, resnet.py:121-122)."""

    @staticmethod
    def forward(ctx, a, b, beta):
        ctx.beta = beta
        return _axpby_raw(_req(a, "a"), 1.0, _req(b, "b"), beta)

    @staticmethod
    def backward(ctx, g):
        ga = g if ctx.needs_input_grad[0] else None
        gb = _Scale.apply(g, ctx.beta) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class _AddScaledAct(torch.autograd.Function):
    """(out, act(out)) with out = a + beta*b: the residual tail and the next block's pre-activation in one pass."""

    @staticmethod
    def forward(ctx, a, b, beta, act, slope):
        a, b = _req(a, "a"), _req(b, "b")
        act_out = torch.empty_like(a)
        out = _axpby_raw(a, 1.0, b, beta, act_out, act, slope)
        ctx.beta, ctx.act, ctx.slope = beta, act, slope
        ctx.save_for_backward(act_out)
        return out, act_out

    @staticmethod
    def backward(ctx, g_out, g_act):
        (act_out,) = ctx.saved_tensors
        g = None
        if g_act is not None:
            g = _ActBwd.apply(g_act, act_out, ctx.act, ctx.slope)
        if g_out is not None:
            g = g_out if g is None else _AddScaled.apply(g_out, g, 1.0)
        ga = g if ctx.needs_input_grad[0] else None
        gb = _Scale.apply(g, ctx.beta) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None, None


def scale(a, alpha):
    return _Scale.apply(a, alpha)


def add_scaled(a, b, beta):
    return _AddScaled.apply(a, b, beta)


def add_scaled_act(a, b, beta, act=ACT_LRELU, slope=0.2):
    return _AddScaledAct.apply(a, b, beta, act, slope)


def _planes(x):
    if x.dim() != 4:
        raise RuntimeError("lightning_gan_zoo_amd: expected an NCHW tensor, got shape %s" % (tuple(x.shape),))
    return x.shape[0] * x.shape[1]


def _avgpool_fwd_raw(x):
    N, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
    check(lib.gz_avgpool3s2_fwd(_p(x), _p(y), _planes(x), H, W, OH, OW, _stream()), "avgpool3s2_fwd")
    return y


def _avgpool_bwd_raw(gy, hw):
    N, C, OH, OW = gy.shape
    H, W = hw
    gx = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32)
    check(lib.gz_avgpool3s2_bwd(_p(gy), _p(gx), _planes(gy), H, W, OH, OW, _stream()), "avgpool3s2_bwd")
    return gx


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = tuple(x.shape[2:])
        return _avgpool_fwd_raw(_req(x, "x"))

    @staticmethod
    def backward(ctx, g):
        return _AvgPoolT.apply(g, ctx.hw)


class _AvgPoolT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, hw):
        return _avgpool_bwd_raw(_req(g, "g"), hw)

    @staticmethod
    def backward(ctx, v):
        return _AvgPool.apply(v), None


def avg_pool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1) (resnet.py:72)."""
    return _AvgPool.apply(x)


def _upsample_fwd_raw(x):
    N, C, H, W = x.shape
    y = torch.empty((N, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(lib.gz_upsample2_fwd(_p(x), _p(y), _planes(x), H, W, _stream()), "upsample2_fwd")
    return y


def _upsample_bwd_raw(gy):
    N, C, H2, W2 = gy.shape
    gx = torch.empty((N, C, H2 // 2, W2 // 2), device=gy.device, dtype=torch.float32)
    check(lib.gz_upsample2_bwd(_p(gy), _p(gx), _planes(gy), H2 // 2, W2 // 2, _stream()), "upsample2_bwd")
    return gx


class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _upsample_fwd_raw(_req(x, "x"))

    @staticmethod
    def backward(ctx, g):
        return _UpsampleT.apply(g)


class _UpsampleT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g):
        return _upsample_bwd_raw(_req(g, "g"))

    @staticmethod
    def backward(ctx, v):
        return _Upsample.apply(v)


def upsample2(x):
    """nn.Upsample(scale_factor=2), nearest (resnet.py:31)."""
    return _Upsample.apply(x)


# ---------------------------------------------------------------------------
# input step
# ---------------------------------------------------------------------------
def normalize_u8_images(u8, mean, std):
    """Decoded uint8 images [N,H,W,C] on the GPU -> float [N,C,H,W] = (x / 255 - mean) / std: ToTensor() +
    Normalize(mean, std) of reference core/lightning_module.py:42-47, one pass on the device."""
    if not u8.is_cuda or u8.dtype != torch.uint8 or u8.dim() != 4:
        raise RuntimeError("lightning_gan_zoo_amd: expected a uint8 NHWC tensor on the GPU")
    u8 = u8 if u8.is_contiguous() else u8.contiguous()
    N, H, W, C = u8.shape
    out = torch.empty((N, C, H, W), device=u8.device, dtype=torch.float32)
    check(lib.gz_u8hwc_to_nchw(_p(u8), _p(out), N, H, W, C, float(mean), float(std), _stream()), "u8hwc_to_nchw")
    return out
