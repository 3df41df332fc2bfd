#!/usr/bin/env python3
"""Headline benchmark: images/sec of the DCGAN G+D training step (64x64, synthetic data, fp32, bs=128/GPU --
the batch BASELINE.json's metric string names) on N MI355X GPUs of one node, through the reference's own surface
(training_step / backward / optimizer.step with Lightning's per-batch optimizer alternation and toggle).
The same run also measures BASELINE config 2 / 4 (bs=512/GPU, the north star's roofline batch) as
``sub_configs.dc_gan_bs512`` -- on one GPU and, with ``--gpus N``, on all N ranks -- and, on one GPU, config 3
(wgan_gp bs 256), config 5's per-GPU workload (hologan bs 64, at 64x64 and as EXT-128) and wgan bs 512.

    python bench.py [--gpus N --steps K --warmup W --batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one G+D pair: training_step(optimizer_idx=0)+backward+Adam on one batch of B reals,
then training_step(optimizer_idx=1)+backward+Adam on another; images/sec
counts both batches (2*B*N / t_pair).  Prints ONE JSON line on rank 0.

``--gpus N`` with N > 1 and no torch.distributed environment: this process only spawns the N ranks
(``python -m torch.distributed.run``, before anything here touches a GPU) and relays their line.
The timed region is K pairs bracketed by barrier + synchronize, repeated ``--reps`` times (default 3);
``ms_per_step`` / ``value`` are the MEDIAN repetition (max over ranks inside each).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic conv/GEMM FLOP per sample per optimizer cycle (BASELINE.md section 3 / SURVEY.md 8-d)
FLOP_PER_SAMPLE_CYCLE = {
    "dc_gan": 4_929_880_064,              # D step 2,054,389,760 + G step 2,875,490,304
    "wgan": 5 * 2_054_389_760 + 2_875_490_304,
    "wgan_gp": 6_175_700_000,             # D step 3.3002 G (3 D fwd + GP double backward) + G step 2.8755 G
    "hologan": 29_060_000_000,            # D + G + G at 64x64: 5.6244 G + 2 x 11.7195 G
    # SURVEY 8-f4, conf/expt/gan_stability_r1.yaml (ResNet nfilter 16, 128x128): D step 4,682,563,584 (3 D fwd-size
    # passes + the R1 double backward) + G step 2,354,610,176, traced on the oracle as PyTorch autograd executes it
    "gan_stability_r1": 7_037_173_760,
}
# HoloGAN EXT-128 (the reference cannot run at 128x128): traced the same way on the oracle's
# extension, tests/diagnostics/flop_trace.py hologan 128 -> D step 11.4411 G + 2 x G step 13.7115 G
FLOP_PER_SAMPLE_CYCLE_EXT128 = {"hologan": 38_864_160_640}
NATIVE_IMG_SIZE = {"gan_stability_r1": 128}
# BASELINE.json configs measured in the default run next to the headline (the metric string's dc_gan bs 128 / GPU):
# configs[1] / [3] = dc_gan bs 512 / GPU (every world size), and on one GPU config 3 (wgan_gp bs 256), config 5's
# per-GPU workload (hologan bs 64, at the parity-pinned 64x64 and as EXT-128) and the 5 : 1 wgan cycle
SUB_CONFIGS = (("dc_gan_bs128", "dc_gan", 128, 64), ("dc_gan_bs512", "dc_gan", 512, 64),
               ("wgan_gp_bs256", "wgan_gp", 256, 64), ("hologan_bs64", "hologan", 64, 64),
               ("hologan_ext128_bs64", "hologan", 64, 128), ("wgan_bs512", "wgan", 512, 64))
# what a --gpus N run times on all ranks: the metric's batch, BASELINE config 4 (dc_gan bs 512 / GPU) and config 5
# (hologan 128x128 bs 64 / GPU: EXT-128, throughput only -- the reference cannot run at 128)
MULTI_GPU_SUB_CONFIGS = ("dc_gan_bs128", "dc_gan_bs512", "hologan_ext128_bs64")
# the data-parallel code path on ONE GPU (single-rank RCCL), timed next to the plain trainer in every default run
GRADSYNC_W1_CONFIGS = ("dc_gan_bs128", "dc_gan_bs512", "wgan_gp_bs256", "hologan_ext128_bs64")
DEFAULT_BATCH = {"dc_gan": 128, "wgan": 512, "wgan_gp": 256, "hologan": 64, "gan_stability_r1": 64}
PEAK_FP32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 256 FLOP/clk x 2.4 GHz
OVERLAP_CYCLES = 4  # untimed cycles behind a timed region in which ddp.GradSync brackets its bucket waits with events
TIMER_EVERY = 4     # per-launch HIP events on every 4th cycle of the timed region (they cost ~5 % when always on)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3, help="repetitions of the K-step timed region; the median is reported")
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch (default: BASELINE's: dc_gan 128 = the metric string, wgan_gp 256, hologan 64)")
    ap.add_argument("--expt", default="dc_gan")
    ap.add_argument("--img-size", type=int, default=None,
                    help="default 64 (128 for gan_stability_r1); 128 with --expt hologan is EXT-128, not parity-pinned")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bs128", "--no-sub-configs", dest="no_bs128", action="store_true",
                    help="headline configuration only (skip the dc_gan bs128 / wgan_gp / hologan sub-records)")
    ap.add_argument("--sub-steps", type=int, default=10, help="timed optimizer cycles per repetition of a sub-record")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--detail", action="store_true",
                    help="full sub-configuration records (per-label igemm tables, bucket layouts); the default line keeps "
                         "them to their numbers so that the whole line fits the ~8 KB of stdout a driver record keeps")
    ap.add_argument("--graph", action="store_true",
                    help="replay each optimizer step from a captured HIP graph (single GPU)")
    ap.add_argument("--fid-samples", type=int, default=50000,
                    help="images of the FID feature pass sub-record (sub_configs.fid50k_features); 0 skips it")
    ap.add_argument("--no-gradsync-w1", action="store_true",
                    help="skip the single-rank-RCCL GradSync sub-records of the default run")
    ap.add_argument("--force-grad-sync", action="store_true",
                    help="use the data-parallel gradient path (flat buffers, deferred step) even on one rank")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = DEFAULT_BATCH.get(args.expt, 128)
    if args.img_size is None:
        args.img_size = NATIVE_IMG_SIZE.get(args.expt, 64)
    return args


def spawn_ranks(args):
    """``python bench.py --gpus N`` outside a torch.distributed launch: start the N ranks as a child process group and
    relay rank 0's JSON line.  Nothing in THIS process touches the GPU (no torch import even), so there is no
    exec-after-HIP-init hazard and the children own the devices."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if line is not None:
        print(line, flush=True)
    return r.returncode if line is not None or r.returncode else 1


def build_trainer(expt, batch, device, world, force_sync=False, img_size=64, graph=False):
    import torch
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import GraphedTrainer, Trainer
    cfg = make_cfg(expt, batch_size=batch, img_size=img_size)
    torch.manual_seed(42)                 # run_network.py:27, same seed on every rank
    module = locate(cfg.model.lm["_target_"])(cfg, None).to(device)
    sync = GradSync(module) if (world > 1 or force_sync) else None
    if graph and sync is None:
        return module, GraphedTrainer(module)
    return module, Trainer(module, grad_sync=sync)


def synthetic_batch(batch, device, rank, img_size=64):
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    real = (torch.rand(batch, 3, img_size, img_size, generator=g) * 2 - 1).to(device)
    return real, torch.zeros(batch, dtype=torch.int64, device=device)


def timed_pairs(trainer, batch, steps, warmup, world, timer=None, reps=1, on_timed_start=None):
    """-> (per-repetition seconds for ``steps`` cycles, max over ranks; per-rank seconds of the median repetition)."""
    import torch
    import torch.distributed as dist
    per_pair = len(trainer.order)         # batches per optimizer cycle (2 for dc_gan)
    if timer is not None:
        timer.enabled = False
    for _ in range(warmup * per_pair):
        trainer.step(batch)
    trainer.finish()
    torch.cuda.synchronize()
    # A full (generation 2) Python garbage collection walks every object torch and this package have created --
    # 70 ms here, i.e. ten bs=128 pairs -- and its trigger point is deterministic in the allocation count, so it
    # kept landing inside one of the timed regions.  Collect now and park the survivors in the permanent
    # generation: later collections only look at the step's own short-lived objects (< 1 ms).
    gc.collect()
    gc.freeze()
    if on_timed_start is not None:
        on_timed_start()
    times, per_rank = [], []
    for rep in range(reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps * per_pair):
            if timer is not None:
                timer.enabled = rep == 0 and (i // per_pair) % TIMER_EVERY == 0
            trainer.step(batch)
        trainer.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mine = dt
        if world > 1:
            dist.barrier()
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            ranks = [float(x.item()) for x in allt]
            dt = max(ranks)
        else:
            ranks = [mine]
        times.append(dt)
        per_rank.append(ranks)
    med = sorted(range(reps), key=lambda r: times[r])[reps // 2]
    return times, per_rank[med], times[med]


def cpu_baseline(batch=128, budget_s=20.0):
    """The CPU oracle (a torch.nn restatement of the reference, oracle/reference_cpu.py) timed on this
    box's host cores: same pair, same counting convention.  A reported baseline, not the target.
    The thread count is the best of a short probe (all hardware threads is far from optimal for
    oneDNN on a two-socket host)."""
    import torch
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from oracle.reference_cpu import run_step
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cfg = make_cfg("dc_gan", module_root="oracle.reference_cpu", batch_size=batch)
    torch.manual_seed(42)
    step = locate(cfg.model.lm["_target_"])(cfg, None)
    opts = step.configure_optimizers()
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(batch, 3, 64, 64, generator=g) * 2 - 1
    labels = torch.zeros(batch, dtype=torch.int64)

    def pair():
        t0 = time.perf_counter()
        run_step(step, opts, (real, labels), 0, 0)
        run_step(step, opts, (real, labels), 1, 1)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    best = None
    probed = []
    for nthreads in sorted({min(avail, n) for n in (16, 32, 64, 128, avail)}):       # ... up to the whole host
        torch.set_num_threads(nthreads)
        pair()                             # warm-up at this thread count
        t = pair()
        probed.append(nthreads)
        if best is None or t < best[0]:
            best = (t, nthreads)
        if time.perf_counter() - t_start > budget_s * 0.6:
            break
    torch.set_num_threads(best[1])
    times = []
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 30):
        times.append(pair())
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(2 * batch / med, 1), "unit": "images/s", "cores": best[1], "kind": "port",
            "sample": "oracle/reference_cpu.py DCGAN G+D pair, fp32, bs=%d, %d threads (best of a %s-thread probe, "
                      "%d hardware threads available), median of %d pairs (%.0f ms/pair)"
                      % (batch, best[1], "/".join(map(str, probed)), avail, len(times), med * 1e3)}


_TRAFFIC = None


def load_traffic(config_key, label):
    """(HBM bytes per launch of kernel ``label`` in configuration ``config_key``, stale) from the round's PMC passes
    (profiles/traffic.json, written by tools/pmc_traffic.py from two separate ``rocprofv3 --pmc`` runs of this same
    command: counters cannot be collected inside a timed run).  The file carries the digest of the kernel sources it
    was measured on (``_source_digest``, = the library's gz_source_digest()); ``stale`` is True when the library that is
    running now was built from other sources -- the number then describes an earlier kernel.  (None, None) when that
    configuration was not profiled."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                _TRAFFIC = json.load(f)
        except Exception:  # noqa: BLE001
            _TRAFFIC = {}
    per_cfg = _TRAFFIC.get(config_key)
    if not isinstance(per_cfg, dict) or label not in per_cfg:
        return None, None
    try:
        from lightning_gan_zoo_amd._lib import lib
        now = lib.gz_source_digest().decode()
    except Exception:  # noqa: BLE001
        now = None
    measured_on = (_TRAFFIC.get("_source_digest") or {})
    measured_on = measured_on.get(config_key) if isinstance(measured_on, dict) else measured_on
    return per_cfg[label], (measured_on is None or now is None or measured_on != now)


def roofline_of(timer, ms_per_step, steps, flop_cycle, config_key=None):
    """The igemm kernel with the largest total time in the sampled cycles, plus the whole step."""
    agg = timer.summary()
    if not agg:
        return None
    sampled_cycles = (steps + TIMER_EVERY - 1) // TIMER_EVERY      # cycles 0, 4, 8, ... of the first repetition
    total_ms = sum(v[1] for v in agg.values())
    label, (n, ms, fl) = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = fl / (ms * 1e-3) / 1e12
    traffic, stale = load_traffic(config_key, label)
    return {
        "bound": "mfma", "kernel": label, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        "traffic": traffic, "traffic_stale": stale,
        "launches": n, "avg_launch_ms": round(ms / n, 4),
        "sampled_cycles": sampled_cycles,
        "share_of_step": round(ms / (ms_per_step * sampled_cycles), 3),
        "all_igemm": {k: {"launches": v[0], "ms": round(v[1], 2),
                          "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in sorted(agg.items())},
        "igemm_share_of_step": round(total_ms / (ms_per_step * sampled_cycles), 3),
        "whole_step": None if flop_cycle != flop_cycle else {
            "flop_per_step": flop_cycle,
            "achieved": round(flop_cycle / (ms_per_step * 1e-3) / 1e12, 2),
            "frac": round(flop_cycle / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
    }


def flop_per_cycle(expt, batch, img_size):
    if img_size == NATIVE_IMG_SIZE.get(expt, 64):
        return FLOP_PER_SAMPLE_CYCLE[expt] * batch
    if img_size == 128 and expt in FLOP_PER_SAMPLE_CYCLE_EXT128:
        return FLOP_PER_SAMPLE_CYCLE_EXT128[expt] * batch
    return float("nan")                # no traced FLOP count for this size


def config_key_of(expt, batch, img_size):
    for key, e, b, i in SUB_CONFIGS:
        if (e, b, i) == (expt, batch, img_size):
            return key
    return "%s_bs%d_%d" % (expt, batch, img_size)


def measure(F, expt, batch, img_size, device, rank, world, steps, warmup, reps, use_timer, force_sync=False,
            graph=False):
    """One configuration, timed as the contract says: ``warmup`` untimed cycles, then ``reps`` regions of exactly
    ``steps`` optimizer cycles, each bracketed by barrier + synchronize, max over ranks, the median region reported;
    per-launch HIP events on the launch stream (every 4th cycle of the first region) give the roofline of the dominant
    kernel.  With world > 1 the trainer carries the gradient exchange (ddp.GradSync) and the record its overlap."""
    import torch
    module, trainer = build_trainer(expt, batch, device, world, force_sync, img_size, graph)
    data = synthetic_batch(batch, device, rank, img_size)
    timer = F.KernelTimer()
    for _ in range(8 if graph else 2):          # graph mode: eager warm-up + capture of both optimizer steps
        trainer.step(data)
    trainer.finish()
    if use_timer and not graph:                 # per-launch events cannot be recorded inside a replayed graph
        F.set_kernel_timer(timer)
    sync = getattr(trainer, "grad_sync", None)
    # the overlap report brackets every wait for a gradient bucket with two timing events on the compute stream -- two
    # more marker packets per bucket, each a ~10-20 us bubble: it is collected in OVERLAP_CYCLES extra cycles behind
    # the timed region, not inside it
    want_overlap = sync is not None and sync.measure
    if want_overlap:
        sync.measure = False
    times, per_rank, dt = timed_pairs(trainer, data, steps, warmup, world, timer, reps)
    F.set_kernel_timer(None)
    torch.cuda.synchronize()
    overlap_cycles = 0
    if want_overlap:
        sync.measure = True
        sync.exposed_wait_ms()
        overlap_cycles = min(steps, OVERLAP_CYCLES)
        for _ in range(overlap_cycles * len(trainer.order)):
            trainer.step(data)
        trainer.finish()
        torch.cuda.synchronize()
    ms = dt / steps * 1e3
    per_cycle = len(trainer.order)              # batches per optimizer cycle: 2 (dc_gan, wgan_gp), 6 (wgan), 3 (hologan)
    rec = {"workload": "%s synthetic %dx%d bs=%d/GPU, one optimizer cycle of %d batches (Lightning alternation + "
                       "toggle), reference optimizer, fp32%s"
                       % (expt, img_size, img_size, batch, per_cycle,
                          " (EXT-128: stride-2 extension, not parity-pinned)" if expt == "hologan" and img_size == 128 else ""),
           "value": round(per_cycle * batch * world * steps / dt, 1), "unit": "images/s", "ms_per_step": round(ms, 3),
           "n_gpus": world, "steps": steps, "batches_per_step": per_cycle,
           "ms_per_step_each": [round(t / steps * 1e3, 3) for t in times],
           "per_rank_ms_per_step": [round(t / steps * 1e3, 3) for t in per_rank]}
    if sync is not None:
        rec["grad_exchange"] = {"buckets": [[(e - s0) * 4 for s0, e, _, _ in fg.buckets] for fg in sync.flats],
                                "deferred_tail_buckets": [sorted(fg.tail_buckets) for fg in sync.flats],
                                "per_layer_gates": list(sync.lazy), **sync.stats}
        if sync.measure:
            # how much of the exchange was EXPOSED on this rank: time the compute stream sat behind a gradient bucket
            # that had not been reduced yet (events around every wait), per optimizer cycle of the timed region
            w = sync.exposed_wait_ms()
            cycles = max(overlap_cycles, 1)
            rec["grad_exchange"]["overlap"] = {
                "exposed_wait_ms_per_step": {"discriminator": round(w["discriminator"] / cycles, 4),
                                             "generator": round(w["generator"] / cycles, 4)},
                "waits_per_step": round(w["waits"] / cycles, 2), "rank": rank,
                "measured_cycles": overlap_cycles,
                "note": "sum of (wait end - wait start) on the compute stream over extra cycles behind the timed region; "
                        "0 = fully hidden behind compute"}
        if hasattr(sync, "cu_budget"):
            rec["grad_exchange"]["cu_budget"] = sync.cu_budget
    fl = flop_per_cycle(expt, batch, img_size)
    if rank == 0:
        roof = roofline_of(timer, times[0] / steps * 1e3, steps, fl, config_key_of(expt, batch, img_size))
        if roof is not None:
            if roof["whole_step"] is not None:      # the whole step is priced on the reported (median) time
                a = fl / (ms * 1e-3) / 1e12
                roof["whole_step"].update(achieved=round(a, 2), frac=round(a / PEAK_FP32_MFMA_TFLOPS, 4))
            rec["roofline"] = roof
        elif fl == fl:
            a = fl / (ms * 1e-3) / 1e12
            rec["roofline"] = {"whole_step": {"flop_per_step": fl, "achieved": round(a, 2),
                                              "frac": round(a / PEAK_FP32_MFMA_TFLOPS, 4)}}
    if sync is not None:
        sync.close()
    del trainer, module, data
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def fid50k_record(device, n_samples=50000, batch=250):
    """BASELINE.json's metric also names FID@50k.  FID *numbers* need the reference's weight file (a URL; no network),
    but the WORK of a 50 000-sample FID pass is well defined and is timed here on the HIP path: the eval-mode
    generator sweep over fixed latents (core/callback_inception_metrics.py:183-198: clamp to [0, 1], x 255 -> integer ->
    image file -> ToTensor: the uint8 round trip is done on the device) and the 2048-d InceptionV3 pool features of
    every image (:204-222; bilinear resize to 299 x 299, the FID-patched torchvision network, pinned to the reference's
    own code by tests/golden/inception_ref.npz) -- random-init weights of the right architecture, as for the training
    benchmark.  Frechet distance / KID on the 50 000 x 2048 activations are host-side scipy arithmetic (eval.py) and
    are not part of the timed region.  ``roofline.whole_step`` prices generator + Inception convolution FLOP per second
    against the fp32 MFMA peak."""
    import torch
    from lightning_gan_zoo_amd import inception as I
    from lightning_gan_zoo_amd.config import locate, make_cfg
    cfg = make_cfg("dc_gan", batch_size=batch)
    torch.manual_seed(42)
    module = locate(cfg.model.lm["_target_"])(cfg, None).to(device).eval()
    net = I.FIDInceptionV3().to(device)
    g = torch.Generator().manual_seed(7)
    z_all = torch.randn(n_samples, cfg.model.noise_dim, generator=g)
    feats = torch.empty(n_samples, 2048, device=device)

    @torch.no_grad()
    def sweep(lo, hi):
        for i in range(lo, hi, batch):
            z = z_all[i:i + batch].to(device, non_blocking=True)
            x = torch.clamp(module.generator(z), 0, 1)
            x = (x * 255).to(torch.int32).to(torch.float32) / 255           # PNG round trip: truncation to 8 bits
            feats[i:i + len(z)] = net(x)

    sweep(0, 2 * batch)                      # warm-up: allocator, folded weights
    torch.cuda.synchronize()
    I.FLOPS = [0.0]
    sweep(0, batch)
    inception_flop_per_image = I.FLOPS[0] / batch
    I.FLOPS = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep(0, n_samples)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g_flop = 2 * 410_583_040                 # generator forward, 64x64
    flop = (inception_flop_per_image + g_flop) * n_samples
    ok = bool(torch.isfinite(feats).all())
    del module, net, feats
    torch.cuda.empty_cache()
    return {"workload": "FID@50k feature pass: DCGAN generator sweep (eval mode) + InceptionV3 pool3 features of %d "
                        "images at 299x299, batches of %d, fp32, random-init weights" % (n_samples, batch),
            "seconds": round(dt, 3), "value": round(n_samples / dt, 1), "unit": "images/s", "finite": ok,
            "flop_per_image": {"inception": inception_flop_per_image, "generator": g_flop},
            "roofline": {"whole_step": {"flop_per_step": flop, "achieved": round(flop / dt / 1e12, 2),
                                        "frac": round(flop / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}}}


class single_rank_rccl:
    """A one-rank RCCL communicator with GZ_DDP_ALWAYS_REDUCE (ddp.GradSync then really issues its all-reduces) and
    GZ_DDP_MEASURE (events around every wait for a bucket): the data-parallel code path on one GPU."""
    KEYS = ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "GZ_DDP_ALWAYS_REDUCE", "GZ_DDP_MEASURE")

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        import torch.distributed as dist
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        self.saved = {k: os.environ.get(k) for k in self.KEYS}
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          GZ_DDP_ALWAYS_REDUCE="1", GZ_DDP_MEASURE="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=self.device)
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        dist.destroy_process_group()
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False


def gradsync_w1_records(F, args, device, use_timer, out, head_key):
    """dc_gan bs 128 / bs 512, wgan_gp bs 256 (config 3) and hologan EXT-128 bs 64 (config 5's per-GPU workload) under
    ddp.GradSync over RCCL at world size 1 (GZ_DDP_ALWAYS_REDUCE: the collective is really issued).  The numbers price
    the data-parallel machinery itself; an N > 1 run adds the wire time."""
    recs = {}
    cfgs = {k: (e, b, i) for k, e, b, i in SUB_CONFIGS}
    with single_rank_rccl(device):
        for key in GRADSYNC_W1_CONFIGS:
            expt, bs, img = cfgs[key]
            steps = args.steps if expt == "dc_gan" else min(args.steps, args.sub_steps)
            rec = measure(F, expt, bs, img, device, 0, 1, steps, args.warmup, args.reps, use_timer, force_sync=True)
            plain = out if key == head_key else out["sub_configs"].get(key)
            if plain is not None:
                rec["vs_plain"] = round(rec["ms_per_step"] / plain["ms_per_step"], 4)
                rec["plain_ms_per_step"] = plain["ms_per_step"]
            rec["transport"] = "RCCL, single-rank communicator (world 1): code-path cost, no wire time"
            recs[key + "_gradsync_w1"] = rec
    return recs


def compact_record(rec):
    """A sub-configuration record reduced to its numbers (the default line; ``--detail`` prints the full records, and
    tools/profile_r06.sh commits that form as profiles/rNN_bench_line.json)."""
    out = {}
    if "workload" in rec:
        w = rec["workload"]
        cut = w.find(" (Lightning alternation")
        tail = " (EXT-128: stride-2 extension, not parity-pinned)" if "EXT-128" in w else ""
        out["workload"] = (w[:cut] + tail) if cut > 0 else w
    for k in ("value", "unit", "ms_per_step", "ms_per_step_each", "seconds", "finite", "vs_plain", "plain_ms_per_step",
              "transport", "error"):
        if k in rec:
            out[k] = rec[k]
    r = rec.get("roofline")
    if r:
        out["roofline"] = {k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "traffic_stale",
                                             "igemm_share_of_step") if k in r}
        if "whole_step" in r:
            out["roofline"]["whole_step"] = {k: r["whole_step"][k] for k in ("achieved", "frac") if k in r["whole_step"]}
    g = rec.get("grad_exchange")
    if g:
        out["grad_exchange"] = {k: g[k] for k in ("buckets", "buckets_from_hooks", "buckets_after_backward",
                                                  "buckets_deferred_tail") if k in g}
        ov = g.get("overlap") or {}
        for k in ("exposed_wait_ms_per_step", "waits_per_step"):
            if k in ov:
                out["grad_exchange"][k] = ov[k]
    return out


def run_rank(args):
    import torch
    import torch.distributed as dist
    # stdout carries exactly one line, the JSON result: library chatter written to file descriptor 1 (RCCL prints its
    # version banner there when the process group is torn down) is sent to stderr instead
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # GZ_REHEARSE_ONE_GPU=1: every rank drives cuda:0 and the collectives run on gloo (RCCL refuses two ranks on one
    # device).  It exists to run the whole N > 1 code path -- rank spawn, gradient hooks on HIP tensors, deferred
    # optimizer steps, barrier / max-over-ranks timing, the JSON line -- on a 1-GPU box; its number is meaningless and
    # the line says so ("rehearsal").
    rehearsal = bool(os.environ.get("GZ_REHEARSE_ONE_GPU")) and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from lightning_gan_zoo_amd import functional as F
    if world > 1:
        os.environ.setdefault("GZ_DDP_MEASURE", "1")      # GradSync brackets its waits for gradient buckets with events

    # the host side of the step is launch-only; a big OpenMP team only burns the container's CPU quota
    torch.set_num_threads(min(8, torch.get_num_threads()))
    use_timer = not args.no_kernel_timer
    if args.force_grad_sync and world == 1:       # the headline itself on the data-parallel code path (profiling runs)
        with single_rank_rccl(device):
            head = measure(F, args.expt, args.batch, args.img_size, device, rank, world, args.steps, args.warmup,
                           args.reps, use_timer, True, args.graph)
    else:
        head = measure(F, args.expt, args.batch, args.img_size, device, rank, world, args.steps, args.warmup, args.reps,
                       use_timer, args.force_grad_sync, args.graph)
    head_key = config_key_of(args.expt, args.batch, args.img_size)

    out = {
        "metric": "images/sec (G+D step) at %dx%d bs=%d/GPU" % (args.img_size, args.img_size, args.batch),
        "value": head["value"],
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": head["workload"], "key": head_key,
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "images_counted": "%d*bs*n_gpus per step (every batch of the cycle)" % head["batches_per_step"]},
        "repetitions": {"count": args.reps, "ms_per_step_each": head["ms_per_step_each"], "reported": "median"},
        "per_rank_ms_per_step": head["per_rank_ms_per_step"],
    }
    if world > 1:
        out["rccl_ranks"] = dist.get_world_size()
        if rehearsal:
            out["rehearsal"] = "all ranks on cuda:0, gloo transport: code-path check only, not a measurement"
    if "grad_exchange" in head:
        out["grad_exchange"] = head["grad_exchange"]
    if "roofline" in head:
        out["roofline"] = head["roofline"]

    default_run = args.expt == "dc_gan" and args.img_size == 64 and not args.no_bs128
    if default_run:
        # the other BASELINE configurations.  Every world size: dc_gan bs 512 / GPU (configs 2 and 4, the north star's
        # roofline batch), timed on ALL ranks exactly like the headline.  One GPU only: config 3 (wgan_gp bs 256),
        # config 5's per-GPU workload (hologan bs 64; 64x64 parity-pinned and EXT-128), wgan bs 512.
        out["sub_configs"] = {}
        for key, expt, bs, img in SUB_CONFIGS:
            if key == head_key or (world > 1 and key not in MULTI_GPU_SUB_CONFIGS):
                continue
            steps = args.steps if expt == "dc_gan" else min(args.steps, args.sub_steps)
            out["sub_configs"][key] = measure(F, expt, bs, img, device, rank, world, steps, args.warmup, args.reps,
                                              use_timer)
        if world == 1 and not args.force_grad_sync and not args.no_gradsync_w1:
            # The data-parallel CODE PATH on this one GPU: the same dc_gan trainer under
            # ddp.GradSync with a single-rank RCCL communicator -- flat gradient buffers, per-parameter hooks, bucketed
            # all-reduce calls on RCCL's stream, per-bucket optimizer steps at the layer gates, the deferred tail -- so
            # that every driver run times it next to the plain trainer (``vs_plain`` = this / plain ms per pair).
            try:
                out["sub_configs"].update(gradsync_w1_records(F, args, device, use_timer, out, head_key))
            except Exception as e:  # noqa: BLE001 -- a broken RCCL install must not take the headline number with it
                out["sub_configs"]["dc_gan_bs128_gradsync_w1"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and args.fid_samples > 0:
            try:
                out["sub_configs"]["fid50k_features"] = fid50k_record(device, args.fid_samples)
            except Exception as e:  # noqa: BLE001
                out["sub_configs"]["fid50k_features"] = {"error": "%s: %s" % (type(e).__name__, e)}
        big = out["sub_configs"].get("dc_gan_bs512")
        if big is not None and rank == 0:
            w = (big.get("roofline") or {}).get("whole_step") or {}
            out["config"]["workload"] += ("; the north star's roofline batch bs=512/GPU in the same run: %.0f images/s, "
                                          "%.3f ms per pair, whole step %.3f of the fp32 MFMA peak "
                                          "(sub_configs.dc_gan_bs512)" % (big["value"], big["ms_per_step"],
                                                                          w.get("frac", float("nan"))))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.expt == "dc_gan":
            out["cpu_baseline"] = cpu_baseline(128, 12.0)
            out["cpu_baseline"]["bs64"] = cpu_baseline(64, 7.0)        # BASELINE config 1's batch
        subs = out.pop("sub_configs", None)
        if subs is not None:              # last in the line: a record that keeps only the tail of stdout keeps these
            # (N > 1 runs carry three records: full, with per-rank times and the overlap report)
            out["sub_configs"] = subs if (args.detail or world > 1) else {k: compact_record(v) for k, v in subs.items()}
        print(json.dumps(out), file=result_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting what actually runs (n_gpus=%d)"
              % (args.gpus, world, world), file=sys.stderr)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
