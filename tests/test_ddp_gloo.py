"""Multi-rank path on CPU (gloo, world_size 2): the flat-buffer gradient exchange with deferred
optimizer step equals "average the per-rank gradients, then step" (torch DDP's semantics), and
leaves every rank with identical parameters.  Uses the CPU oracle networks as the model."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, overlap, ret, bucket_bytes=16 << 20, expt="dc_gan", cycles=2, product_layout=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import FixedNoise, fill_closed_form, synthetic_noise, synthetic_real
        from lightning_gan_zoo_amd.config import locate, make_cfg
        from lightning_gan_zoo_amd.ddp import GradSync
        from lightning_gan_zoo_amd.harness import Trainer, optimizer_schedule, toggle_optimizer
        import numpy as np

        def build():
            cfg = make_cfg(expt, module_root="oracle.reference_cpu", batch_size=4, features=8, noise_dim=16)
            torch.manual_seed(42)      # same seed on every rank, as run_network.py:27
            step = locate(cfg.model.lm["_target_"])(cfg, None)
            fill_closed_form(step.generator, 1)
            fill_closed_form(step.discriminator, 2)
            return step

        probe = build()
        order = optimizer_schedule([o["frequency"] for o in probe.configure_optimizers()])
        nsteps = cycles * len(order)          # dc_gan / wgan_gp: D G; wgan: 5 x D, G; hologan: D G G
        del probe
        labels = torch.zeros(4, dtype=torch.int64)
        batches = [(synthetic_real(4, seed=10 * k + rank), labels) for k in range(nsteps)]
        noises = [synthetic_noise(4, 16, 50 + 10 * k + rank, uniform=expt == "hologan") for k in range(nsteps)]
        alphas = [torch.rand(4, 1, 1, 1, generator=torch.Generator().manual_seed(900 + 10 * k + rank))
                  for k in range(nsteps)]

        def per_step_inputs(step, k):
            step.noise_distn = FixedNoise(noises[k])
            if expt == "wgan_gp":
                step.gp_alpha = alphas[k]          # the reference draws it from the host generator (utils.py:41)
            np.random.seed(7000 + 10 * k + rank)   # HoloGAN's views (numpy's global generator)

        # A: the product harness with GradSync
        a = build()
        if expt == "wgan":
            # the product's WGAN step class declares that its training_step mutates the critic before the critic's
            # forward (the weight clamp); the oracle class that stands in for it here gets the same declaration
            from lightning_gan_zoo_amd.core.lightning_module import WGAN as ProductWGAN
            assert ProductWGAN.mutates_discriminator_before_forward is True
            a.mutates_discriminator_before_forward = True
        kw = {}
        if product_layout:
            # the product's HoloGAN modules declare per-layer gates, their gradients' arrival order and their own
            # deferred tail (core/models/hologan_generator.py); the oracle's plain torch modules (same attribute names)
            # borrow those declarations, their parameter-owning children are gated by forward-pre hooks
            import types
            from lightning_gan_zoo_amd.core.models.hologan_generator import Generator as ProductG
            for name in ("_zmaps", "grad_arrival_order", "deferred_tail_parameters"):
                setattr(a.generator, name, types.MethodType(getattr(ProductG, name), a.generator))
            a.generator.gates_parameters = a.discriminator.gates_parameters = True
            kw = dict(defer_tail=True, tail_min_bytes=256)
        sync = GradSync(a, overlap=overlap, bucket_bytes=bucket_bytes, **kw)
        if product_layout:
            sync.trace = []
        tr = Trainer(a, grad_sync=sync)
        assert tr.order == order
        for k in range(nsteps):
            per_step_inputs(a, k)
            tr.step(batches[k])
        tr.finish()
        layout = ([len(fg.buckets) for fg in sync.flats], dict(sync.stats))
        if product_layout:
            fg = sync.flats[1]
            names = {id(p): n for n, p in a.generator.named_parameters()}
            layout += (list(sync.trace), sorted(fg.tail_buckets),
                       [[names[id(p)] for p in fg.bucket_params(b)] for b in range(len(fg.buckets))])

        # B: explicit DDP semantics -- all-reduce(mean) every gradient, then step
        b = build()
        opts = b.configure_optimizers()
        for k in range(nsteps):
            idx = order[k % len(order)]
            toggle_optimizer(b, idx)
            per_step_inputs(b, k)
            # torch DDP broadcasts rank 0's buffers at the top of every forward (broadcast_buffers=True, its default);
            # GradSync's default does so for the buffers training reads: the spectral-norm vectors (HoloGAN)
            for name, buf in b.named_buffers():
                if name.endswith(("weight_u", "weight_v")):
                    dist.broadcast(buf, src=0)
            b.training_step(batches[k], k, idx).backward()
            net = b.discriminator if idx == 0 else b.generator
            for p in net.parameters():
                if p.grad is None:       # parameters without a gradient in this step (none in these experiments)
                    continue
                dist.all_reduce(p.grad)
                p.grad /= world
            opts[idx]["optimizer"].step()
            opts[idx]["optimizer"].zero_grad()

        worst = 0.0
        for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            worst = max(worst, float((p - q).abs().max() / q.abs().max().clamp_min(1e-12)))
        # identical parameters on every rank
        flat = torch.cat([p.detach().reshape(-1) for p in a.parameters()])
        other = flat.clone()
        dist.broadcast(other, src=0)
        same = bool(torch.equal(flat, other))
        ret[rank] = (worst, same, layout)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_gradsync_equals_ddp_mean_then_step(overlap):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, ret), nprocs=world, join=True)
    for rank in range(world):
        worst, same, (nbuckets, stats) = ret[rank]
        assert same, "ranks diverged"
        assert worst < 1e-6, worst
        assert nbuckets == [1, 1]                    # the 8-feature nets fit one 16 MB bucket each
        assert stats["buckets_from_hooks"] == (4 if overlap else 0)


@pytest.mark.parametrize("expt", ["wgan", "wgan_gp", "hologan"])
def test_gradsync_other_experiments_equal_ddp_mean_then_step(expt):
    """The schedules DCGAN does not exercise, two optimizer cycles each on two ranks, small buckets (several
    all-reduces per backward, all issued from the gradient hooks):
      * wgan (5 critic batches per generator batch, conf/expt/wgan.yaml:22-23; the weight clamp at the top of EVERY
        training_step mutates the critic, so its pending exchange + RMSprop step must land before the clamp:
        GradSync's early finalize, `mutates_discriminator_before_forward`);
      * wgan_gp (the penalty's double backward reaches the critic's parameters twice; `fake` is not detached in the
        penalty, the frozen generator must receive no gradient and issue no bucket);
      * hologan (1 : 2 -- two consecutive generator steps: the pending step of the SAME network lands before its next
        backward).
    Parameters after the run equal explicit DDP semantics (all-reduce(mean) every gradient, then step) and are
    identical on both ranks."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, ret, 2048, expt, 2), nprocs=world, join=True)
    for rank in range(world):
        worst, same, (nbuckets, stats) = ret[rank]
        assert same, "ranks diverged"
        assert worst < 1e-6, (expt, worst)
        d_steps, g_steps = {"wgan": (10, 2), "wgan_gp": (2, 2), "hologan": (2, 4)}[expt]
        assert nbuckets[0] >= 2 and nbuckets[1] >= 2, nbuckets
        assert stats["buckets_from_hooks"] + stats["buckets_after_backward"] == d_steps * nbuckets[0] + g_steps * nbuckets[1]
        # every bucket whose parameters all received a gradient came from a hook; HoloGAN has parameters with no
        # gradient path in some steps (the latent head in G steps): those buckets go out after backward
        if expt != "hologan":
            assert stats["buckets_after_backward"] == 0, stats


def test_hologan_generator_hands_its_exchange_to_the_next_generator_step():
    """BASELINE config 5 under data parallelism (conf/expt/hologan.yaml:16-17: D, G, G).  With the HoloGAN generator's
    own declarations -- per-layer gates, gradient arrival order (the five ZMapping layers last), deferred tail = block3
    / block4 -- a generator pass is NOT landed as a whole at the top of the generator step that follows it: the main
    buckets are waited for at the first layers' gates, the tail buckets (issued last) at block3 / block4's gates, after
    earlier generator layers have been queued; and the result still equals all-reduce(mean)-then-step on both ranks."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, ret, 2048, "hologan", 2, True), nprocs=world, join=True)
    for rank in range(world):
        worst, same, (nbuckets, stats, trace, tail_buckets, names) = ret[rank]
        assert same, "ranks diverged"
        assert worst < 1e-6, worst
        # layout: the tail holds exactly the two 2-D blocks' weights and sits behind everything else; the ZMapping
        # layers and the constant close the main section
        tail_names = [n for b in tail_buckets for n in names[b]]
        assert tail_names == ["block4.convTranspose.weight", "block3.convTranspose.weight"], tail_names
        assert min(tail_buckets) == nbuckets[1] - len(tail_buckets)
        main_names = [n for b in range(min(tail_buckets)) for n in names[b]]
        assert main_names[0].startswith("final_layer") and main_names.index("x") < main_names.index("zMapping.linear1.weight")
        assert all(".zMapping." in n or n.startswith("zMapping.") for n in main_names[main_names.index("x") + 1:])
        # every generator pass: main buckets, then the (on the GPU: postponed) tail, issued last
        g_issue_runs, run = [], []
        for ev, i, b in trace:
            if ev == "issue" and i == 1:
                run.append(b)
            elif run and not (ev == "deferred" and i == 1):
                g_issue_runs.append(run)
                run = []
        if run:
            g_issue_runs.append(run)
        g_issue_runs = [r for r in g_issue_runs if len(r) == nbuckets[1]] or g_issue_runs
        assert len(g_issue_runs) >= 4
        for r in g_issue_runs:
            assert r[-len(tail_buckets):] == tail_buckets, (r, tail_buckets)
        # G -> G hand-over: between the last issue of a generator pass and the first issue of the next generator pass
        # with no discriminator pass in between, the tail buckets are waited for AFTER a gate on a main bucket
        handovers = 0
        k_issue = [k for k, t in enumerate(trace) if t[0] == "issue" and t[1] == 1]
        for a_, b_ in zip(k_issue, k_issue[1:]):
            if b_ - a_ < 2:
                continue
            between = trace[a_ + 1:b_]
            if any(t[0] == "issue" and t[1] == 0 for t in between):
                continue                   # a discriminator pass ran in between: that is the G -> D -> G hand-over
            g = [t for t in between if t[1] == 1 and t[0] in ("gate", "wait")]
            if not g:
                continue
            handovers += 1
            assert g[0][0] == "gate" and g[0][2] not in tail_buckets, g[:4]         # nothing is waited for ungated
            first_tail_wait = next(k for k, t in enumerate(g) if t[0] == "wait" and t[2] in tail_buckets)
            assert any(t[0] == "gate" and t[2] in tail_buckets for t in g[:first_tail_wait]), g
            assert any(t[0] == "wait" and t[2] not in tail_buckets for t in g[:first_tail_wait])
        assert handovers >= 2, handovers


def test_gradsync_many_buckets_reduced_from_backward_hooks():
    """Small buckets: each one's all-reduce is issued from the autograd thread as soon as its last gradient has been
    accumulated (so it overlaps the rest of backward); the result still equals mean-then-step and ranks agree."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, ret, 4096), nprocs=world, join=True)
    for rank in range(world):
        worst, same, (nbuckets, stats) = ret[rank]
        assert same and worst < 1e-6, (same, worst)
        assert nbuckets[0] >= 4 and nbuckets[1] >= 4, nbuckets
        # two D steps and two G steps: every bucket of the active network came from a hook, none was left over
        assert stats["buckets_from_hooks"] == 2 * nbuckets[0] + 2 * nbuckets[1] and stats["buckets_after_backward"] == 0


def _ckpt_worker(rank, world, port, ckpt_dir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_harness import run_on_cpu
        from lightning_gan_zoo_amd.ddp import GradSync
        module, trainer, step, _ = run_on_cpu(
            "dc_gan", ["train.features_gen=8", "train.features_disc=8", "model.noise_dim=16", "log_every=1000",
                       "train.batch_size=2", "train.ckpt_dir=" + ckpt_dir, "max_steps=5", "steps_per_epoch=2"],
            sync_factory=GradSync, rank=rank, world=world)
        bufs = torch.cat([b.detach().float().reshape(-1) for b in module.buffers()])
        other = bufs.clone()
        dist.broadcast(other, src=0)
        ret[rank] = (step, bool(torch.equal(bufs, other)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_checkpoint_without_deadlock(tmp_path):
    """Round-1 advisor finding: ``sync_buffers`` (a broadcast per norm buffer) sat inside the rank-0 guard, so a
    multi-rank run that saved a checkpoint hung.  Now every rank joins the broadcast, rank 0 writes, all meet at a
    barrier; the per-rank BatchNorm buffers equal rank 0's afterwards and the file is Lightning-shaped."""
    world = 2
    ck = str(tmp_path / "ck")
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_ckpt_worker, args=(world, _free_port(), ck, ret), nprocs=world, join=True)
    assert [ret[r] for r in range(world)] == [(5, True), (5, True)]
    assert os.listdir(ck) == ["step=5.ckpt"]
    blob = torch.load(os.path.join(ck, "step=5.ckpt"), weights_only=False)
    assert blob["global_step"] == 5 and blob["epoch"] == 2


def _buffer_worker(rank, world, port, broadcast, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import FixedNoise, synthetic_noise, synthetic_real
        from lightning_gan_zoo_amd.config import locate, make_cfg
        from lightning_gan_zoo_amd.ddp import GradSync
        from lightning_gan_zoo_amd.harness import Trainer
        cfg = make_cfg("dc_gan", module_root="oracle.reference_cpu", batch_size=4, features=8, noise_dim=16)
        torch.manual_seed(42)
        step = locate(cfg.model.lm["_target_"])(cfg, None)
        sync = GradSync(step, broadcast_buffers=broadcast)
        tr = Trainer(step, grad_sync=sync)
        labels = torch.zeros(4, dtype=torch.int64)
        for k in range(4):
            step.noise_distn = FixedNoise(synthetic_noise(4, 16, 50 + 10 * k + rank))
            tr.step((synthetic_real(4, seed=10 * k + rank), labels))         # every rank its own shard
        tr.finish()

        def same_as_rank0():
            mine = torch.cat([b.detach().double().reshape(-1) for b in step.buffers()])
            ref = mine.clone()
            dist.broadcast(ref, src=0)
            return float((mine - ref).abs().max())

        before = same_as_rank0()
        if broadcast:                    # DDP's broadcast happens at the TOP of a step: the last step's own update is
            sync._broadcast_all_buffers()    # still local; one more forward (or a checkpoint) would broadcast it
        sync.sync_buffers()
        after = same_as_rank0()
        params = torch.cat([p.detach().reshape(-1) for p in step.parameters()])
        other = params.clone()
        dist.broadcast(other, src=0)
        ret[rank] = (before, after, bool(torch.equal(params, other)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("broadcast", [False, True])
def test_batchnorm_buffers_under_data_parallelism(broadcast):
    """States the one deviation from torch DDP: DDP broadcasts rank 0's buffers at the top of every
    forward; GradSync keeps the BatchNorm running statistics per rank during training (training-mode BatchNorm never
    reads them: parameters stay identical on all ranks either way) and broadcasts rank 0's at every checkpoint.
    ``broadcast_buffers=True`` reproduces DDP: at most the last step's own update separates a rank from rank 0."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_buffer_worker, args=(world, _free_port(), broadcast, ret), nprocs=world, join=True)
    assert ret[0][2] and ret[1][2], "parameters diverged"
    assert ret[0][0] == 0.0 and ret[0][1] == 0.0                 # rank 0 is the reference
    assert ret[1][1] == 0.0                                      # after sync_buffers (what a checkpoint does): equal
    drift = ret[1][0]
    assert drift > 0.0                                           # rank 1's own statistics differ from rank 0's ...
    if broadcast:
        # ... but only by the last step's momentum-0.1 update of rank 1's own batch statistics on top of rank 0's
        # buffers, not by four steps of independent history
        mp_ret = mp.get_context("spawn").Manager().dict()
        mp.spawn(_buffer_worker, args=(world, _free_port(), False, mp_ret), nprocs=world, join=True)
        assert drift < mp_ret[1][0], (drift, mp_ret[1][0])


def test_optimizer_schedule_follows_frequencies():
    from lightning_gan_zoo_amd.harness import optimizer_schedule
    assert optimizer_schedule([1, 1]) == [0, 1]
    assert optimizer_schedule([5, 1]) == [0, 0, 0, 0, 0, 1]          # conf/expt/wgan.yaml:22-23
    assert optimizer_schedule([1, 2]) == [0, 1, 1]                   # conf/expt/hologan.yaml:16-17


def test_flat_gradient_views_start_on_16_byte_boundaries():
    """The sink kernels read and write float4: every ``p.grad`` view of the flat exchange buffer starts at a multiple of
    four floats whatever the parameter sizes in front of it (HoloGAN's 1-element logit bias, 3-element image bias), the
    views do not overlap and the padding stays zero through a fill."""
    import torch
    from lightning_gan_zoo_amd.ddp import _FlatGrads
    params = [torch.nn.Parameter(torch.zeros(*shape)) for shape in ((3,), (64, 3, 3, 3), (1,), (1, 8192), (5, 7), (128,))]
    fg = _FlatGrads(params, bucket_bytes=4096)
    assert all(off % 4 == 0 for off in fg.offsets)
    assert all(p.grad.data_ptr() == fg.flat.data_ptr() + 4 * off for p, off in zip(fg.params, fg.offsets))
    for k, p in enumerate(fg.params):
        p.grad.fill_(k + 1.0)
    covered = torch.zeros_like(fg.flat)
    for k, (p, off) in enumerate(zip(fg.params, fg.offsets)):
        assert float(covered[off:off + p.numel()].abs().sum()) == 0.0           # no overlap
        covered[off:off + p.numel()] = 1.0
        assert torch.equal(fg.flat[off:off + p.numel()], torch.full((p.numel(),), k + 1.0))
    assert float(fg.flat[covered == 0].abs().sum()) == 0.0                      # padding untouched
    # the buckets tile [0, end of the last parameter) without gaps
    assert fg.buckets[0][0] == 0 and all(a[1] == b[0] for a, b in zip(fg.buckets, fg.buckets[1:]))
    assert fg.buckets[-1][1] == fg.offsets[-1] + fg.params[-1].numel()
