"""Host logic of harness.Trainer / ddp.GradSync that needs no GPU (the CPU oracle networks are the model):
gradient accumulation with Lightning's semantics (reference run_network.py:61-68), the autograd behaviour the
gradient sinks rely on, per-bucket landing of the gradient exchange with per-layer gates (world size 2, gloo)."""
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


def test_post_accumulate_hook_fires_once_for_none_gradients():
    """What functional's gradient sinks + ddp.GradSync rely on (functional/_base.py, "WHEN a sunk parameter's gradient is
    complete"): a parameter's post-accumulate-grad hook runs ONCE per backward pass, after every node that uses the
    parameter has run -- also when those nodes return None for it (the sink took the gradient), and also when part of
    the uses were created by a double backward."""
    fired = []

    class Sunk(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x * w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            fired.append("node")
            return g * w, None                      # the parameter's gradient went elsewhere

    w = torch.nn.Parameter(torch.ones(3))
    w.register_post_accumulate_grad_hook(lambda p: fired.append("hook"))
    x = torch.ones(3, requires_grad=True)
    (Sunk.apply(x, w) + Sunk.apply(2 * x, w)).sum().backward()
    assert fired == ["node", "node", "hook"]        # once, after BOTH uses
    assert w.grad is None

    # a double-backward graph: the hook still waits for every contribution of the final backward pass
    fired.clear()
    v = torch.nn.Parameter(torch.full((3,), 2.0))
    v.register_post_accumulate_grad_hook(lambda p: fired.append("hook"))
    y = (x * v * v).sum()
    (gx,) = torch.autograd.grad(y, x, create_graph=True)          # no hook: autograd.grad does not accumulate
    assert fired == []
    (gx.pow(2).sum() + (x * v).sum()).backward()
    assert fired == ["hook"] and v.grad is not None


def test_forward_of_a_function_runs_without_grad_mode():
    """Why round 4's use counter inside Function.forward never counted: grad mode is off there."""
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            seen.append(torch.is_grad_enabled())
            return x * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    Probe.apply(torch.ones(2, requires_grad=True)).sum().backward()
    assert seen == [False]


def test_accumulation_factor_follows_lightnings_scheduler():
    from lightning_gan_zoo_amd.harness import accumulation_factor
    assert accumulation_factor(1, 7) == 1 and accumulation_factor(4, 0) == 4
    sched = {400: 4}                       # conf/machine/big.yaml: start_epoch 400, accumulation_factor 4
    assert [accumulation_factor(sched, e) for e in (0, 399, 400, 900)] == [1, 1, 4, 4]
    assert accumulation_factor({2: 2, 5: 8}, 4) == 2 and accumulation_factor({2: 2, 5: 8}, 5) == 8


def _build(expt="dc_gan", bs=4):
    from helpers import fill_closed_form
    from lightning_gan_zoo_amd.config import locate, make_cfg
    cfg = make_cfg(expt, module_root="oracle.reference_cpu", batch_size=bs, features=8, noise_dim=16)
    torch.manual_seed(42)
    step = locate(cfg.model.lm["_target_"])(cfg, None)
    fill_closed_form(step.generator, 1)
    fill_closed_form(step.discriminator, 2)
    return step


def _explicit_accumulation(step, batches, noises, factor, per_epoch, world=1):
    """Lightning 1.1's run_training_batch, spelled out: optimizer by running batch count, loss / factor, optimizer step
    + zero_grad on batches with (index in epoch + 1) % factor == 0 or the epoch's last batch; under DDP the gradients
    are averaged on those batches only (block_ddp_sync_behaviour on the others)."""
    from helpers import FixedNoise
    from lightning_gan_zoo_amd.harness import optimizer_schedule, toggle_optimizer
    opts = step.configure_optimizers()
    order = optimizer_schedule([o["frequency"] for o in opts])
    for k, batch in enumerate(batches):
        idx = order[k % len(order)]
        toggle_optimizer(step, idx)
        step.noise_distn = FixedNoise(noises[k])
        (step.training_step(batch, k, idx) / factor).backward()
        in_epoch = k % per_epoch
        if (in_epoch + 1) % factor == 0 or in_epoch + 1 == per_epoch:
            net = step.discriminator if idx == 0 else step.generator
            if world > 1:
                for p in net.parameters():
                    if p.grad is not None:
                        dist.all_reduce(p.grad)
                        p.grad /= world
            opts[idx]["optimizer"].step()
            opts[idx]["optimizer"].zero_grad()
    return step


def _run_trainer(step, batches, noises, factor, per_epoch, sync=None):
    from helpers import FixedNoise
    from lightning_gan_zoo_amd.harness import Trainer
    tr = Trainer(step, grad_sync=sync, accumulate_grad_batches=factor)
    for k, batch in enumerate(batches):
        step.noise_distn = FixedNoise(noises[k])
        tr.step(batch, last_in_epoch=(k % per_epoch) + 1 == per_epoch)
        if (k % per_epoch) + 1 == per_epoch:
            tr.end_epoch()
    tr.finish()
    return step


def _worst(a, b):
    return max(float((p.detach() - q.detach()).abs().max() / q.detach().abs().max().clamp_min(1e-12))
               for p, q in zip(a.parameters(), b.parameters()))


@pytest.mark.parametrize("factor,per_epoch", [(2, 5), (3, 4)])
def test_trainer_accumulates_gradients_like_lightning(factor, per_epoch):
    """harness.Trainer(accumulate_grad_batches=k) against the spelled-out loop: with the 1 : 1 alternation of dc_gan
    and k = 2 the discriminator only steps when an epoch's last batch is its own -- that IS the reference harness's
    behaviour, and what is pinned here."""
    from helpers import synthetic_noise, synthetic_real
    torch.set_num_threads(2)
    nb = 2 * per_epoch
    labels = torch.zeros(4, dtype=torch.int64)
    batches = [(synthetic_real(4, seed=k), labels) for k in range(nb)]
    noises = [synthetic_noise(4, 16, 70 + k) for k in range(nb)]
    a = _run_trainer(_build(), batches, noises, factor, per_epoch)
    b = _explicit_accumulation(_build(), batches, noises, factor, per_epoch)
    assert _worst(a, b) < 1e-6
    c = _run_trainer(_build(), batches, noises, 1, per_epoch)
    assert _worst(a, c) > 1e-4                          # (and it is not the un-accumulated run)


def _lazy_worker(rank, world, port, ret, factor):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import synthetic_noise, synthetic_real
        from lightning_gan_zoo_amd.ddp import GradSync
        per_epoch, nb = 4, 8
        labels = torch.zeros(4, dtype=torch.int64)
        batches = [(synthetic_real(4, seed=10 * k + rank), labels) for k in range(nb)]
        noises = [synthetic_noise(4, 16, 50 + 10 * k + rank) for k in range(nb)]
        a = _build()
        # the oracle's networks are plain torch modules whose parameter-owning children are CALLED: GradSync gates
        # them with forward-pre hooks (the product modules announce their parameters with functional.ready instead)
        a.generator.gates_parameters = True
        a.discriminator.gates_parameters = True
        # defer_tail on CPU tensors: no launch can be postponed here (that is the HIP path's gradient sinks), but the
        # tail's parameters are reported after backward, so its buckets travel LAST -- the order the landing logic sees
        sync = GradSync(a, bucket_bytes=4096, defer_tail=True, tail_min_bytes=1024)
        assert sync.lazy == [True, True] and all(len(t) >= 1 for t in sync.tails)
        sync.trace = []
        _run_trainer(a, batches, noises, factor, per_epoch, sync)
        b = _explicit_accumulation(_build(), batches, noises, factor, per_epoch, world)
        flat = torch.cat([p.detach().reshape(-1) for p in a.parameters()])
        other = flat.clone()
        dist.broadcast(other, src=0)
        ret[rank] = (_worst(a, b), bool(torch.equal(flat, other)), list(sync.trace),
                     [len(fg.buckets) for fg in sync.flats], dict(sync.stats), [sorted(fg.tail_buckets) for fg in sync.flats])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("factor", [1, 2])
def test_gradsync_lands_bucket_by_bucket_at_the_layer_gates(factor):
    """World size 2, 4 KB buckets, per-layer gates on: the exchange of a pass is waited for + stepped bucket by bucket
    in issue order, each bucket at the first layer that reads it (never for the whole network at the top of its
    forward), the optimizer runs on parameter subsets, and the result still equals all-reduce(mean)-then-step -- with
    gradient accumulation (factor 2) the all-reduce happens on the stepping batches only."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_lazy_worker, args=(world, _free_port(), ret, factor), nprocs=world, join=True)
    for rank in range(world):
        worst, same, trace, nbuckets, stats, tails = ret[rank]
        assert same, "ranks diverged"
        assert worst < 1e-6, worst
        assert min(nbuckets) >= 4
        issues = [(i, b) for ev, i, b in trace if ev == "issue"]
        waits = [(i, b) for ev, i, b in trace if ev == "wait"]
        steps = [(i, b) for ev, i, b in trace if ev == "step"]
        gates = [(i, b) for ev, i, b in trace if ev == "gate"]
        assert len(issues) == len(waits) == len(steps) and len(gates) >= 2
        assert stats["buckets_after_backward"] == 0
        assert stats["buckets_from_hooks"] + stats["buckets_deferred_tail"] == len(issues)
        assert stats["buckets_deferred_tail"] >= len(issues) // max(nbuckets)          # one tail bucket (or more) per pass
        # per network: waited and stepped in the order issued, a bucket's step after its own wait and before the next
        # pass of that network issues anything (a finalize that covers a whole pass waits for all its buckets and
        # steps them with one optimizer launch; a gated one steps bucket by bucket)
        for net in (0, 1):
            assert [b for i, b in waits if i == net] == [b for i, b in issues if i == net]
            assert [b for i, b in steps if i == net] == [b for i, b in issues if i == net]
            where = {ev: [k for k, t in enumerate(trace) if t[0] == ev and t[1] == net] for ev in ("issue", "wait", "step")}
            for k, (ki, kw, ks) in enumerate(zip(where["issue"], where["wait"], where["step"])):
                assert ki < kw < ks
        bucketwise = sum(1 for k, t in enumerate(trace[:-1]) if t[0] == "wait" and trace[k + 1] == ("step",) + t[1:])
        assert bucketwise >= len(issues) // 2, (bucketwise, len(issues))     # most buckets land one by one
        # laziness: the generator's tail bucket (its LAST layers) is gated by itself, after the gate of an earlier layer
        # has landed the main buckets -- i.e. generator layers are queued between the two waits
        g_gates = [b for i, b in gates if i == 1]
        assert any(b in tails[1] for b in g_gates) and any(b not in tails[1] for b in g_gates), (g_gates, tails)
        k_tail = next(k for k, t in enumerate(trace) if t[0] == "gate" and t[1] == 1 and t[2] in tails[1])
        assert any(t[0] == "gate" and t[1] == 1 and t[2] not in tails[1] for t in trace[:k_tail])
        if factor == 2:
            # 8 batches, 2 epochs of 4, alternation D G D G: stepping batches are 1, 3 (G) in each epoch -> four
            # generator passes exchange, no discriminator pass does
            assert {i for i, _ in issues} == {1} and len(issues) == 4 * nbuckets[1]
