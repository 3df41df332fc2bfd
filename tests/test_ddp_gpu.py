"""The data-parallel gradient path on the GPU with RCCL ("nccl" backend), single rank: flat gradient
buffers + asynchronous all-reduce + deferred optimizer step must give bit-identical parameters to the
plain Trainer (world size 1 => the mean is the identity)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("layout", ["default", "small_buckets"])
@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp", "hologan"])
def test_gradsync_on_rccl_single_rank_matches_plain_trainer(expt, layout):
    """GradSync (gradients in the flat exchange buffer, every contribution through the sinks with beta = 1, one
    all-reduce per bucket) against the plain trainer, four steps, same parameters bit for bit.  hologan covers the
    complete-gradient sources (biases, Linear weights, spectral-norm weight_orig: functional._sink_grad) and parameters
    whose view into the flat buffer is not 16-byte aligned (they stay with autograd)."""
    import numpy as np
    import torch.distributed as dist
    from helpers import FixedNoise, fill_closed_form, synthetic_noise, synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      GZ_DDP_ALWAYS_REDUCE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def build():
            cfg = make_cfg(expt, batch_size=8, features=8, noise_dim=16)
            torch.manual_seed(42)
            m = locate(cfg.model.lm["_target_"])(cfg, None)
            if expt != "hologan":          # (hologan keeps its own initialisation: spectral-norm buffers, zero biases)
                fill_closed_form(m.generator, 1)
                fill_closed_form(m.discriminator, 2)
            return m.cuda()

        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        batches = [(synthetic_real(8, seed=k).cuda(), labels) for k in range(4)]
        noises = [synthetic_noise(8, 16, 40 + k) for k in range(4)]
        results = []
        for use_sync in (True, False):
            m = build()
            # small_buckets: several buckets per network, a deferred tail (round 5: the weight gradients of the last
            # layers are launched after backward and travel last), per-bucket optimizer steps at the layer gates
            kw = {} if layout == "default" else dict(bucket_bytes=32 << 10, tail_min_bytes=1024)
            tr = Trainer(m, grad_sync=GradSync(m, **kw) if use_sync else None)
            torch.manual_seed(7)
            np.random.seed(7)              # (hologan's views, wgan_gp's interpolation weights)
            for k in range(4):
                m.noise_distn = FixedNoise(noises[k])
                tr.step(batches[k])
            tr.finish()
            torch.cuda.synchronize()
            results.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
        assert torch.equal(results[0], results[1])
    finally:
        dist.destroy_process_group()
        os.environ.pop("GZ_DDP_ALWAYS_REDUCE", None)


def test_two_rank_bench_rehearsal_on_one_gpu():
    """``bench.py --gpus 2`` end to end on this 1-GPU box: the parent spawns the ranks, both drive cuda:0 and the
    collectives run on gloo (GZ_REHEARSE_ONE_GPU).  Everything of the N > 1 path except RCCL itself executes on HIP
    tensors: gradient hooks, bucketed exchange, deferred optimizer steps, barrier + max-over-ranks timing, one JSON
    line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GZ_REHEARSE_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--reps", "1", "--no-cpu-baseline", "--no-kernel-timer"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # the headline is the metric string's configuration (BASELINE.json: bs=128/GPU) ...
    assert "bs=128/GPU" in out["metric"] and out["config"]["key"] == "dc_gan_bs128"
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["config"]["global_batch"] == 256 and out["scaling"] == "weak" and "rehearsal" in out
    # ... and the north star's bs=512/GPU is timed on ALL ranks in the same run (a SCALE run yields both curves)
    big = out["sub_configs"]["dc_gan_bs512"]
    # ... and so is BASELINE config 5, the 8-GPU HoloGAN workload (128x128 bs 64 / GPU; round 6)
    holo = out["sub_configs"]["hologan_ext128_bs64"]
    assert set(out["sub_configs"]) == {"dc_gan_bs512", "hologan_ext128_bs64"}
    assert big["n_gpus"] == 2 and len(big["per_rank_ms_per_step"]) == 2
    assert big["value"] > 0 and "bs=512/GPU" in big["workload"]
    assert holo["n_gpus"] == 2 and holo["value"] > 0 and "hologan synthetic 128x128 bs=64/GPU" in holo["workload"]
    assert holo["grad_exchange"]["per_layer_gates"] == [True, True]
    for rec in (out, big, holo):
        ex = rec["grad_exchange"]
        assert ex["buckets_from_hooks"] > 0 and ex["buckets_after_backward"] == 0
        assert ex["buckets_deferred_tail"] > 0          # the generator's last layers: launched + exchanged last
        # the overlap report of the first real multi-GPU run: exposed wait per optimizer cycle and network
        ov = ex["overlap"]
        assert set(ov["exposed_wait_ms_per_step"]) == {"discriminator", "generator"} and ov["waits_per_step"] >= 2
        assert all(v >= 0.0 for v in ov["exposed_wait_ms_per_step"].values())


@pytest.mark.parametrize("expt", ["dc_gan", "hologan"])
def test_two_rank_training_run_rehearsal_on_one_gpu(tmp_path, expt):
    """``python -m torch.distributed.run --nproc-per-node 2 -m lightning_gan_zoo_amd.run_network +expt=dc_gan ...`` on
    one GPU (GZ_REHEARSE_ONE_GPU): sharded synthetic data, gradient exchange from the backward hooks, rank 0 alone
    writes the Lightning-format checkpoint, both ranks leave together."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = str(tmp_path / "ckpt")
    env = dict(os.environ, GZ_REHEARSE_ONE_GPU="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "-m", "lightning_gan_zoo_amd.run_network", "+expt=" + expt,
           "dataset=synthetic", "model.noise_dim=16", "train.batch_size=8", "train.ckpt_dir=" + ck, "log_every=1000",
           "max_steps=4"]
    cmd += (["train.features_gen=8", "train.features_disc=8"] if expt == "dc_gan"
            else ["generator.in_planes=8", "discriminator.out_planes=8"])      # hologan: gradients of every kind of parameter
                                                                               # through the flat exchange buffer
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.listdir(ck) == ["step=4.ckpt"]
    blob = torch.load(os.path.join(ck, "step=4.ckpt"), weights_only=False)
    assert blob["global_step"] == 4
    assert all(torch.isfinite(v).all() for v in blob["state_dict"].values() if v.is_floating_point())


@pytest.mark.skipif(not os.environ.get("GZ_REHEARSE_8"), reason="opt-in (GZ_REHEARSE_8=1): eight processes on one GPU, ~1 min")
def test_eight_rank_rehearsal_at_reference_widths_on_one_gpu(tmp_path):
    """the driver's largest world size on this 1-GPU box -- eight ranks on cuda:0 over gloo, dc_gan
    and hologan at the reference's widths (features 64), DDP's default buffer broadcast on: no deadlock, parameters and
    buffers bit-identical on all ranks after three cycles, every bucket from a hook or the deferred tail, and the
    host's enqueue time per cycle reported next to the GPU's (tools/rehearse_ranks.py; the round's run is
    profiles/r06_rehearse_8ranks.json)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GZ_REHEARSE_ONE_GPU="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = str(tmp_path / "rehearse.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "tools", "rehearse_ranks.py"), "--out", out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.load(open(out))
    assert rep["world"] == 8
    for expt in ("dc_gan", "hologan"):
        assert rep[expt]["identical_on_all_ranks"] and rep[expt]["finite"] and rep[expt]["buckets_after_backward"] == 0


def test_deferred_tail_hides_the_generator_exchange_behind_launches():
    """VERDICT r4 item 1b.  The stacked discriminator pass needs G(z) first, so the generator's gradient exchange used
    to be waited for, as a whole, at the top of every discriminator step.  Now (ddp.py): the weight gradients of the
    generator's LAST layers are launched after everything else has been issued, their bucket travels last, and a bucket
    is only waited for at the first layer that reads it.  Asserted on the event trace of a single-rank RCCL run
    (dc_gan, features 16, 64 KB buckets): in every generator pass the main buckets are issued BEFORE the postponed
    launches run and the tail bucket after them; in the following discriminator step the tail bucket's wait comes after
    the gates of at least two earlier generator layers (their kernels are queued in between); parameters equal the
    plain trainer's bit for bit."""
    import torch.distributed as dist
    from helpers import FixedNoise, fill_closed_form, synthetic_noise, synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      GZ_DDP_ALWAYS_REDUCE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def build():
            cfg = make_cfg("dc_gan", batch_size=8, features=16, noise_dim=16)
            torch.manual_seed(42)
            m = locate(cfg.model.lm["_target_"])(cfg, None)
            fill_closed_form(m.generator, 1)
            fill_closed_form(m.discriminator, 2)
            return m.cuda()

        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        batches = [(synthetic_real(8, seed=k).cuda(), labels) for k in range(6)]
        noises = [synthetic_noise(8, 16, 40 + k) for k in range(6)]
        results, trace, tails = [], None, None
        for use_sync in (True, False):
            m = build()
            sync = GradSync(m, bucket_bytes=64 << 10, tail_min_bytes=1024) if use_sync else None
            if sync is not None:
                sync.trace = trace = []
                tails = [sorted(fg.tail_buckets) for fg in sync.flats]
                assert sync.lazy == [True, True] and tails[1], tails
                n_tail_params = len(sync.tails[1])
            tr = Trainer(m, grad_sync=sync)
            for k in range(6):
                m.noise_distn = FixedNoise(noises[k])
                tr.step(batches[k])
            tr.finish()
            torch.cuda.synchronize()
            results.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
        assert torch.equal(results[0], results[1])
        # generator passes: [issue main ...] deferred [issue tail ...]
        k_def = [k for k, t in enumerate(trace) if t[0] == "deferred" and t[1] == 1]
        assert len(k_def) == 3 and all(trace[k][2] == n_tail_params for k in k_def)
        for k in k_def:
            before = [t for t in trace[:k] if t[0] == "issue" and t[1] == 1]
            after = [t for t in trace[k + 1:k + 1 + len(tails[1])]]
            assert before and before[-1][2] not in tails[1]
            assert [t[0] for t in after] == ["issue"] * len(tails[1]) and all(t[2] in tails[1] for t in after)
        # discriminator steps that follow a generator pass: the tail bucket is waited for at ITS layers' gate
        waits_tail = [k for k, t in enumerate(trace) if t[0] == "wait" and t[1] == 1 and t[2] in tails[1]]
        gated = 0
        for k in waits_tail:
            j = k
            while j >= 0 and not (trace[j][0] == "issue" and trace[j][1] == 1):
                j -= 1                       # back to this pass's last issue
            g_gates = [t for t in trace[j:k] if t[0] == "gate" and t[1] == 1]
            if g_gates:                      # (the run's final flush lands without gates)
                gated += 1
                assert len(g_gates) >= 2 and g_gates[0][2] not in tails[1], g_gates
        assert gated >= 2
    finally:
        dist.destroy_process_group()
        os.environ.pop("GZ_DDP_ALWAYS_REDUCE", None)


def test_hologan_generator_pass_is_landed_by_the_next_generator_steps_gates():
    """BASELINE config 5's schedule is D, G, G (conf/expt/hologan.yaml:16-17).  On the GPU, over
    single-rank RCCL (in_planes 8, 16 KB buckets): HoloGAN's networks gate per layer, the generator's gradients are laid
    out in arrival order with the ZMapping layers last, block3 / block4's weight-gradient launches are postponed and
    their bucket is issued after them; in the generator step that FOLLOWS a generator step nothing is waited for before the
    first gate, the tail bucket's wait comes after a gate on a main bucket, and the parameters equal the plain trainer's
    bit for bit."""
    import numpy as np
    import torch.distributed as dist
    from helpers import FixedNoise, synthetic_noise, synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      GZ_DDP_ALWAYS_REDUCE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def build():
            cfg = make_cfg("hologan", batch_size=8, features=8, noise_dim=16)
            torch.manual_seed(42)
            return locate(cfg.model.lm["_target_"])(cfg, None).cuda()

        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        batches = [(synthetic_real(8, seed=k).cuda(), labels) for k in range(6)]
        noises = [synthetic_noise(8, 16, 40 + k, uniform=True) for k in range(6)]
        results, trace, tails = [], None, None
        for use_sync in (True, False):
            m = build()
            sync = GradSync(m, bucket_bytes=16 << 10, tail_min_bytes=1024) if use_sync else None
            if sync is not None:
                sync.trace = trace = []
                fg = sync.flats[1]
                tails = sorted(fg.tail_buckets)
                names = {id(p): n for n, p in m.generator.named_parameters()}
                assert sync.lazy == [True, True]
                assert [names[id(p)] for p in sync.tails[1]] == ["block4.convTranspose.weight", "block3.convTranspose.weight"]
                order = [names[id(p)] for p in fg.params[:fg.n_main]]
                assert order[0].startswith("final_layer") and order[-1].endswith("zMapping.linear1.bias")
                assert order.index("x") < order.index("zMapping.linear1.weight")
            tr = Trainer(m, grad_sync=sync)
            assert tr.order == [0, 1, 1]
            np.random.seed(7)
            for k in range(6):
                m.noise_distn = FixedNoise(noises[k])
                tr.step(batches[k])
            tr.finish()
            torch.cuda.synchronize()
            results.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
            if sync is not None:
                stats = dict(sync.stats)
                sync.close()
        assert torch.equal(results[0], results[1])
        assert stats["buckets_after_backward"] == 0 and stats["buckets_deferred_tail"] == 4 * len(tails), stats
        # every generator pass: [issue main ...] ("deferred", 1, 2) [issue tail ...]
        k_def = [k for k, t in enumerate(trace) if t[0] == "deferred" and t[1] == 1]
        assert len(k_def) == 4 and all(trace[k][2] == 2 for k in k_def)
        for k in k_def:
            assert trace[k - 1][0] == "issue" and trace[k - 1][1] == 1 and trace[k - 1][2] not in tails
            assert [t[:2] for t in trace[k + 1:k + 1 + len(tails)]] == [("issue", 1)] * len(tails)
        # G -> G hand-over: between two generator passes with no discriminator issue in between
        k_issue = [k for k, t in enumerate(trace) if t[0] == "issue" and t[1] == 1]
        handovers = 0
        for a, b in zip(k_issue, k_issue[1:]):
            between = trace[a + 1:b]
            g = [t for t in between if t[1] == 1 and t[0] in ("gate", "wait")]
            if not g or any(t[0] == "issue" and t[1] == 0 for t in between):
                continue
            handovers += 1
            assert g[0][0] == "gate" and g[0][2] not in tails, g[:4]
            first_tail_wait = next(k for k, t in enumerate(g) if t[0] == "wait" and t[2] in tails)
            assert any(t[0] == "gate" and t[2] in tails for t in g[:first_tail_wait])
            assert any(t[0] == "wait" and t[2] not in tails for t in g[:first_tail_wait])
        assert handovers == 2, handovers
    finally:
        dist.destroy_process_group()
        os.environ.pop("GZ_DDP_ALWAYS_REDUCE", None)


def test_a_failed_step_leaves_no_pending_gradient_behind():
    """a training_step / backward that raises must not have its half-built sink state reduced (a second error
    that masks the first) nor leak slabs into the next step's gradients."""
    from helpers import FixedNoise, fill_closed_form, synthetic_noise, synthetic_real
    from lightning_gan_zoo_amd import functional as F
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.harness import Trainer

    def build():
        cfg = make_cfg("dc_gan", batch_size=8, features=8, noise_dim=16)
        torch.manual_seed(42)
        m = locate(cfg.model.lm["_target_"])(cfg, None)
        fill_closed_form(m.generator, 1)
        fill_closed_form(m.discriminator, 2)
        return m.cuda()

    labels = torch.zeros(8, dtype=torch.int64, device="cuda")
    batch = (synthetic_real(8, seed=3).cuda(), labels)
    out = []
    for fail in (True, False):
        m = build()
        tr = Trainer(m)
        if fail:
            real_step = m.training_step

            def broken(b, i, idx):
                loss = real_step(b, i, idx)
                loss.backward(retain_graph=True)          # slabs are pending in the sinks now
                raise ValueError("boom")

            m.training_step = broken
            m.noise_distn = FixedNoise(synthetic_noise(8, 16, 5))
            with pytest.raises(ValueError, match="boom"):
                tr.step(batch)
            assert not F._sinks.pending and not F.grad_sinks_enabled()
            m.training_step = real_step
            for p in m.parameters():
                p.grad = None
        for k in range(2):
            m.noise_distn = FixedNoise(synthetic_noise(8, 16, 9 + k))
            tr.step(batch)
        torch.cuda.synchronize()
        out.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
    assert torch.equal(out[0], out[1])
