"""One-GPU stress of the N-rank data-parallel code path at the reference's network widths (reference
run_network.py:27,66 -- ``seed_everything(42)`` on every rank, ``accelerator="ddp"``).

    GZ_REHEARSE_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P tools/rehearse_ranks.py [--cycles 3] [--batch 16] [--out file.json]

Every rank drives cuda:0, the collectives run on gloo (RCCL refuses several ranks on one device): for dc_gan and
hologan at features 64, ``ddp.GradSync(broadcast_buffers=True)`` (torch DDP's default buffer broadcast) under
``harness.Trainer``, per-rank batches, ``cycles`` optimizer cycles.  Checks, on every rank: no deadlock; parameters AND
buffers bit-identical on all ranks afterwards; every bucket issued from a hook or as the deferred tail; and reports the
host's enqueue time per cycle (the step calls, no synchronisation) next to the GPU's time per cycle -- eight ranks
share this box's cores and ONE GPU, so the GPU time is ~8x a real rank's: the ratio that matters is host enqueue /
(GPU time / world).  Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ.get("GZ_REHEARSE_ONE_GPU"), "one-GPU rehearsal only"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    torch.set_num_threads(1)
    import numpy as np
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import Trainer
    report = {"world": world, "cycles": args.cycles}
    for expt, bs, img in (("dc_gan", args.batch, 64), ("hologan", max(8, args.batch // 2), 64)):
        cfg = make_cfg(expt, batch_size=bs, img_size=img)          # the reference's default widths (features 64)
        torch.manual_seed(42)
        np.random.seed(42)
        module = locate(cfg.model.lm["_target_"])(cfg, None).to(dev)
        sync = GradSync(module, broadcast_buffers=True)
        tr = Trainer(module, grad_sync=sync)
        g = torch.Generator().manual_seed(1234 + rank)
        real = (torch.rand(bs, 3, img, img, generator=g) * 2 - 1).to(dev)
        batch = (real, torch.zeros(bs, dtype=torch.int64, device=dev))
        per = len(tr.order)
        for _ in range(per):                                        # warm-up cycle (allocator, packs)
            tr.step(batch)
        tr.finish()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.cycles * per):
            tr.step(batch)
        t_host = time.perf_counter() - t0
        tr.finish()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        dist.barrier()
        state = torch.cat([p.detach().reshape(-1).float() for p in module.parameters()] +
                          [b.detach().reshape(-1).float() for b in module.buffers()]).cpu()
        ref = state.clone()
        dist.broadcast(ref, src=0)
        same = bool(torch.equal(state, ref))
        flags = torch.tensor([int(same), int(bool(torch.isfinite(state).all()))])
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        times = torch.tensor([t_host, t_all], dtype=torch.float64)
        tmax = times.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        stats = dict(sync.stats)
        sync.close()
        report[expt] = {"batch_per_rank": bs, "identical_on_all_ranks": bool(flags[0]), "finite": bool(flags[1]),
                        "host_enqueue_ms_per_cycle_max": round(float(tmax[0]) / args.cycles * 1e3, 2),
                        "gpu_ms_per_cycle_all_ranks_on_one_gpu": round(float(tmax[1]) / args.cycles * 1e3, 2),
                        "buckets": [[(e - s) * 4 for s, e, _, _ in fg.buckets] for fg in sync.flats], **stats}
        assert flags[0] and flags[1], (expt, rank, same)
        assert stats["buckets_after_backward"] == 0 and stats["buckets_from_hooks"] > 0, stats
        del tr, sync, module
        torch.cuda.empty_cache()
    if rank == 0:
        line = json.dumps(report)
        print(line, flush=True)
        if args.out:
            with open(args.out, "w") as f:
                f.write(line + "\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
