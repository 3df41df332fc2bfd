"""Host wall-clock per section of harness.Trainer.step (no device sync inside), plain vs ddp.GradSync (single-rank RCCL).
   python tools/step_sections.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lightning_gan_zoo_amd import functional as F, ddp

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.set_num_threads(8)

def run(use_sync):
    module, trainer = bench.build_trainer("dc_gan", batch, dev, 1, use_sync, 64)
    data = bench.synthetic_batch(batch, dev, 0, 64)
    acc = {}
    def wrap(obj, name, label):
        fn = getattr(obj, name)
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        setattr(obj, name, w)
    wrap(module, "training_step", "training_step")
    wrap(module.generator, "forward", "  G.forward")
    wrap(module.discriminator, "forward", "  D.forward")
    wrap(module, "sample_noise", "  sample_noise")
    wrap(F, "ready", "  F.ready")
    wrap(F, "_repack_group", "  repack")
    orig_backward = torch.Tensor.backward
    def bw(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return orig_backward(self, *a, **k)
        finally:
            acc["backward"] = acc.get("backward", 0.0) + time.perf_counter() - t0
    torch.Tensor.backward = bw
    if use_sync:
        s = trainer.grad_sync
        for n in ("before_step", "after_backward", "finalize", "_issue", "_flush_sinks", "_step", "_timed_wait"):
            wrap(s, n, "sync." + n)
    for o in trainer.optim:
        wrap(o["optimizer"], "step", "optimizer.step")
    for _ in range(6):
        trainer.step(data)
    trainer.finish(); torch.cuda.synchronize(); acc.clear()
    n = 20
    t0 = time.perf_counter()
    for _ in range(2 * n):
        trainer.step(data)
    trainer.finish()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.Tensor.backward = orig_backward
    print("%s: enqueue %.3f ms/pair, drained %.3f ms/pair" % ("gradsync" if use_sync else "plain", (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("   %-22s %.3f ms/pair" % (k, v / n * 1e3))
    if use_sync:
        trainer.grad_sync.close()

run(False)
with bench.single_rank_rccl(dev):
    run(True)
