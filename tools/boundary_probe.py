"""Host wall-clock of what happens around a step boundary (between the last launch of one step's backward and the first
kernels of the next forward): where the GPU idles when the backward's tail is host-bound (HoloGAN).
   python tools/boundary_probe.py [expt] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lightning_gan_zoo_amd import functional as F, harness

expt = sys.argv[1] if len(sys.argv) > 1 else "hologan"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, img_size=bench.NATIVE_IMG_SIZE.get(expt, 64))
data = bench.synthetic_batch(batch, dev, 0, bench.NATIVE_IMG_SIZE.get(expt, 64))
acc, cnt = {}, {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
            cnt[label] = cnt.get(label, 0) + 1
    setattr(obj, name, w)


wrap(module, "training_step", "training_step")
wrap(module, "sample_noise", "  sample_noise")
wrap(module.generator, "forward", "  G.forward")
wrap(module.discriminator, "forward", "  D.forward")
for n in ("prefetch_view", "_take_prefetched"):
    if hasattr(module.generator, n):
        wrap(module.generator, n, "  G." + n)
wrap(harness, "toggle_optimizer", "toggle_optimizer")
wrap(F, "take_grad_sinks", "take_grad_sinks")
wrap(F, "flush_grad_sinks", "flush_grad_sinks")
wrap(F, "set_grad_sinks", "set_grad_sinks")
wrap(F, "_repack_group", "repack_group")
wrap(F, "linear_act_multi", "  linear_act_multi")
wrap(F, "spectral_normalize_multi", "  spectral_normalize_multi")
for o in trainer.optim:
    wrap(o["optimizer"], "step", "optimizer.step")
    if hasattr(o["optimizer"], "_step_from_slabs"):
        wrap(o["optimizer"], "_step_from_slabs", "  adam._step_from_slabs")
    wrap(o["optimizer"], "zero_grad", "optimizer.zero_grad")
orig_backward = torch.Tensor.backward


def bw(self, *a, **k):
    t0 = time.perf_counter()
    try:
        return orig_backward(self, *a, **k)
    finally:
        acc["backward"] = acc.get("backward", 0.0) + time.perf_counter() - t0
        cnt["backward"] = cnt.get("backward", 0) + 1


torch.Tensor.backward = bw
n = len(trainer.order)
for _ in range(3 * n):
    trainer.step(data)
torch.cuda.synchronize(); acc.clear(); cnt.clear()
cycles = 10
t0 = time.perf_counter()
for _ in range(cycles * n):
    trainer.step(data)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("%s bs %d: enqueue %.3f ms/cycle, drained %.3f ms/cycle (%d steps per cycle)" % (expt, batch, (t1 - t0) / cycles * 1e3, (t2 - t0) / cycles * 1e3, n))
tot = (t1 - t0) / cycles * 1e3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-28s %7.3f ms/cycle  %5.1f calls/cycle  %6.1f us/call" % (k, v / cycles * 1e3, cnt[k] / cycles, v / cnt[k] * 1e6))
