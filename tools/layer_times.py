"""Per-layer GPU time of every implicit-GEMM / GEMM launch of one optimizer cycle (HIP events around each launch).

    python tools/layer_times.py <expt> [batch]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightning_gan_zoo_amd import functional as F  # noqa: E402

expt = sys.argv[1] if len(sys.argv) > 1 else "hologan"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH[expt]
img = int(sys.argv[3]) if len(sys.argv) > 3 else bench.NATIVE_IMG_SIZE.get(expt, 64)
torch.set_num_threads(min(8, torch.get_num_threads()))
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, False, img)
data = bench.synthetic_batch(batch, dev, 0, img)
per = len(trainer.order)
for _ in range(2 * per):
    trainer.step(data)
trainer.finish()
torch.cuda.synchronize()
timer = F.KernelTimer(detail=True)
F.set_kernel_timer(timer)
cycles = 4
for _ in range(cycles * per):
    trainer.step(data)
    torch.cuda.synchronize()        # keep the queue empty: event intervals = kernel time, not launch gaps
trainer.finish()
F.set_kernel_timer(None)
torch.cuda.synchronize()
agg = timer.summary()
tot = sum(v[1] for v in agg.values()) / cycles
print("%s bs %d: %.2f ms of timed GEMM-class launches per cycle" % (expt, batch, tot))
for label, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.3f ms/cycle %5.1f calls %8.1f us %6.1f TF  %s" % (ms / cycles, n / cycles, ms / n * 1e3,
                                                             fl / (ms * 1e-3) / 1e12, label))
