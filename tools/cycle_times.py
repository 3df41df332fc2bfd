"""Per-cycle GPU time of a configuration from HIP events around every optimizer cycle (no profiler):
   python tools/cycle_times.py [expt] [batch] [cycles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

expt = sys.argv[1] if len(sys.argv) > 1 else "dc_gan"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, img_size=bench.NATIVE_IMG_SIZE.get(expt, 64))
data = bench.synthetic_batch(batch, dev, 0, bench.NATIVE_IMG_SIZE.get(expt, 64))
n = len(trainer.order)
for _ in range(5 * n):
    trainer.step(data)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(cycles + 1)]
ev[0].record()
for c in range(cycles):
    for _ in range(n):
        trainer.step(data)
    ev[c + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(cycles)]
print("%s bs %d: per-cycle ms, %d cycles: min %.3f median %.3f mean %.3f max %.3f" % (expt, batch, cycles, min(t), sorted(t)[len(t) // 2], sum(t) / len(t), max(t)))
print(" ".join("%.2f" % x for x in t))
