"""Host-side cost of one optimizer cycle: enqueue time (no sync inside) vs GPU time, plus a cProfile of the
enqueue loop.   python tools/host_profile.py <expt> [batch]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

expt = sys.argv[1] if len(sys.argv) > 1 else "wgan_gp"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH[expt]
img = bench.NATIVE_IMG_SIZE.get(expt, 64)
torch.set_num_threads(min(8, torch.get_num_threads()))
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, False, img)
data = bench.synthetic_batch(batch, dev, 0, img)
per = len(trainer.order)
for _ in range(3 * per):
    trainer.step(data)
trainer.finish()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n * per):
    trainer.step(data)
trainer.finish()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s bs %d: enqueue %.2f ms/cycle, enqueue+drain %.2f ms/cycle" % (expt, batch, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5 * per):
    trainer.step(data)
trainer.finish()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
