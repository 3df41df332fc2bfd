"""Host-side cost of one optimizer cycle: enqueue time (no sync inside) vs GPU time, plus a cProfile of the
enqueue loop.   python tools/host_profile.py <expt> [batch] [img] [--brief]   (--brief: the two timings only)"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

brief = "--brief" in sys.argv
use_sync = "--sync" in sys.argv          # the data-parallel code path: ddp.GradSync over a single-rank RCCL communicator
argv = [a for a in sys.argv if a not in ("--brief", "--sync")]
expt = argv[1] if len(argv) > 1 else "wgan_gp"
batch = int(argv[2]) if len(argv) > 2 else bench.DEFAULT_BATCH[expt]
img = int(argv[3]) if len(argv) > 3 else bench.NATIVE_IMG_SIZE.get(expt, 64)
torch.set_num_threads(min(8, torch.get_num_threads()))
dev = torch.device("cuda", 0)
if use_sync:
    ctx = bench.single_rank_rccl(dev)
    ctx.__enter__()
module, trainer = bench.build_trainer(expt, batch, dev, 1, use_sync, img)
data = bench.synthetic_batch(batch, dev, 0, img)
per = len(trainer.order)
for _ in range(3 * per):
    trainer.step(data)
trainer.finish()
torch.cuda.synchronize()
import gc  # noqa: E402
gc.collect()
gc.freeze()
n = 20
t0 = time.perf_counter()
for _ in range(n * per):
    trainer.step(data)
trainer.finish()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s bs %d img %d: host enqueue %.3f ms/cycle, enqueue + drain %.3f ms/cycle (%s)"
      % (expt, batch, img, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3,
         "the host runs ahead of the GPU: idle gaps are dependent-dispatch latency, not an empty queue"
         if (t1 - t0) < 0.9 * (t2 - t0) else "the host is the limit"))
if brief:
    if use_sync:
        trainer.grad_sync.close()
        ctx.__exit__(None, None, None)
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(5 * per):
    trainer.step(data)
trainer.finish()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(45)
if use_sync:
    trainer.grad_sync.close()
    ctx.__exit__(None, None, None)
