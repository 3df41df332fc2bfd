import sys, time, cProfile, pstats, io
sys.path.insert(0,'.')
import torch, bench
dev = torch.device('cuda',0)
module, trainer = bench.build_trainer('dc_gan', 512, dev, 1)
batch = bench.synthetic_batch(512, dev, 0)
for _ in range(6): trainer.step(batch)
torch.cuda.synchronize()
# per-phase host timing without syncs
import lightning_gan_zoo_amd.harness as H
ts=[]
for i in range(6):
    t0=time.perf_counter(); trainer.step(batch); ts.append((time.perf_counter()-t0)*1e3)
torch.cuda.synchronize()
print('host ms per trainer.step (async):', [round(t,2) for t in ts])
pr = cProfile.Profile(); pr.enable()
for i in range(6): trainer.step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4000])
