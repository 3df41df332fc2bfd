import sys, time, gc
sys.path.insert(0,'.')
import torch, bench
dev = torch.device('cuda',0)
bs = 512
module, trainer = bench.build_trainer('dc_gan', bs, dev, 1)
batch = bench.synthetic_batch(bs, dev, 0)
for _ in range(6): trainer.step(batch)
torch.cuda.synchronize()
def stats():
    s = torch.cuda.memory_stats()
    return (s.get('num_device_alloc',0), s.get('num_device_free',0), s['reserved_bytes.all.current']>>20, s['allocated_bytes.all.peak']>>20, s.get('num_alloc_retries',0))
print('stats', stats(), 'gc', gc.get_count())
for i in range(8):
    t0=time.perf_counter(); trainer.step(batch); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('step', i, 'opt', i%2, 'enqueue %.2f ms  drain %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3), stats(), gc.get_count())
gc.disable()
for i in range(8):
    t0=time.perf_counter(); trainer.step(batch); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('nogc step', i, 'opt', i%2, 'enqueue %.2f ms  drain %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3), stats())
