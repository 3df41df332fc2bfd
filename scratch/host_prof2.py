import sys, time
sys.path.insert(0,'.')
import torch, bench
dev = torch.device('cuda',0)
bs = int(sys.argv[1]) if len(sys.argv)>1 else 512
module, trainer = bench.build_trainer('dc_gan', bs, dev, 1)
batch = bench.synthetic_batch(bs, dev, 0)
for _ in range(6): trainer.step(batch)
torch.cuda.synchronize()
for i in range(6):
    t0=time.perf_counter(); trainer.step(batch); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('step', i, 'opt', i%2, 'enqueue %.2f ms  drain %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3))
# now break down G step: time Adam alone
opt_g = trainer.optim[1]['optimizer']; opt_d = trainer.optim[0]['optimizer']
import lightning_gan_zoo_amd.harness as H
for idx in (0,1,0,1):
    H.toggle_optimizer(module, idx)
    loss = module.training_step(batch, 0, idx); torch.cuda.synchronize(); t0=time.perf_counter()
    loss.backward(); torch.cuda.synchronize(); t1=time.perf_counter()
    trainer.optim[idx]['optimizer'].step(); torch.cuda.synchronize(); t2=time.perf_counter()
    trainer.optim[idx]['optimizer'].zero_grad(); torch.cuda.synchronize(); t3=time.perf_counter()
    print('idx',idx,'backward %.2f ms, opt.step %.2f ms, zero_grad %.2f ms' % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
