"""Diagnostic: are forward uses counted, is the sink listener called, where are buckets issued from?"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from lightning_gan_zoo_amd import functional as F
from lightning_gan_zoo_amd.config import locate, make_cfg
from lightning_gan_zoo_amd.ddp import GradSync
from lightning_gan_zoo_amd.harness import Trainer

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", GZ_DDP_ALWAYS_REDUCE="1")
dist.init_process_group("nccl", rank=0, world_size=1)
for expt in sys.argv[1:] or ["dc_gan"]:
    cfg = make_cfg(expt, batch_size=8)
    torch.manual_seed(42)
    m = locate(cfg.model.lm["_target_"])(cfg, None).cuda()
    sync = GradSync(m, bucket_bytes=4 << 20)
    calls = []
    orig = sync.sink_listener
    sync.sink_listener = lambda p: (calls.append(id(p)), orig(p))
    tr = Trainer(m, grad_sync=sync)
    real = torch.rand(8, 3, 64, 64).cuda() * 2 - 1
    lab = torch.zeros(8, dtype=torch.int64).cuda()
    for k in range(4):
        tr.step((real, lab))
    tr.finish()
    print(expt, "stats", sync.stats, "listener calls", len(calls), "buckets", [len(f.buckets) for f in sync.flats])
dist.destroy_process_group()
