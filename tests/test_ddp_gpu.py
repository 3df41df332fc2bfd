"""The data-parallel gradient path on the GPU with RCCL ("nccl" backend), single rank: flat gradient
buffers + asynchronous all-reduce + deferred optimizer step must give bit-identical parameters to the
plain Trainer (world size 1 => the mean is the identity)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gradsync_on_rccl_single_rank_matches_plain_trainer():
    import torch.distributed as dist
    from helpers import FixedNoise, fill_closed_form, synthetic_noise, synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.ddp import GradSync
    from lightning_gan_zoo_amd.harness import Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      GZ_DDP_ALWAYS_REDUCE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def build():
            cfg = make_cfg("dc_gan", batch_size=8, features=8, noise_dim=16)
            torch.manual_seed(42)
            m = locate(cfg.model.lm["_target_"])(cfg, None)
            fill_closed_form(m.generator, 1)
            fill_closed_form(m.discriminator, 2)
            return m.cuda()

        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        batches = [(synthetic_real(8, seed=k).cuda(), labels) for k in range(4)]
        noises = [synthetic_noise(8, 16, 40 + k) for k in range(4)]
        results = []
        for use_sync in (True, False):
            m = build()
            tr = Trainer(m, grad_sync=GradSync(m) if use_sync else None)
            for k in range(4):
                m.noise_distn = FixedNoise(noises[k])
                tr.step(batches[k])
            tr.finish()
            torch.cuda.synchronize()
            results.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu())
        assert torch.equal(results[0], results[1])
    finally:
        dist.destroy_process_group()
        os.environ.pop("GZ_DDP_ALWAYS_REDUCE", None)
