"""Shared test utilities: deterministic closed-form parameter fill, fixed noise
source, synthetic batches, tensor summaries.  No product or oracle imports."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def closed_form(numel, j, kind):
    """Pseudo-random but platform-independent values (float64 sin, cast to f32).
    kind: 'w' conv/linear weight (std 0.02), 'g' norm scale (1 +- 0.1), 'b' bias."""
    i = np.arange(numel, dtype=np.float64)
    s = np.sin(12.9898 * i + 78.233 * (j + 1) + 0.5 * np.sin(0.001 * i * (j + 3)))
    if kind == "w":
        v = 0.02 * np.sqrt(2.0) * s
    elif kind == "g":
        v = 1.0 + 0.1 * s
    else:
        v = 0.05 * s
    return torch.from_numpy(v.astype(np.float32))


@torch.no_grad()
def fill_closed_form(module, salt=0):
    """Overwrite every parameter of ``module`` (named order) with closed_form."""
    for j, (name, p) in enumerate(module.named_parameters()):
        leaf = name.rsplit(".", 1)[-1]
        if p.ndim >= 2 or leaf == "x":
            kind = "w"
        elif leaf == "weight":
            kind = "g"
        else:
            kind = "b"
        amp = 50.0 if leaf == "x" else 1.0   # HoloGAN's learned constant is O(1)
        p.copy_((closed_form(p.numel(), j + 101 * salt, kind) * amp).view_as(p).to(p.device))


class FixedNoise:
    """Stands in for ``cfg.model.noise_distn``: returns the queued tensors."""

    def __init__(self, *tensors):
        self.queue = list(tensors)
        self.calls = 0

    def push(self, *tensors):
        self.queue.extend(tensors)

    def sample(self, shape):
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        self.calls += 1
        return t.clone()


def synthetic_real(n, c=3, size=64, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, c, size, size, generator=g) * 2 - 1


def synthetic_noise(n, dim, seed, uniform=False):
    g = torch.Generator().manual_seed(seed)
    if uniform:
        return torch.rand(n, dim, generator=g) * 2 - 1
    return torch.randn(n, dim, generator=g)


def sample_indices(numel, k=16, seed=7):
    rng = np.random.RandomState(seed + numel % 9973)
    return np.sort(rng.randint(0, numel, size=min(k, numel)))


def summarize(t, k=16):
    """(l2 norm, sum, sampled entries) of a tensor as float64 numpy."""
    a = t.detach().double().cpu().reshape(-1).numpy()
    idx = sample_indices(a.size, k)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[idx]])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def seeded_inception_state(net, seed=0):
    """A state dict for an InceptionV3 (torchvision keys): He-scaled convolution weights, BatchNorm affine / running
    statistics away from their defaults.  Drawn key by key in state_dict order from ONE seeded host generator, so the
    reference-built network (tests/golden/make_inception_golden.py), the CPU oracle and the product get the same
    numbers as long as their state_dict key order agrees -- which the fixture also pins."""
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith("conv.weight"):
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("bn.weight"):
            sd[k] = 1 + 0.2 * (torch.rand(v.shape, generator=g) - 0.5)
        elif k.endswith("bn.bias"):
            sd[k] = 0.2 * (torch.rand(v.shape, generator=g) - 0.5)
        elif k.endswith("running_mean"):
            sd[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
    return sd
