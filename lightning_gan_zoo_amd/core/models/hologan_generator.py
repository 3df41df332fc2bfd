"""HoloGAN generator on the MI355X HIP kernels; drop-in for reference
core/models/hologan_generator.py:7-345 (same constructor, parameter names / shapes / init order):

    learned constant [1, 8C, 4,4,4] -> AdaIN(z)+ReLU
    -> 2 x [ConvTranspose3d k3 s2 p1 op1 (MFMA implicit GEMM, 8 sub-voxel phases) -> AdaIN(z)+ReLU]
    -> rigid-body resampling of the 16^3 volume under the sampled view (trilinear gather kernel that
       writes the permuted/flipped [N, 16C, 16, 16] map directly)
    -> 1x1 "learned projection" ConvTranspose2d + ReLU (fused epilogue)
    -> 2 x [ConvTranspose2d k4 s2 p1 -> AdaIN(z)+ReLU] -> Conv2d k3 p1 + tanh (fused epilogue)

The child nn modules are parameter holders; forward only calls lightning_gan_zoo_amd.functional.
"""
import math

import numpy as np
import torch
from torch import nn

from ... import functional as F
from ...harness import draw_on_host, few_host_threads

K3S1P1 = F.Geom(3, 3, 1, 1)
K1S1P0 = F.Geom(1, 1, 1, 0)


class ZMapping(nn.Module):
    def __init__(self, z_dimension, output_channel):
        super().__init__()
        self.output_channel = output_channel
        self.linear1 = nn.Linear(z_dimension, output_channel * 2)
        nn.init.normal_(self.linear1.weight, std=0.02)
        nn.init.constant_(self.linear1.bias, val=0.0)
        self.relu = nn.ReLU()

    def forward(self, x):
        """[N, 2C]: the AdaIN scale (first half) and shift (second half), consumed packed by adain_act_packed
        (the reference slices them, hologan_generator.py:17-18)."""
        F.ready(self.linear1.weight, self.linear1.bias)
        return F.linear_act(x, self.linear1.weight, self.linear1.bias, F.ACT_RELU)


class BasicBlock(nn.Module):
    def __init__(self, z_planes, in_planes, out_planes, transpose_dim):
        super().__init__()
        if transpose_dim == 2:
            self.convTranspose = nn.ConvTranspose2d(in_planes, out_planes, kernel_size=4, stride=2, padding=1)
        else:
            self.convTranspose = nn.ConvTranspose3d(in_planes, out_planes, kernel_size=3, stride=2,
                                                    output_padding=1, padding=1)
        nn.init.normal_(self.convTranspose.weight, std=0.02)
        nn.init.constant_(self.convTranspose.bias, val=0.0)
        self.zMapping = ZMapping(z_planes, out_planes)
        self.relu = nn.ReLU()
        self.transpose_dim = transpose_dim

    def forward(self, h, z, style=None):
        """style: this block's zMapping(z) when the caller already has it (Generator.forward maps z through all five
        ZMapping layers in one launch)."""
        ct = self.convTranspose
        F.ready(ct.weight, ct.bias)
        # (the AdaIN below normalises every (sample, channel) plane: the bias cannot reach the loss, its gradient is 0)
        if self.transpose_dim == 2:
            h = F.conv_transpose2d(h, ct.weight, ct.bias, F.K4S2P1, bias_cancels=True)
        else:
            h = F.conv_transpose3d(h, ct.weight, ct.bias, bias_cancels=True)
        return F.adain_act_packed(h, self.zMapping(z) if style is None else style, 1e-8, F.ACT_RELU)


def _mat(rows):
    return torch.cat([torch.cat(r, dim=2) for r in rows], dim=1)


def view_inverse_matrices(view, size=16, new_size=16):
    """Per-sample inverse of  C_new . (T . S . (Rz . Ry)) . C_old  -- computed on the HOST in fp32 with
    the reference's operation order (hologan_generator.py:145-214) so the matrices, and hence the
    integer voxel indices derived from them, agree with the reference bit for bit.  [N, 4, 4]."""
    view = torch.as_tensor(view)
    col = lambda i: view[:, i].reshape(-1, 1, 1).float()      # noqa: E731
    th, ga, sc, tx, ty, tz = (col(i) for i in range(6))
    one, zero = torch.ones_like(th), torch.zeros_like(th)
    rot_z = _mat([[th.cos(), th.sin(), zero, zero], [-th.sin(), th.cos(), zero, zero],
                  [zero, zero, one, zero], [zero, zero, zero, one]])
    rot_y = _mat([[ga.cos(), zero, ga.sin(), zero], [zero, one, zero, zero],
                  [-ga.sin(), zero, ga.cos(), zero], [zero, zero, zero, one]])
    scl = _mat([[sc, zero, zero, zero], [zero, sc, zero, zero], [zero, zero, sc, zero], [zero, zero, zero, one]])
    trn = _mat([[one, zero, zero, tx], [zero, one, zero, ty], [zero, zero, one, tz], [zero, zero, zero, one]])
    m = torch.matmul(torch.matmul(trn, scl), torch.matmul(rot_z, rot_y))
    n = view.shape[0]

    def centre(v):
        return torch.tensor([[1, 0, 0, v], [0, 1, 0, v], [0, 0, 1, v], [0, 0, 0, 1.0]]).reshape(1, 4, 4).repeat(n, 1, 1)

    return torch.matmul(torch.matmul(centre(new_size * 0.5), m), centre(-size * 0.5)).inverse()


class Generator(nn.Module):
    gates_parameters = True      # every layer announces its parameters with F.ready before reading them (ddp.GradSync)
    _warned_numpy_touched = False

    def __init__(self, in_planes, out_planes, z_planes, view_args, img_size, view_planes=6, gpu=True,
                 ext128=False):
        super().__init__()
        # `gpu` is accepted for signature parity; the module follows its parameters' device (.to(device))
        tensor = (torch.randn(1, in_planes * 8, 4, 4, 4) - 0.5) / 0.5
        self.x = nn.Parameter(tensor)
        self.view_args = view_args
        self.zMapping = ZMapping(z_planes, in_planes * 8)
        self.block1 = BasicBlock(z_planes, in_planes=in_planes * 8, out_planes=in_planes * 2, transpose_dim=3)
        self.block2 = BasicBlock(z_planes, in_planes=in_planes * 2, out_planes=in_planes, transpose_dim=3)
        self.convTranspose2d1 = nn.ConvTranspose2d(in_planes * 16, in_planes * 16, kernel_size=1)
        nn.init.normal_(self.convTranspose2d1.weight, std=0.02)
        nn.init.constant_(self.convTranspose2d1.bias, val=0.0)
        self.block3 = BasicBlock(z_planes, in_planes=in_planes * 16, out_planes=in_planes * 4, transpose_dim=2)
        self.block4 = BasicBlock(z_planes, in_planes=in_planes * 4, out_planes=in_planes, transpose_dim=2)
        if img_size == 64:
            self.final_layer = nn.Conv2d(in_planes, out_planes, kernel_size=3, padding=1)
        elif img_size == 128:
            # The reference's 128 branch (ConvTranspose2d k4 p1 with the default stride 1) yields 65x65 and
            # cannot feed its own discriminator.  EXT-128 (opt-in, throughput-only,
            # NOT parity-pinned) adopts the evident intent: stride 2, i.e. 64 -> 128.
            if not ext128:
                raise NotImplementedError("HoloGAN img_size=128 is broken in the reference (65x65 output); pass "
                                          "ext128=True for the stride-2 extension (not parity-pinned)")
            self.final_layer = nn.ConvTranspose2d(in_planes, out_planes, kernel_size=4, stride=2, padding=1)
        else:
            raise ValueError("img_size must be 64")
        nn.init.normal_(self.final_layer.weight, std=0.02)
        nn.init.constant_(self.final_layer.bias, val=0.0)
        self.relu = nn.ReLU()
        self.tanh = nn.Tanh()
        self.staged_minv = None     # harness.GraphedTrainer: the step's inverse view matrices, already on the device
        self._prefetched = None     # harness.Trainer: (matrices on the device, numpy state before, after, batch size)

    # -- data parallelism (ddp.GradSync; reference run_network.py:66, conf/expt/hologan.yaml:16-17: D, G, G) ----------
    def _zmaps(self):
        return (self.zMapping, self.block1.zMapping, self.block2.zMapping, self.block3.zMapping, self.block4.zMapping)

    def grad_arrival_order(self):
        """The parameters in the order their gradients complete during backward.  Not the reverse definition order: the
        five ZMapping layers are evaluated by ONE launch at the top of forward(), so their gradients come out of one
        launch at the very end of backward, whichever block owns them."""
        order = []
        for m in (self.final_layer, self.block4.convTranspose, self.block3.convTranspose, self.convTranspose2d1,
                  self.block2.convTranspose, self.block1.convTranspose):
            order += [m.weight, m.bias]
        order.append(self.x)
        for m in self._zmaps():
            order += [m.linear1.weight, m.linear1.bias]
        return order

    def deferred_tail_parameters(self):
        """Weight gradients whose LAUNCH is postponed to the end of backward and whose bucket travels last (ddp.py).
        The schedule runs two generator steps back to back, so the next user of these gradients is this network's own
        next forward: the constant, the ZMapping layers, the two 3-D blocks and the 1x1 projection (13 of 31 MB,
        needed first) travel underneath these launches -- block3 / block4 are ~60 % of the generator's FLOP -- and
        block3 / block4's own 17.8 MB underneath the early layers of that forward."""
        tail = [self.block4.convTranspose.weight, self.block3.convTranspose.weight]
        if isinstance(self.final_layer, nn.ConvTranspose2d):       # EXT-128: bias-free sink path (functional._ConvDg)
            tail.insert(0, self.final_layer.weight)
        return tail

    def sample_view(self, batch_size):
        """Transformation parameters from numpy's global generator, reference :80-114 (call order kept)."""
        a = self.view_args
        theta = np.random.randint(a["azimuth_low"], a["azimuth_high"], (batch_size)).astype(float) * math.pi / 180.0
        if a["elevation_low"] < a["elevation_high"]:
            gamma = np.random.randint(a["elevation_low"], a["elevation_high"], (batch_size)).astype(float)
            gamma = gamma * math.pi / 180.0
        else:
            gamma = np.zeros(batch_size).astype(float)
        scale = float(np.random.uniform(a["scale_low"], a["scale_high"]))
        shift = [a["trans%s_low" % ax] + np.random.random(batch_size) * (a["trans%s_high" % ax] - a["trans%s_low" % ax])
                 for ax in ("X", "Y", "Z")]
        view = np.zeros((batch_size, 6))
        view[:, 0], view[:, 1], view[:, 2] = theta, gamma, scale
        view[:, 3], view[:, 4], view[:, 5] = shift
        return view

    def prefetch_view(self, batch_size):
        """Draw and invert the NEXT forward's view matrices now (harness.Trainer calls this right after a step's
        training_step, with that forward still queued on the GPU): the ~0.5 ms of small host operations otherwise sit at the top
        of the next forward, where the queue holds only the first few small kernels and runs dry (0.13 ms idle per
        forward in the kernel trace, three forwards per optimizer cycle).  numpy's stream is consumed in the same order
        -- one draw per forward -- and the draw is provisional: if anything touches numpy's generator before the next
        forward (a reseed, another draw), or that forward has another batch size, the state is rolled back and the view
        is drawn where the reference draws it."""
        before = np.random.get_state()
        minv = view_inverse_matrices(self.sample_view(batch_size)).reshape(batch_size, 16).contiguous()
        after = np.random.get_state()
        self._prefetched = (draw_on_host(lambda: minv, self.x.device), before, after, batch_size)

    def _take_prefetched(self, n):
        item, self._prefetched = self._prefetched, None
        if item is None:
            return None
        minv, before, after, bs = item
        now = np.random.get_state()
        untouched = now[0] == after[0] and now[2:] == after[2:] and np.array_equal(now[1], after[1])
        if untouched and bs == n:
            return minv
        if untouched:                    # another batch size (an epoch's last batch), or an explicit view: undo the
            np.random.set_state(before)  # provisional draw
            return None
        # Someone used numpy's global generator between the provisional draw and this forward.  A RESEED (tests, a
        # resumed run) makes the provisional draw irrelevant: the view is drawn below from the new state, as the
        # reference would.  A DRAW cannot be told from a reseed here and cannot be undone: numpy's stream then reads
        # (view, other, view) where the reference reads (other, view) -- loudly, once, instead of silently (ADVICE r5;
        # nothing in this package draws there during training: eval.py's KID subsets follow a generator forward).
        if not Generator._warned_numpy_touched:
            Generator._warned_numpy_touched = True
            import warnings
            warnings.warn("HoloGAN: numpy's global generator was used between a training step and the next generator "
                          "forward; the view prefetched for that forward is dropped and drawn again.  After a reseed that "
                          "is exactly the reference's stream; after another draw it is not (call "
                          "generator.prefetch_view = lambda n: None to draw views inside forward() only).", stacklevel=3)
        return None

    def forward(self, z, view_in=None):
        n = z.shape[0]
        dev = self.x.device

        # the five ZMapping layers (reference :33, :57, :141) read the same z: one launch, [N, 2C] each
        maps = self._zmaps()
        F.ready(*[p for m in maps for p in (m.linear1.weight, m.linear1.bias)])
        s0, s1, s2, s3, s4 = F.linear_act_multi(z, [(m.linear1.weight, m.linear1.bias) for m in maps], F.ACT_RELU)
        F.ready(self.x)
        h = F.adain_const_act(self.x, s0, 1e-8, F.ACT_RELU)     # = AdaIN(self.x.repeat(n, ...)), reference :141
        h = self.block1(h, z, s1)
        h = self.block2(h, z, s2)
        # The view is sampled HERE, with the launches above already queued: drawing and inverting 4x4 matrices is
        # ~0.3 ms of small host ops during which the GPU would otherwise sit idle at every step boundary.  It is the
        # only numpy draw of a forward pass (reference :118-119 samples it first), so its place in numpy's stream --
        # and nothing else -- is unchanged.
        pre = self._take_prefetched(n if view_in is None else -1)      # (an explicit view consumes no draw)
        if pre is not None:
            minv = pre                       # drawn at the end of the previous step (prefetch_view)
        elif view_in is None and self.staged_minv is not None and self.staged_minv.shape[0] == n:
            minv = self.staged_minv          # drawn and inverted on the host by the trainer, in the reference's order
        else:
            if view_in is None:
                view_in = self.sample_view(n)
            minv = draw_on_host(lambda: view_inverse_matrices(view_in).reshape(n, 16).contiguous(), dev)
        h = F.rigid_resample(h, minv)                                 # [N, 16*C, 16, 16]
        p = self.convTranspose2d1
        F.ready(p.weight, p.bias)
        h = F.conv_transpose2d(h, p.weight, p.bias, K1S1P0, F.ACT_RELU)
        h = self.block3(h, z, s3)
        h = self.block4(h, z, s4)
        f = self.final_layer
        F.ready(f.weight, f.bias)
        if isinstance(f, nn.ConvTranspose2d):                          # EXT-128
            return F.conv_transpose2d(h, f.weight, f.bias, F.K4S2P1, F.ACT_TANH)
        return F.conv2d(h, f.weight, f.bias, K3S1P1, F.ACT_TANH)
