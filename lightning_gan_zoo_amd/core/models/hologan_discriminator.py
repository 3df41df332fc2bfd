"""HoloGAN discriminator on the MI355X HIP kernels; drop-in for reference
core/models/hologan_discriminator.py:7-78 (same constructor, same state_dict -- including the
``conv2d`` / ``conv2d_spec_norm`` double registration and the ``weight_orig/weight_u/weight_v``
spectral-norm entries -- same init order and quirks).

    Conv2d k5 s2 p2 + bias + LeakyReLU(0.2)            (fused epilogue)
    3 x [spectral-norm Conv2d k5 s2 p2 + bias -> InstanceNorm2d (no affine) + LeakyReLU]
    heads: Linear 8192->1 (logit); Linear 8192->128 + LeakyReLU -> Linear 128->z + tanh
"""
from torch import nn

from ... import functional as F

K5S2P2 = F.Geom(5, 5, 2, 2)


def truncated_normal_initializer(weight, mean=0, std=0.02):
    size = weight.shape
    tmp = weight.new_empty(size + (4,)).normal_()
    valid = (tmp < 2) & (tmp > -2)
    ind = valid.max(-1, keepdim=True)[1]
    weight.data.copy_(tmp.gather(-1, ind).squeeze(-1))
    weight.data.mul_(std).add_(mean)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.conv2d = nn.Conv2d(in_planes, out_planes, kernel_size=5, stride=2, padding=2)
        truncated_normal_initializer(self.conv2d.weight)
        nn.init.constant_(self.conv2d.bias, val=0.0)
        self.conv2d_spec_norm = nn.utils.spectral_norm(self.conv2d)     # parameter holder: weight_orig, weight_u/v
        self.instance_norm = nn.InstanceNorm2d(out_planes)
        self.lrelu = nn.LeakyReLU(0.2)

    def forward(self, x, w=None):
        """w: this call's spectral-normalised weight when the caller already has it (Discriminator.forward runs the
        power iteration of all three blocks in one set of launches)."""
        c = self.conv2d
        F.ready(c.bias)
        if w is None:
            F.ready(c.weight_orig)
            w = F.spectral_normalize(c.weight_orig, c.weight_u, c.weight_v, self.training)
        y = F.conv2d(x, w, c.bias, K5S2P2)
        return F.instance_norm_act(y, None, None, self.instance_norm.eps, F.ACT_LRELU, self.lrelu.negative_slope)


class Discriminator(nn.Module):
    def __init__(self, in_planes, out_planes, z_planes, img_size=64, final_sigmoid=False):
        """``img_size`` other than 64 is the EXT-128 extension (heads sized for the 8x8 final map); the
        reference hard-wires 4x4 (hologan_discriminator.py:41,45).  ``img_size`` / ``final_sigmoid`` are what the
        reference's ROOT config injects into every ``discriminator`` node (conf/config.yaml:35-37); the reference's
        own class (:28) rejects them, i.e. ``+expt=hologan`` does not construct there as shipped -- accepted here so
        that the shipped tree composes (the heads never apply a sigmoid: the criterion is BCE-with-logits)."""
        super().__init__()
        if final_sigmoid:
            raise ValueError("HoloGAN's discriminator returns logits (reference hologan_discriminator.py:64-70)")
        fmap = (img_size // 16) ** 2
        self.conv2d = nn.Conv2d(in_planes, out_planes, kernel_size=5, stride=2, padding=2)
        truncated_normal_initializer(self.conv2d.weight)
        nn.init.constant_(self.conv2d.bias, val=0.0)
        self.lrelu = nn.LeakyReLU(0.2)
        self.blocks = nn.Sequential(
            BasicBlock(out_planes * 1, out_planes * 2),
            BasicBlock(out_planes * 2, out_planes * 4),
            BasicBlock(out_planes * 4, out_planes * 8))
        self.linear1 = nn.Linear(out_planes * 8 * fmap, 1)
        truncated_normal_initializer(self.linear1.weight)
        nn.init.constant_(self.linear1.bias, val=0.0)
        self.linear2 = nn.Linear(out_planes * 8 * fmap, 128)
        truncated_normal_initializer(self.linear1.weight)     # sic (reference :46)
        nn.init.constant_(self.linear2.bias, val=0.0)
        self.linear3 = nn.Linear(128, z_planes)
        truncated_normal_initializer(self.linear1.weight)     # sic (reference :50)
        nn.init.constant_(self.linear3.bias, val=0.0)
        self.sigmoid = nn.Sigmoid()
        self.tanh = nn.Tanh()

    # Training-mode blocks run as IN_{eps sigma^2}(conv(x, weight_orig)) (functional.sn_conv_in_act): the same function of
    # the inputs as conv(x, weight_orig / sigma) + bias -> InstanceNorm, on weight_orig's cached packed images.  With it a
    # discriminator step can apply D ONCE to [real; fake] (`groups` = 2: two consecutive power iterations, one sigma per
    # half -- what the reference's two calls see).  False: the per-call weight copy of rounds 2-4a.
    fused_sn_blocks = True
    gates_parameters = True      # every layer announces its parameters with F.ready before reading them (ddp.GradSync)

    @property
    def supports_stacked_batches(self):
        return self.fused_sn_blocks and all(blk.conv2d.weight_orig.shape[1] % 4 == 0 for blk in self.blocks)

    def forward(self, x, groups=1):
        batch_size = x.size(0)
        slope = self.lrelu.negative_slope
        F.ready(self.conv2d.weight, self.conv2d.bias)
        h = F.conv2d(x, self.conv2d.weight, self.conv2d.bias, K5S2P2, F.ACT_LRELU, slope)
        # torch.nn.utils.spectral_norm's pre-forward hook of each block (reference :15,32) depends only on the block's
        # own weight and buffers: the three power iterations share their launches
        convs = [blk.conv2d for blk in self.blocks]
        layers = [(c.weight_orig, c.weight_u, c.weight_v) for c in convs]
        F.ready(*[p for c in convs for p in (c.weight_orig, c.bias)])       # (the power iterations read all three)
        if self.training and x.is_cuda and self.supports_stacked_batches:
            for blk, (sigma, us, vs) in zip(self.blocks, F.spectral_power_iterations(layers, calls=groups)):
                c = blk.conv2d
                h = F.sn_conv_in_act(h, c.weight_orig, c.bias, sigma, us, vs, K5S2P2, blk.instance_norm.eps, F.ACT_LRELU,
                                     blk.lrelu.negative_slope)
        else:
            if groups != 1:
                raise ValueError("a stacked batch needs the fused spectral-norm blocks (training mode, channel counts "
                                 "that are multiples of 4)")
            ws = F.spectral_normalize_multi(layers, self.training, K5S2P2)
            for blk, w in zip(self.blocks, ws):
                h = blk(h, w)
        h = h.reshape(batch_size, -1)
        F.ready(*[p for lin in (self.linear1, self.linear2, self.linear3) for p in (lin.weight, lin.bias)])
        logit = F.linear_act(h, self.linear1.weight, self.linear1.bias)
        enc = F.linear_act(h, self.linear2.weight, self.linear2.bias, F.ACT_LRELU, slope)
        z_prediction = F.linear_act(enc, self.linear3.weight, self.linear3.bias, F.ACT_TANH)
        return logit, z_prediction
