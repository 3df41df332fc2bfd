"""TEST INFRASTRUCTURE ONLY -- a stand-in for ``torchvision.models.inception`` (torchvision is not installed in this
image) so that the reference's OWN FID network code can be executed when the fixtures are generated.

What the reference runs (core/submodules/gan_stability/metrics/inception.py, a vendored copy of pytorch-fid's
inception.py, which core/callback_inception_metrics.py:8,210-211 uses): ``fid_inception_v3()`` builds
``torchvision.models.inception_v3(num_classes=1008, aux_logits=False, pretrained=False)`` and replaces nine of its
blocks by ``FIDInceptionA / C / E_1 / E_2`` -- subclasses of torchvision's ``InceptionA / C / E`` that inherit the
CONSTRUCTORS and override ``forward`` -- then ``InceptionV3.forward`` resizes, rescales and walks the blocks.  Those
classes and forwards are the reference's own and run unmodified (tests/golden/make_inception_golden.py); what this file
supplies is the part that lives in the absent dependency: the layer definitions of torchvision's blocks and of
``Inception3`` (restated from torchvision 0.10.0, torchvision/models/inception.py: BasicConv2d :451-463, InceptionA
:194-231, InceptionB :234-262, InceptionC :265-309, InceptionD :312-344, InceptionE :347-393, Inception3.__init__
:64-121; the layout is unchanged from 0.6 to 0.15) and the forwards of the two blocks the reference does not patch
(InceptionB, InceptionD).  Nothing in the product or in the GPU-side tests imports it: the product network
(lightning_gan_zoo_amd/inception.py) and the flat CPU oracle (oracle/inception_cpu.py) are held to the fixture this
produces.
"""
import torch
import torch.nn.functional as F
from torch import nn


class BasicConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, bias=False, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)

    def forward(self, x):
        x = self.conv(x)
        x = self.bn(x)
        return F.relu(x, inplace=True)


class InceptionA(nn.Module):
    def __init__(self, in_channels, pool_features, conv_block=None):
        super().__init__()
        conv_block = conv_block or BasicConv2d
        self.branch1x1 = conv_block(in_channels, 64, kernel_size=1)
        self.branch5x5_1 = conv_block(in_channels, 48, kernel_size=1)
        self.branch5x5_2 = conv_block(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = conv_block(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = conv_block(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = conv_block(96, 96, kernel_size=3, padding=1)
        self.branch_pool = conv_block(in_channels, pool_features, kernel_size=1)

    def forward(self, x):        # torchvision's own (the reference overrides it)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([self.branch1x1(x), b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        conv_block = conv_block or BasicConv2d
        self.branch3x3 = conv_block(in_channels, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = conv_block(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = conv_block(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = conv_block(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, bd, bp], 1)


class InceptionC(nn.Module):
    def __init__(self, in_channels, channels_7x7, conv_block=None):
        super().__init__()
        conv_block = conv_block or BasicConv2d
        c7 = channels_7x7
        self.branch1x1 = conv_block(in_channels, 192, kernel_size=1)
        self.branch7x7_1 = conv_block(in_channels, c7, kernel_size=1)
        self.branch7x7_2 = conv_block(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = conv_block(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = conv_block(in_channels, c7, kernel_size=1)
        self.branch7x7dbl_2 = conv_block(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = conv_block(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = conv_block(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = conv_block(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = conv_block(in_channels, 192, kernel_size=1)

    def forward(self, x):        # torchvision's own (the reference overrides it)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_1(x)
        for name in ("branch7x7dbl_2", "branch7x7dbl_3", "branch7x7dbl_4", "branch7x7dbl_5"):
            bd = getattr(self, name)(bd)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([self.branch1x1(x), b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        conv_block = conv_block or BasicConv2d
        self.branch3x3_1 = conv_block(in_channels, 192, kernel_size=1)
        self.branch3x3_2 = conv_block(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = conv_block(in_channels, 192, kernel_size=1)
        self.branch7x7x3_2 = conv_block(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = conv_block(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = conv_block(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, b7, bp], 1)


class InceptionE(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        conv_block = conv_block or BasicConv2d
        self.branch1x1 = conv_block(in_channels, 320, kernel_size=1)
        self.branch3x3_1 = conv_block(in_channels, 384, kernel_size=1)
        self.branch3x3_2a = conv_block(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = conv_block(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = conv_block(in_channels, 448, kernel_size=1)
        self.branch3x3dbl_2 = conv_block(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = conv_block(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = conv_block(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = conv_block(in_channels, 192, kernel_size=1)

    def forward(self, x):        # torchvision's own (the reference overrides it)
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([self.branch1x1(x), b3, bd, bp], 1)


class Inception3(nn.Module):
    """Attribute layout of torchvision's Inception3 (the reference takes the layers by name; it never calls forward)."""

    def __init__(self, num_classes=1000, aux_logits=True, transform_input=False, inception_blocks=None,
                 init_weights=None):
        super().__init__()
        if aux_logits:
            raise NotImplementedError("the FID network is built with aux_logits=False (reference inception.py:174-175)")
        self.aux_logits, self.transform_input = aux_logits, transform_input
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.maxpool2 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Mixed_5b = InceptionA(192, pool_features=32)
        self.Mixed_5c = InceptionA(256, pool_features=64)
        self.Mixed_5d = InceptionA(288, pool_features=64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, channels_7x7=128)
        self.Mixed_6c = InceptionC(768, channels_7x7=160)
        self.Mixed_6d = InceptionC(768, channels_7x7=160)
        self.Mixed_6e = InceptionC(768, channels_7x7=192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout()
        self.fc = nn.Linear(2048, num_classes)


def inception_v3(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError("no network: pretrained weights cannot be fetched")
    return Inception3(**kwargs)


def install():
    """Register ``torchvision``, ``torchvision.models``, ``torchvision.models.inception`` and
    ``torchvision.models.utils`` stand-ins in sys.modules (only if torchvision is really absent).  Returns the
    ``torchvision.models.utils`` module so that the caller can plant ``load_state_dict_from_url``."""
    import sys
    import types
    try:
        import torchvision  # noqa: F401
        raise RuntimeError("torchvision IS installed: generate the fixture with the real package instead")
    except ImportError:
        pass
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    inc = types.ModuleType("torchvision.models.inception")
    utils = types.ModuleType("torchvision.models.utils")
    for name in ("BasicConv2d", "InceptionA", "InceptionB", "InceptionC", "InceptionD", "InceptionE", "Inception3",
                 "inception_v3"):
        setattr(inc, name, globals()[name])
    models.inception, models.inception_v3, models.Inception3, models.utils = inc, inception_v3, Inception3, utils
    tv.models = models

    def no_network(*a, **k):
        raise RuntimeError("no network: plant a state dict provider before building the FID network")

    utils.load_state_dict_from_url = no_network
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.models.inception": inc,
                        "torchvision.models.utils": utils})
    return utils


def uninstall():
    import sys
    for name in ("torchvision", "torchvision.models", "torchvision.models.inception", "torchvision.models.utils"):
        sys.modules.pop(name, None)
