"""Parameter tree of the YOLOX-24p network with the reference's module / state-dict names.

These modules only HOLD parameters and describe the graph; the arithmetic is the HIP plan in
``ep24.engine`` (NHWC bf16 MFMA convs, fused BN+SiLU, ...), entered through ``YOLOX.forward``.
Names follow the reference so its checkpoints load unchanged:
  network_blocks.py:29-210 (BaseConv, DWConv, Bottleneck, CSPLayer, SPPBottleneck, Focus), darknet.py:95-177
  (CSPDarknet), yolo_pafpn.py:27-81 (YOLOPAFPN), yolo_head_24p.py:47-141 (YOLOXHead), yolox.py:17-34 (YOLOX).
"""
import math

import torch
import torch.nn as nn

from . import _lib


ACT_CODES = {"silu": 1, "relu": 2, "lrelu": 3}          # activation codes of the BN + act kernels (csrc/common.h act_fwd)



def _unit_relu_forward(mod, x):
    """conv -> BatchNorm -> ReLU units of the swapped backbones; a unit over 3-channel images runs as im2col rows x GEMM."""
    if mod.conv.in_channels == 3:
        return _sub(mod, "stem_unit", x)[0]
    return _sub(mod, "unit_relu", x)[0]


def _sub(mod, kind, *tensors, train=True):
    """forward of one module of the tree on its own: a sub-plan of the same engine (ep24.engine.SubEngine)."""
    from .engine import run_submodule
    return run_submodule(mod, kind, tensors, train)


class BaseConv(nn.Module):
    """conv (no bias) -> BatchNorm (eps 1e-3, momentum 0.03 as patched by Exp.get_model) -> SiLU."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        depthwise = groups == in_channels == out_channels and groups > 1
        if bias or (groups != 1 and not depthwise):
            raise NotImplementedError("ep24 hot path: conv without bias, dense or depthwise (groups = channels: DWConv); other groupings are not in the reference")
        if depthwise and (ksize != 3 or in_channels % 8):
            raise NotImplementedError("ep24: depthwise convs are the 3x3 ones of DWConv over a multiple of 8 channels (got k=%d, C=%d)" % (ksize, in_channels))
        if act not in ACT_CODES:
            raise AttributeError("Unsupported act type: {}".format(act))       # get_activation, network_blocks.py:17-26
        self.conv = nn.Conv2d(in_channels, out_channels, ksize, stride, (ksize - 1) // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, eps=1e-3, momentum=0.03)
        self.act = {"silu": nn.SiLU, "relu": nn.ReLU}[act](inplace=True) if act != "lrelu" else nn.LeakyReLU(0.1, inplace=True)
        self.act_code = ACT_CODES[act]

    def forward(self, x):
        """act(bn(conv(x))) (network_blocks.py:50-51); NCHW in / NCHW fp32 out."""
        return _sub(self, "baseconv", x)[0]


class DWConv(nn.Module):
    """Depthwise conv + 1x1 conv, each a BaseConv (network_blocks.py:57-76); `depthwise=True` of the constructors below puts it where
    the reference does (darknet.py:107, network_blocks.py:92, yolo_pafpn.py:30, yolo_head_24p.py:45).  The depthwise unit runs as
    HBM-bound elementwise kernels (csrc/dwconv.hip), the 1x1 unit as every other 1x1 unit."""

    def __init__(self, in_channels, out_channels, ksize, stride=1, act="silu"):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride, groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def forward(self, x):
        return _sub(self, "dwconv", x)[0]


def _conv_cls(depthwise):
    return DWConv if depthwise else BaseConv


class Focus(nn.Module):
    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    def forward(self, x):
        """space-to-depth + conv (network_blocks.py:199-210): images [B,3,H,W] -> [B,C,H/2,W/2]."""
        return _sub(self, "focus", x)[0]


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, 1, act=act)
        self.conv2 = _conv_cls(depthwise)(hidden, out_channels, 3, 1, act=act)           # network_blocks.py:92
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x):
        return _sub(self, "bottleneck", x)[0]                   # network_blocks.py:91-95


class CSPLayer(nn.Module):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, 1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, 1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, 1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def forward(self, x):
        return _sub(self, "csp", x)[0]                          # network_blocks.py:179-185


class SPPBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        if tuple(kernel_sizes) != (5, 9, 13):
            raise NotImplementedError("SPP kernel sizes are fixed to (5, 9, 13)")
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, 1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(k, 1, k // 2) for k in kernel_sizes])
        self.conv2 = BaseConv(hidden * 4, out_channels, 1, 1, act=activation)

    def forward(self, x):
        return _sub(self, "spp", x)[0]                          # network_blocks.py:139-144


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, out_features=("dark3", "dark4", "dark5"), depthwise=False, act="silu"):
        super().__init__()
        self.out_features = out_features
        c = int(wid_mul * 64)
        d = max(round(dep_mul * 3), 1)
        Conv, dw = _conv_cls(depthwise), depthwise                                    # darknet.py:107
        self.stem = Focus(3, c, ksize=3, act=act)
        self.dark2 = nn.Sequential(Conv(c, c * 2, 3, 2, act=act), CSPLayer(c * 2, c * 2, n=d, depthwise=dw, act=act))
        self.dark3 = nn.Sequential(Conv(c * 2, c * 4, 3, 2, act=act), CSPLayer(c * 4, c * 4, n=d * 3, depthwise=dw, act=act))
        self.dark4 = nn.Sequential(Conv(c * 4, c * 8, 3, 2, act=act), CSPLayer(c * 8, c * 8, n=d * 3, depthwise=dw, act=act))
        self.dark5 = nn.Sequential(Conv(c * 8, c * 16, 3, 2, act=act), SPPBottleneck(c * 16, c * 16, activation=act),
                                   CSPLayer(c * 16, c * 16, n=d, shortcut=False, depthwise=dw, act=act))

    def forward(self, x):
        """{"dark3", "dark4", "dark5"} feature maps of the images (darknet.py:165-177; out_features as constructed)."""
        d3, d4, d5 = _sub(self, "darknet", x)
        feats = {"dark3": d3, "dark4": d4, "dark5": d5}
        return {k: v for k, v in feats.items() if k in self.out_features}


class ResBottleneck(nn.Module):
    """torchvision-style bottleneck of the swapped backbone (darknet.py:230-271): 1x1 -> 3x3 (stride) -> 1x1 (x4), each
    followed by BatchNorm, ReLU after the first two and after ``out += identity``; attribute names as in the reference."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity) (darknet.py:252-271) as a sub-plan."""
        return _sub(self, "resblock", x)[0]


class ResNet(nn.Module):
    """The reference's half-width ResNet-50 feature extractor (darknet.py:274-429, ``resnet50()``): 7x7/2 stem with 32
    filters, 3x3/2 max pool, stages of [3, 4, 6, 3] bottlenecks with 32/64/128/256 planes; dark3/4/5 = the outputs of
    layer2/3/4 (256/512/1024 channels at /8, /16, /32).  ``fc`` and ``baseconv1..3`` exist in the reference (and in its
    checkpoints) but take no part in the forward pass; they are kept so state dicts match."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.out_features = ("dark3", "dark4", "dark5")
        self.inplanes = 32
        self.conv1 = nn.Conv2d(3, 32, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(32, layers[0], 1)
        self.layer2 = self._make_layer(64, layers[1], 2)
        self.layer3 = self._make_layer(128, layers[2], 2)
        self.layer4 = self._make_layer(256, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)                         # 512 * expansion, as written (never run)

        def unused(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.SiLU())
        self.baseconv1, self.baseconv2, self.baseconv3 = unused(512, 128), unused(1024, 256), unused(2048, 256)
        for m in self.modules():                                       # darknet.py:332-338
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [ResBottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [ResBottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        """images -> {"dark3", "dark4", "dark5"} = the outputs of layer2 / layer3 / layer4 (darknet.py:405-429)."""
        d3, d4, d5 = _sub(self, "backbone", x)
        return {"dark3": d3, "dark4": d4, "dark5": d5}

    def used_units(self):
        """(conv, bn) pairs in execution order - what the plan runs and the flat parameter buffers are ordered by."""
        yield self.conv1, self.bn1
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                yield blk.conv1, blk.bn1
                yield blk.conv2, blk.bn2
                if blk.downsample is not None:
                    yield blk.downsample[0], blk.downsample[1]
                yield blk.conv3, blk.bn3


def resnet50():
    return ResNet((3, 4, 6, 3))


class BaseConv_DN(nn.Module):
    """conv -> BatchNorm -> ReLU of the DenseNet stem and output convs (darknet.py:518-529)."""

    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return _unit_relu_forward(self, x)                     # darknet.py:525-529


class ConvBlock(nn.Module):
    """Pre-activation BatchNorm -> ReLU -> conv (darknet.py:532-543)."""

    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.bn = nn.BatchNorm2d(in_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(in_channels, out_channels, **kwargs)

    def forward(self, x):
        return _sub(self, "convblock", x)[0]                   # conv(relu(bn(x))), darknet.py:539-543


class Transition(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.trans = nn.Sequential(ConvBlock(in_channels, out_channels, kernel_size=1, stride=1, bias=False), nn.AvgPool2d(2, 2))

    def forward(self, x):
        return _sub(self, "transition", x)[0]                  # darknet.py:555-556


class DenseLayer(nn.Module):
    """1x1 conv to 4k channels, 3x3 conv to k = 32, Dropout2d (darknet.py:560-577)."""

    def __init__(self, in_channels, drop_rate=0):
        super().__init__()
        self.growth_rate, self.bn_size = 32, 4
        self.conv_block = nn.Sequential(ConvBlock(in_channels, 128, kernel_size=1, stride=1, bias=False),
                                        ConvBlock(128, 32, kernel_size=3, stride=1, padding=1, bias=False))
        self.drop_rate = float(drop_rate)
        self.dropout = nn.Dropout2d(self.drop_rate)

    def forward(self, x):
        """The 32 new channels; Dropout2d draws on the device in training mode (darknet.py:573-577)."""
        return _sub(self, "denselayer", x)[0]


class DenseBlock(nn.Module):
    def __init__(self, num_layers, in_channels, drop_rate=0):
        super().__init__()
        self.growth_rate = 32
        self.denseblock = nn.Sequential(*[DenseLayer(in_channels + i * 32, drop_rate=drop_rate) for i in range(num_layers)])

    def forward(self, x):
        return _sub(self, "denseblock", x)[0]                  # cat(x, layer_0(x), layer_1(...), ...), darknet.py:593-597


class DenseNet(nn.Module):
    """``densenet121()`` of the reference (darknet.py:586-674): growth 32, blocks [6, 12, 24, 16], Dropout2d(0.3) in every dense
    layer; dark3 / dark4 = 1x1 conv-BN-ReLU of the D2 / D3 outputs, dark5 = the raw D4 concatenation (1024 channels)."""

    def __init__(self, growth_rate=32, block_layer=(6, 12, 24, 16)):
        super().__init__()
        self.out_features = ("dark3", "dark4", "dark5")
        self.growth_rate, self.block_config, self.num_init_channels = growth_rate, list(block_layer), 64
        self.stem = nn.Sequential(BaseConv_DN(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.MaxPool2d(3, 2, 1))
        t1 = 64 + block_layer[0] * 32
        t2 = t1 // 2 + block_layer[1] * 32
        t3 = t2 // 2 + block_layer[2] * 32
        self.D1 = DenseBlock(block_layer[0], 64, drop_rate=0.3)
        self.T1 = Transition(t1, t1 // 2)
        self.D2 = DenseBlock(block_layer[1], t1 // 2, drop_rate=0.3)
        self.T2 = Transition(t2, t2 // 2)
        self.D3 = DenseBlock(block_layer[2], t2 // 2, drop_rate=0.3)
        self.T3 = Transition(t3, t3 // 2)
        self.D4 = DenseBlock(block_layer[3], t3 // 2, drop_rate=0.3)
        self.baseconv1 = BaseConv_DN(t2, t2 // 2, kernel_size=1, bias=False)
        self.baseconv2 = BaseConv_DN(t3, t3 // 2, kernel_size=1, bias=False)

    def forward(self, x):
        """images -> {"dark3", "dark4", "dark5"} (darknet.py:640-674)."""
        d3, d4, d5 = _sub(self, "backbone", x)
        return {"dark3": d3, "dark4": d4, "dark5": d5}


def densenet121():
    return DenseNet(32, (6, 12, 24, 16))


class ConvBNReLU(nn.Module):
    """conv -> BatchNorm -> ReLU of the VGG backbone (darknet.py:432-445)."""

    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return _unit_relu_forward(self, x)                     # darknet.py:441-445


class VGG(nn.Module):
    """``vgg19()`` of the reference (darknet.py:447-513): five conv-pool stages of [2, 2, 4, 4, 4] 3x3 conv-BN-ReLU units with
    64/128/256/512/512 channels at full / 2 / 4 / 8 / 16 resolution before their 2x2 max pool, and a 1x1 conv to 1024 channels;
    dark3 / dark4 / dark5 = the outputs of stage 3, stage 4 and conv_add."""

    def __init__(self, layer=(2, 2, 4, 4, 4)):
        super().__init__()
        self.out_features = ("dark3", "dark4", "dark5")
        c = 64
        self.conv_pool1 = self._make_layer(3, c, layer[0])
        self.conv_pool2 = self._make_layer(c, c * 2, layer[1])
        self.conv_pool3 = self._make_layer(c * 2, c * 4, layer[2])
        self.conv_pool4 = self._make_layer(c * 4, c * 8, layer[3])
        self.conv_pool5 = self._make_layer(c * 8, c * 8, layer[4])
        self.conv_add = ConvBNReLU(c * 8, c * 16, kernel_size=1, bias=False)

    @staticmethod
    def _make_layer(cin, cout, n):
        layers = [ConvBNReLU(cin, cout, kernel_size=3, stride=1, padding=1, bias=False)]
        layers += [ConvBNReLU(cout, cout, kernel_size=3, stride=1, padding=1, bias=False) for _ in range(1, n)]
        layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        return nn.Sequential(*layers)

    def stages(self):
        return (self.conv_pool1, self.conv_pool2, self.conv_pool3, self.conv_pool4, self.conv_pool5)

    def forward(self, x):
        """images -> {"dark3", "dark4", "dark5"} (darknet.py:483-513)."""
        d3, d4, d5 = _sub(self, "backbone", x)
        return {"dark3": d3, "dark4": d4, "dark5": d5}


def vgg19():
    return VGG((2, 2, 4, 4, 4))


class YOLOPAFPN(nn.Module):
    """``backbone_type`` is the switch of yolox/models/yolo_pafpn.py:31-38 ('darknet' | 'resnet' | 'densenet' | 'vgg'; the 24p tree's own
    YOLOPAFPN hard-codes CSPDarknet, so it is a keyword here and the positional signature stays the 24p one)."""

    def __init__(self, depth=1.0, width=1.0, in_features=("dark3", "dark4", "dark5"), in_channels=[256, 512, 1024],
                 depthwise=False, act="silu", backbone_type="darknet"):
        super().__init__()
        if backbone_type == "darknet":
            self.backbone = CSPDarknet(depth, width, depthwise=depthwise, act=act)
        elif backbone_type == "resnet":
            if width != 1.0:
                raise NotImplementedError("resnet50() emits 256/512/1024 channels: it pairs with width 1.0 (BASELINE config 4)")
            self.backbone = resnet50()
        elif backbone_type == "densenet":
            if width != 1.0:
                raise NotImplementedError("densenet121() emits 256/512/1024 channels: it pairs with width 1.0 (BASELINE config 4)")
            self.backbone = densenet121()
        elif backbone_type == "vgg":
            if width != 1.0:
                raise NotImplementedError("vgg19() emits 256/512/1024 channels: it pairs with width 1.0")
            self.backbone = vgg19()
        else:
            raise NotImplementedError("backbone_type %r is not one of 'darknet', 'resnet', 'densenet', 'vgg'" % backbone_type)
        if depthwise and backbone_type != "darknet":
            raise NotImplementedError("depthwise=True belongs to the CSPDarknet network (yolo_pafpn.py:27-30)")
        self.backbone_type = backbone_type
        self.in_features = in_features
        self.in_channels = in_channels
        c3, c4, c5 = [int(c * width) for c in in_channels]
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        Conv, dw = _conv_cls(depthwise), depthwise                                    # yolo_pafpn.py:30
        self.lateral_conv0 = BaseConv(c5, c4, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c4, c4, n, False, depthwise=dw, act=act)
        self.reduce_conv1 = BaseConv(c4, c3, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c3, c3, n, False, depthwise=dw, act=act)
        self.bu_conv2 = Conv(c3, c3, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c3, c4, n, False, depthwise=dw, act=act)
        self.bu_conv1 = Conv(c4, c4, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c4, c5, n, False, depthwise=dw, act=act)

    def forward(self, x):
        """images -> (pan_out2, pan_out1, pan_out0), the tuple the head consumes (yolo_pafpn.py:83-124)."""
        return tuple(_sub(self, "pafpn", x))


class YOLOXHead(nn.Module):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu",
                 depthwise=False):
        super().__init__()
        self.n_anchors = 1
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.strides = strides
        self.use_l1 = False
        h = int(256 * width)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        Conv = _conv_cls(depthwise)                                                   # yolo_head_24p.py:45
        self.depthwise = bool(depthwise)
        for c in in_channels:
            self.stems.append(BaseConv(int(c * width), h, 1, 1, act=act))
            self.cls_convs.append(nn.Sequential(Conv(h, h, 3, 1, act=act), Conv(h, h, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(Conv(h, h, 3, 1, act=act), Conv(h, h, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(h, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(h, 26, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(h, self.n_anchors, 1, 1, 0))

    def initialize_biases(self, prior_prob):
        """bias = -log((1-p)/p) on the class and objectness predictors (yolo_head_24p.py:132-141)."""
        v = -math.log((1 - prior_prob) / prior_prob)
        with torch.no_grad():
            for conv in list(self.cls_preds) + list(self.obj_preds):
                conv.bias.fill_(v)

    def forward(self, xin, train=False):
        """xin: the three PAFPN outputs.  train=True: the 5-tuple (x_shifts, y_shifts, expanded_strides, outputs [B,A,27+C],
        origin_preds) of yolo_head_24p.py:143-189,205-206; train=False: decoded predictions with sigmoid scores (:190-210)."""
        from .engine import run_submodule
        out = run_submodule(self, "head", tuple(xin), train)[0]
        if not train:
            return out
        eng = self.__dict__["_ep24_sub"][("head", tuple(tuple(x.shape) for x in xin), getattr(self, "compute_dtype", torch.bfloat16))]
        origin_preds, a0 = [], 0
        if getattr(self, "use_l1", False):
            for H, W, _ in eng.levels:
                origin_preds.append(eng.origin[:, a0:a0 + H * W].clone())
                a0 += H * W
        return eng.x_shifts, eng.y_shifts, eng.exp_strides, out, origin_preds


class YOLOX(nn.Module):
    """``forward(x, train=True)`` returns the reference's train-mode 5-tuple
    ``(x_shifts[3], y_shifts[3], expanded_strides[3], outputs[B,A,27+C], origin_preds=[])`` (yolox.py:24-34,
    yolo_head_24p.py:143-210); ``train=False`` returns decoded predictions with sigmoid scores."""

    def __init__(self, backbone=None, head=None):
        super().__init__()
        self.backbone = YOLOPAFPN() if backbone is None else backbone
        self.head = YOLOXHead(80) if head is None else head
        self._engines = {}
        self.compute_dtype = torch.bfloat16

    def set_compute_dtype(self, dtype):
        """torch.bfloat16: the product path.  torch.float32: the fp32 parity mode of the same plan (the reference trains
        in fp32, train_24p.py:86-104) - what end-to-end comparisons with the CPU oracle run; not a fast path."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise _lib.Ep24Error("ep24: compute dtype is torch.bfloat16 or torch.float32")
        self.compute_dtype = dtype
        return self

    def engine(self, batch, size, dtype=None):
        """The launch plan for [batch, 3, size, size] inputs; ``dtype=torch.float32`` selects the fp32 parity mode."""
        from .engine import Engine
        dtype = dtype or torch.bfloat16
        if not isinstance(size, int) and size[0] == size[1]:
            size = int(size[0])
        key = (batch, size) if dtype == torch.bfloat16 else (batch, size, dtype)
        if key not in self._engines:
            # one parameter home (flat buffers) per model; plans for other input shapes share it
            self._engines[key] = Engine(self, batch, size, dtype)
        return self._engines[key]

    def forward(self, x, train=False):
        _lib.require_gpu()
        if not x.is_cuda:
            raise _lib.Ep24Error("ep24: input images must live on the GPU (no CPU fallback on the product path)")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise IndexError("expected images [B,3,H,W] with H and W multiples of 32, got %s" % (tuple(x.shape),))
        eng = self.engine(x.shape[0], (x.shape[2], x.shape[3]), self.compute_dtype)
        return eng.run_module_forward(x, train)
