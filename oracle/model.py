"""Oracle: fp32 torch-CPU YOLOX-24p network.  Test infrastructure only.

A compact restatement of the reference graph with the reference's state-dict key names, so golden weights
load by name:
  conv-BN-act block / Focus / Bottleneck / CSP / SPP   yolox_24p/models/network_blocks.py:29-210
  CSPDarknet(dep_mul, wid_mul)                          yolox_24p/models/darknet.py:95-177
  PAFPN neck                                            yolox_24p/models/yolo_pafpn.py:27-124
  decoupled 24p head + train-mode decode                yolox_24p/models/yolo_head_24p.py:47-237
  factory (BN eps 1e-3 / momentum 0.03, prior bias)     yolox_24p/exp/yolox_base.py:55-72
  swapped backbone resnet50() (BASELINE config 4)       yolox_24p/models/darknet.py:179-429, yolox/models/yolo_pafpn.py:31-38
  depthwise variants (DWConv = depthwise unit + 1x1)    yolox_24p/models/network_blocks.py:57-76; the `depthwise` switches of
                                                        darknet.py:107, network_blocks.py:92, yolo_pafpn.py:30, yolo_head_24p.py:45
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


EMULATE_BF16 = False      # tests flip this to mirror the product's storage precision (see Unit.forward)


def _q(x):
    """Round to bf16 and back with a straight-through gradient."""
    return x + (x.to(torch.bfloat16).float() - x).detach()


class Unit(nn.Module):
    """conv (no bias) -> BatchNorm -> SiLU; attribute names conv / bn as in BaseConv.

    With EMULATE_BF16 the same fp32 arithmetic is applied to bf16-rounded operands at the points where the
    HIP plan stores bf16 (conv inputs, packed weights, raw conv output, activated output) while the batch
    statistics come from the unrounded conv result, exactly as the MFMA epilogue accumulates them.  This
    isolates kernel errors from the (expected) bf16 storage error when the two paths are compared."""

    def __init__(self, cin, cout, k, s=1, groups=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, (k - 1) // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03)

    def forward(self, x):
        return conv_bn_act(x, self.conv, self.bn, "silu", self.training)


class DW(nn.Module):
    """DWConv (network_blocks.py:57-76): a depthwise unit (groups = channels) followed by a 1x1 unit; names dconv / pconv."""

    def __init__(self, cin, cout, k, s=1):
        super().__init__()
        self.dconv = Unit(cin, cin, k, s, groups=cin)
        self.pconv = Unit(cin, cout, 1, 1)

    def forward(self, x):
        return self.pconv(self.dconv(x))


def Conv(depthwise):
    """`Conv = DWConv if depthwise else BaseConv` of the reference's constructors."""
    return DW if depthwise else Unit


def conv_bn_act(x, conv, bn, act, training, residual=None):
    """conv -> BatchNorm -> (+ residual) -> activation, plain or with the product's bf16 storage points emulated."""
    fn = {"silu": F.silu, "relu": F.relu, None: lambda v: v}[act]
    if not EMULATE_BF16:
        u = bn(conv(x))
        return fn(u if residual is None else u + residual)
    z = F.conv2d(_q(x), _q(conv.weight), None, conv.stride, conv.padding, 1, conv.groups)
    mean = z.mean((0, 2, 3))
    var = z.var((0, 2, 3), unbiased=False)
    if training:
        with torch.no_grad():
            n = z.numel() / z.shape[1]
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(m * mean)
            bn.running_var.mul_(1 - m).add_(m * var * n / max(n - 1, 1))
            bn.num_batches_tracked += 1
    scale = bn.weight / torch.sqrt(var + bn.eps)
    u = _q(z) * scale.view(1, -1, 1, 1) + (bn.bias - mean * scale).view(1, -1, 1, 1)
    if residual is not None:
        u = _q(u + residual)                       # the sum is stored (bf16) before the in-place ReLU
    return _q(fn(u))


class Stem(nn.Module):
    """Focus: 2x2 space-to-depth in the order TL, BL, TR, BR, then a conv unit (network_blocks.py:188-210)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = Unit(4 * cin, cout, k)

    def forward(self, x):
        parts = [x[..., 0::2, 0::2], x[..., 1::2, 0::2], x[..., 0::2, 1::2], x[..., 1::2, 1::2]]
        return self.conv(torch.cat(parts, 1))


class Res(nn.Module):
    def __init__(self, c, add, depthwise=False):
        super().__init__()
        self.conv1 = Unit(c, c, 1)
        self.conv2 = Conv(depthwise)(c, c, 3)
        self.add = add

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.add else y


class CSP(nn.Module):
    def __init__(self, cin, cout, n, add=True, depthwise=False):
        super().__init__()
        h = int(cout * 0.5)
        self.conv1 = Unit(cin, h, 1)
        self.conv2 = Unit(cin, h, 1)
        self.conv3 = Unit(2 * h, cout, 1)
        self.m = nn.Sequential(*[Res(h, add, depthwise) for _ in range(n)])

    def forward(self, x):
        return self.conv3(torch.cat((self.m(self.conv1(x)), self.conv2(x)), 1))


class SPP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = Unit(cin, cin // 2, 1)
        self.conv2 = Unit(cin // 2 * 4, cout, 1)

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1))


class Backbone(nn.Module):
    def __init__(self, depth, width, depthwise=False):
        super().__init__()
        c = int(width * 64)
        d = max(round(depth * 3), 1)
        C, dw = Conv(depthwise), depthwise
        self.stem = Stem(3, c, 3)
        self.dark2 = nn.Sequential(C(c, 2 * c, 3, 2), CSP(2 * c, 2 * c, d, depthwise=dw))
        self.dark3 = nn.Sequential(C(2 * c, 4 * c, 3, 2), CSP(4 * c, 4 * c, 3 * d, depthwise=dw))
        self.dark4 = nn.Sequential(C(4 * c, 8 * c, 3, 2), CSP(8 * c, 8 * c, 3 * d, depthwise=dw))
        self.dark5 = nn.Sequential(C(8 * c, 16 * c, 3, 2), SPP(16 * c, 16 * c), CSP(16 * c, 16 * c, d, add=False, depthwise=dw))

    def forward(self, x):
        x = self.dark2(self.stem(x))
        c3 = self.dark3(x)
        c4 = self.dark4(c3)
        return c3, c4, self.dark5(c4)


class ResBlock(nn.Module):
    """Bottleneck of the swapped backbone (darknet.py:230-271)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idn = x
        if self.downsample is not None:
            idn = conv_bn_act(x, self.downsample[0], self.downsample[1], None, self.training)
        t = conv_bn_act(x, self.conv1, self.bn1, "relu", self.training)
        t = conv_bn_act(t, self.conv2, self.bn2, "relu", self.training)
        return conv_bn_act(t, self.conv3, self.bn3, "relu", self.training, residual=idn)     # out += identity; relu (:266-268)


class ResNetBackbone(nn.Module):
    """resnet50() of darknet.py:274-429: inplanes 32, stages [3,4,6,3] x planes 32/64/128/256, outputs layer2/3/4.
    fc / baseconv1..3 are constructed by the reference and never run; they are here so state dicts match."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv1 = nn.Conv2d(3, 32, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.layer1 = self._stage(32, 3, 1)
        self.layer2 = self._stage(64, 4, 2)
        self.layer3 = self._stage(128, 6, 2)
        self.layer4 = self._stage(256, 3, 2)
        self.fc = nn.Linear(2048, 1000)                         # 512 * expansion, as written (never run)
        for i, (a, b) in enumerate(((512, 128), (1024, 256), (2048, 256))):
            setattr(self, "baseconv%d" % (i + 1), nn.Sequential(nn.Conv2d(a, b, 1, bias=False), nn.BatchNorm2d(b), nn.SiLU()))

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [ResBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [ResBlock(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = conv_bn_act(x, self.conv1, self.bn1, "relu", self.training)
        x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer1(x)
        c3 = self.layer2(x)
        c4 = self.layer3(c3)
        return c3, c4, self.layer4(c4)


def bn_act_conv(x, bn, conv, training):
    """Pre-activation block of DenseNet (darknet.py:532-543): BatchNorm over the (concatenated) input -> ReLU -> conv, with
    the product's bf16 storage points emulated on request (the normalised, rectified tensor is stored; the conv output is
    stored raw)."""
    if not EMULATE_BF16:
        return conv(F.relu(bn(x)))
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    if training:
        with torch.no_grad():
            n = x.numel() / x.shape[1]
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(m * mean)
            bn.running_var.mul_(1 - m).add_(m * var * n / max(n - 1, 1))
            bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean, bn.running_var
    scale = bn.weight / torch.sqrt(var + bn.eps)
    a = _q(F.relu(x * scale.view(1, -1, 1, 1) + (bn.bias - mean * scale).view(1, -1, 1, 1)))
    return _q(F.conv2d(a, _q(conv.weight), None, conv.stride, conv.padding))


class DenseNetBackbone(nn.Module):
    """densenet121() of darknet.py:515-674 with the reference's state-dict names.  ``keep`` (optional, [n_layers, B, 32] of 0
    or 1/(1-p)) replaces the Dropout2d draws, so that training-mode runs are reproducible (the reference's masks are
    captured by the golden generator and handed to both sides)."""

    def __init__(self, blocks=(6, 12, 24, 16)):
        super().__init__()

        def cb(cin, cout, k):
            m = nn.Module()
            m.bn, m.relu, m.conv = nn.BatchNorm2d(cin), nn.ReLU(), nn.Conv2d(cin, cout, k, 1, k // 2, bias=False)
            return m

        def dn(cin, cout, k, s=1):
            m = nn.Module()
            m.conv, m.bn, m.relu = nn.Conv2d(cin, cout, k, s, k // 2, bias=False), nn.BatchNorm2d(cout, eps=0.001), nn.ReLU()
            return m

        def block(n, cin):
            m = nn.Module()
            layers = []
            for i in range(n):
                lay = nn.Module()
                lay.conv_block = nn.Sequential(cb(cin + 32 * i, 128, 1), cb(128, 32, 3))
                lay.dropout = nn.Dropout2d(0.3)
                layers.append(lay)
            m.denseblock = nn.Sequential(*layers)
            return m

        def trans(cin):
            m = nn.Module()
            m.trans = nn.Sequential(cb(cin, cin // 2, 1), nn.AvgPool2d(2, 2))
            return m

        self.stem = nn.Sequential(dn(3, 64, 7, 2), nn.MaxPool2d(3, 2, 1))
        t1 = 64 + blocks[0] * 32
        t2 = t1 // 2 + blocks[1] * 32
        t3 = t2 // 2 + blocks[2] * 32
        self.D1, self.T1 = block(blocks[0], 64), trans(t1)
        self.D2, self.T2 = block(blocks[1], t1 // 2), trans(t2)
        self.D3, self.T3 = block(blocks[2], t2 // 2), trans(t3)
        self.D4 = block(blocks[3], t3 // 2)
        self.baseconv1, self.baseconv2 = dn(t2, t2 // 2, 1), dn(t3, t3 // 2, 1)
        self.keep = None

    def _block(self, blk, x, base):
        for i, lay in enumerate(blk.denseblock):
            c1, c2 = lay.conv_block
            z1 = bn_act_conv(x, c1.bn, c1.conv, self.training)
            z2 = bn_act_conv(z1, c2.bn, c2.conv, self.training)
            if self.training:
                if self.keep is not None:
                    z2 = z2 * self.keep[base + i].view(z2.shape[0], 32, 1, 1)
                    z2 = _q(z2) if EMULATE_BF16 else z2
                else:
                    z2 = lay.dropout(z2)
            x = torch.cat((x, z2), 1)
        return x, base + len(blk.denseblock)

    def _trans(self, t, x):
        return F.avg_pool2d(bn_act_conv(x, t.trans[0].bn, t.trans[0].conv, self.training), 2, 2)

    def forward(self, x):
        st = self.stem[0]
        x = F.max_pool2d(conv_bn_act(x, st.conv, st.bn, "relu", self.training), 3, 2, 1)
        x, n = self._block(self.D1, x, 0)
        x, n = self._block(self.D2, self._trans(self.T1, x), n)
        c3 = conv_bn_act(x, self.baseconv1.conv, self.baseconv1.bn, "relu", self.training)
        x, n = self._block(self.D3, self._trans(self.T2, x), n)
        c4 = conv_bn_act(x, self.baseconv2.conv, self.baseconv2.bn, "relu", self.training)
        x, n = self._block(self.D4, self._trans(self.T3, x), n)
        return c3, c4, x


class VGGBackbone(nn.Module):
    """vgg19() of darknet.py:447-513 with the reference's state-dict names (conv_pool1..5.{i}.{conv,bn}, conv_add)."""

    def __init__(self, layer=(2, 2, 4, 4, 4)):
        super().__init__()

        def cbr(cin, cout, k):
            m = nn.Module()
            m.conv, m.bn, m.relu = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False), nn.BatchNorm2d(cout, eps=0.001), nn.ReLU()
            return m

        def stage(cin, cout, n):
            return nn.Sequential(*([cbr(cin, cout, 3)] + [cbr(cout, cout, 3) for _ in range(1, n)] + [nn.MaxPool2d(2, 2)]))
        self.conv_pool1, self.conv_pool2 = stage(3, 64, layer[0]), stage(64, 128, layer[1])
        self.conv_pool3, self.conv_pool4 = stage(128, 256, layer[2]), stage(256, 512, layer[3])
        self.conv_pool5 = stage(512, 512, layer[4])
        self.conv_add = cbr(512, 1024, 1)

    def _stage(self, st, x):
        for m in st:
            x = F.max_pool2d(x, 2, 2) if isinstance(m, nn.MaxPool2d) else conv_bn_act(x, m.conv, m.bn, "relu", self.training)
        return x

    def forward(self, x):
        x = self._stage(self.conv_pool2, self._stage(self.conv_pool1, x))
        c3 = self._stage(self.conv_pool3, x)
        c4 = self._stage(self.conv_pool4, c3)
        x = self._stage(self.conv_pool5, c4)
        return c3, c4, conv_bn_act(x, self.conv_add.conv, self.conv_add.bn, "relu", self.training)


class Neck(nn.Module):
    def __init__(self, depth, width, in_channels=(256, 512, 1024), backbone_type="darknet", depthwise=False):
        super().__init__()
        c3, c4, c5 = [int(c * width) for c in in_channels]
        n = round(3 * depth)
        dw = depthwise
        self.backbone = {"darknet": lambda: Backbone(depth, width, dw), "resnet": ResNetBackbone, "densenet": DenseNetBackbone, "vgg": VGGBackbone}[backbone_type]()
        self.lateral_conv0 = Unit(c5, c4, 1)
        self.C3_p4 = CSP(2 * c4, c4, n, add=False, depthwise=dw)
        self.reduce_conv1 = Unit(c4, c3, 1)
        self.C3_p3 = CSP(2 * c3, c3, n, add=False, depthwise=dw)
        self.bu_conv2 = Conv(dw)(c3, c3, 3, 2)
        self.C3_n3 = CSP(2 * c3, c4, n, add=False, depthwise=dw)
        self.bu_conv1 = Conv(dw)(c4, c4, 3, 2)
        self.C3_n4 = CSP(2 * c4, c5, n, add=False, depthwise=dw)

    def forward(self, x):
        x2, x1, x0 = self.backbone(x)
        f0 = self.lateral_conv0(x0)
        p4 = self.C3_p4(torch.cat([F.interpolate(f0, scale_factor=2, mode="nearest"), x1], 1))
        f1 = self.reduce_conv1(p4)
        out2 = self.C3_p3(torch.cat([F.interpolate(f1, scale_factor=2, mode="nearest"), x2], 1))
        out1 = self.C3_n3(torch.cat([self.bu_conv2(out2), f1], 1))
        out0 = self.C3_n4(torch.cat([self.bu_conv1(out1), f0], 1))
        return out2, out1, out0


class Head(nn.Module):
    def __init__(self, num_classes, width, in_channels=(256, 512, 1024), strides=(8, 16, 32), depthwise=False):
        super().__init__()
        h = int(256 * width)
        C = Conv(depthwise)
        self.num_classes = num_classes
        self.strides = strides
        self.stems = nn.ModuleList(Unit(int(c * width), h, 1) for c in in_channels)
        self.cls_convs = nn.ModuleList(nn.Sequential(C(h, h, 3), C(h, h, 3)) for _ in in_channels)
        self.reg_convs = nn.ModuleList(nn.Sequential(C(h, h, 3), C(h, h, 3)) for _ in in_channels)
        self.cls_preds = nn.ModuleList(nn.Conv2d(h, num_classes, 1) for _ in in_channels)
        self.reg_preds = nn.ModuleList(nn.Conv2d(h, 26, 1) for _ in in_channels)
        self.obj_preds = nn.ModuleList(nn.Conv2d(h, 1, 1) for _ in in_channels)
        prior = -math.log((1 - 1e-2) / 1e-2)
        for m in list(self.cls_preds) + list(self.obj_preds):
            nn.init.constant_(m.bias, prior)

    def forward(self, feats, train=False):
        outs, xs, ys, ss = [], [], [], []
        for k, x in enumerate(feats):
            x = self.stems[k](x)
            cf = self.cls_convs[k](x)
            rf = self.reg_convs[k](x)
            if EMULATE_BF16:
                pred = lambda m, f: F.conv2d(_q(f), _q(m.weight), m.bias)
                reg, obj, cls = pred(self.reg_preds[k], rf), pred(self.obj_preds[k], rf), pred(self.cls_preds[k], cf)
            else:
                reg, obj, cls = self.reg_preds[k](rf), self.obj_preds[k](rf), self.cls_preds[k](cf)
            if not train:
                outs.append(torch.cat([reg, obj.sigmoid(), cls.sigmoid()], 1).flatten(2))
                continue
            o = torch.cat([reg, obj, cls], 1)
            B, C, H, W = o.shape
            o = o.permute(0, 2, 3, 1).reshape(B, H * W, C)
            yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            grid = torch.stack((xv, yv), 2).reshape(1, -1, 2).to(o.dtype)
            s = self.strides[k]
            o = torch.cat([(o[..., :2] + grid) * s, torch.exp(o[..., 2:26]) * s, o[..., 26:]], -1)
            outs.append(o)
            xs.append(grid[:, :, 0])
            ys.append(grid[:, :, 1])
            ss.append(torch.full((1, H * W), float(s)))
        if train:
            return xs, ys, ss, torch.cat(outs, 1), []
        hw = [(f.shape[-2], f.shape[-1]) for f in feats]
        o = torch.cat(outs, 2).permute(0, 2, 1)
        grids, strides = [], []
        for (H, W), s in zip(hw, self.strides):
            yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            grids.append(torch.stack((xv, yv), 2).reshape(1, -1, 2))
            strides.append(torch.full((1, H * W, 1), float(s)))
        grids, strides = torch.cat(grids, 1).to(o.dtype), torch.cat(strides, 1).to(o.dtype)
        return torch.cat([(o[..., :2] + grids) * strides, torch.exp(o[..., 2:26]) * strides, o[..., 26:]], -1)


class Net(nn.Module):
    def __init__(self, depth=1.0, width=1.0, num_classes=80, backbone_type="darknet", depthwise=False):
        super().__init__()
        self.backbone = Neck(depth, width, backbone_type=backbone_type, depthwise=depthwise)
        self.head = Head(num_classes, width, depthwise=depthwise)
        for m in self.modules():                         # init_yolo (yolox_base.py:58-62) reaches every BatchNorm2d
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def forward(self, x, train=False):
        return self.head(self.backbone(x), train)


def sgd_nesterov_step(params, bufs, lr, momentum=0.9):
    """torch.optim.SGD(momentum, nesterov=True, no decay) update (yolox_base.py:120-124):
    buf = m*buf + g (buf = g on the first step); p -= lr*(g + m*buf)."""
    with torch.no_grad():
        for i, p in enumerate(params):
            g = p.grad
            if bufs[i] is None:
                bufs[i] = g.clone()
            else:
                bufs[i].mul_(momentum).add_(g)
            p.add_(g + momentum * bufs[i], alpha=-lr)
