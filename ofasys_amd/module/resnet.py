"""ResNet feature extractor of the image_resnet / video adaptors (reference: module/resnet.py:85-261: conv1-bn-relu-
maxpool-layer1..3 of a torchvision-style Bottleneck ResNet, **no layer4**, output stride 16).

Same module tree and parameter / buffer names as the reference (nn.Conv2d and nn.BatchNorm2d are kept as parameter
containers, so checkpoints interchange), but the arithmetic runs on NHWC rows through the gfx950 kernels:
1x1 convolutions are plain MFMA GEMMs, the others im2col + GEMM, BatchNorm (+ residual) (+ ReLU) one fused pass over
per-column statistics (csrc/conv.hip)."""
import torch
import torch.nn as nn

from .. import ops
from .layers import DropPath

__all__ = ["resnet50_backbone", "resnet101_backbone", "resnet152_backbone"]


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    if groups != 1 or dilation != 1:
        raise NotImplementedError("grouped / dilated convolutions are not used by OFASys' backbones")
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None,
                 drop_path_rate=0.0):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.drop_path = DropPath(drop_path_rate, 0)          # per-sample stochastic depth of the residual branch (no parameters)

    def forward(self, fm):
        """fm = (rows [B*H*W, C], B, H, W)   -- module/resnet.py:112-137."""
        x, B, H, W = fm
        # an identity block reads x twice (conv1 and the residual add): the residual's gradient is handed to conv1's input-gradient
        # GEMM, which accumulates onto it, instead of a separate add of two [B*H*W, C] tensors per block (ops.batch_norm)
        dropping = self.training and self.drop_path.drop_prob > 0.0
        mailbox = [] if (self.downsample is None and torch.is_grad_enabled() and x.requires_grad and not dropping) else None
        # conv_bn: in training the BatchNorm's statistics come out of the convolution's GEMM epilogue (no pass over its output)
        out, _, _ = ops.conv_bn(x, self.conv1, self.bn1, B, H, W, relu=True, conv_mailbox=mailbox)
        out, Ho, Wo = ops.conv_bn(out, self.conv2, self.bn2, B, H, W, relu=True)
        identity = x
        if self.downsample is not None:
            identity, _, _ = ops.conv_bn(x, self.downsample[0], self.downsample[1], B, H, W)
        if dropping:                                          # relu(identity + drop_path(bn3(out))), module/resnet.py:127-135
            out, _, _ = ops.conv_bn(out, self.conv3, self.bn3, B, Ho, Wo)
            out = self.drop_path(out.view(B, Ho * Wo, out.shape[1])).reshape(B * Ho * Wo, out.shape[1])
            return ops.relu(ops.dropout_add(out, identity, 0.0, False)), B, Ho, Wo
        out, _, _ = ops.conv_bn(out, self.conv3, self.bn3, B, Ho, Wo, relu=True, residual=identity, bn_mailbox=mailbox)   # relu(identity + bn3(out))
        return out, B, Ho, Wo


class ResNet(nn.Module):
    def __init__(self, layers, zero_init_residual=False, groups=1, width_per_group=64, replace_stride_with_dilation=None,
                 norm_layer=None, drop_path_rate=0.0):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation not in (None, [False, False, False]):
            raise NotImplementedError("dilated ResNet stages are not used by OFASys")
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(Bottleneck, 64, layers[0], drop_path_rate=drop_path_rate)
        self.layer2 = self._make_layer(Bottleneck, 128, layers[1], stride=2, drop_path_rate=drop_path_rate)
        self.layer3 = self._make_layer(Bottleneck, 256, layers[2], stride=2, drop_path_rate=drop_path_rate)
        for m in self.modules():                                               # module/resnet.py:180-185
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.SyncBatchNorm, nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False, drop_path_rate=0.0):
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, self.dilation, norm_layer)]
        self.inplanes = planes * block.expansion
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, blocks)]
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation,
                                norm_layer=norm_layer, drop_path_rate=dpr[i]))
        return nn.Sequential(*layers)

    def _apply(self, fn, *args, **kwargs):
        """model.bfloat16() / .half() must not round the BatchNorm running statistics: they are fp32 accumulators updated
        with momentum 0.1 every step (the reference lets .to(bf16) cast them; loading either dtype works)."""
        keep = [(m, m.running_mean, m.running_var) for m in self.modules()
                if isinstance(m, nn.BatchNorm2d) and m.running_mean is not None]
        super()._apply(fn, *args, **kwargs)
        for m, rm, rv in keep:
            if m.running_mean.dtype != torch.float32:
                m.running_mean = rm.float().to(m.running_mean.device)
                m.running_var = rv.float().to(m.running_var.device)
        return self

    def forward(self, x):
        """x: [B, 3, H, W] image -> (rows [B*h*w, 1024] in (b, h, w) order, h, w); the reference returns [B,1024,h,w] and
        its caller immediately flattens to [B, h*w, 1024] (adaptor/image_resnet.py:159) -- the same rows."""
        B, _, H, W = x.shape
        with ops.bn_scope():
            out, H, W = ops.conv_bn(x, self.conv1, self.bn1, B, H, W, relu=True, nchw=True)
            out, H, W = ops.max_pool(out, B, H, W, self.maxpool.kernel_size, self.maxpool.stride, self.maxpool.padding)
            fm = (out, B, H, W)
            for layer in (self.layer1, self.layer2, self.layer3):
                for blk in layer:
                    fm = blk(fm)
        out, B, H, W = fm
        return out, H, W


def resnet50_backbone(norm_layer=None, drop_path_rate=0.0) -> ResNet:
    return ResNet([3, 4, 6], norm_layer=norm_layer, drop_path_rate=drop_path_rate)


def resnet101_backbone(norm_layer=None, drop_path_rate=0.0) -> ResNet:
    return ResNet([3, 4, 23], norm_layer=norm_layer, drop_path_rate=drop_path_rate)


def resnet152_backbone(norm_layer=None, drop_path_rate=0.0) -> ResNet:
    return ResNet([3, 8, 36], norm_layer=norm_layer, drop_path_rate=drop_path_rate)
