"""PyTorch-ROCm backbone / neck / semantic head with the reference's parameter names, so its checkpoints load
unchanged: ResNet (mmdet/models/backbones/resnet.py), FPN (mmdet/models/necks/fpn.py), UPSNetFPN
(mmdet/models/panoptic/upsnetFPN.py). These stay framework code by design (north
star); only the deformable convolution inside UPSNetFPN runs on a HIP kernel (K7)."""
import torch
import torch.nn.functional as F
from torch import nn

from .dcn import DeformConvWithOffset
from .registry import BACKBONES, NECKS, PANOPTIC
from .slot_head import ConvModule as Conv1x1Module


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch"):
        super().__init__()
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style="pytorch", frozen_stages=-1, norm_eval=True, **unused):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f"invalid depth {depth} for resnet")
        self.out_indices = out_indices
        self.norm_eval, self.frozen_stages = norm_eval, frozen_stages
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(self.arch_settings[depth][:num_stages]):
            planes = 64 * 2 ** i
            blocks = []
            for b in range(nb):
                stride = strides[i] if b == 0 else 1
                down = None
                if b == 0 and (stride != 1 or inplanes != planes * 4):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                         nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride, dilations[i], down, style))
                inplanes = planes * 4
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)

    def init_weights(self, pretrained=None):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _ConvModule(nn.Module):
    """conv (+ bias) held as `.conv`: the reference's ConvModule without norm / activation."""

    def __init__(self, i, o, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(i, o, k, padding=padding)

    def forward(self, x):
        return self.conv(x)


@NECKS.register_module
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False, **unused):
        super().__init__()
        if add_extra_convs:
            raise NotImplementedError("add_extra_convs is not used by the Slot-VPS configs")
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(_ConvModule(in_channels[i], out_channels, 1))
            self.fpn_convs.append(_ConvModule(out_channels, out_channels, 3, padding=1))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [lc(inputs[i + self.start_level]) for i, lc in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], scale_factor=2, mode="nearest")
        outs = [self.fpn_convs[i](laterals[i]) for i in range(len(laterals))]
        for _ in range(self.num_outs - len(outs)):
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)


@PANOPTIC.register_module
class UPSNetFPN(nn.Module):
    """Shared 3 x [DeformConv3x3 + GroupNorm(32) + ReLU] tower on P2..P5; returns the 128-channel per-level maps
    coarse -> fine for the slot head and the 19-class semantic logits (upsnetFPN.py:36-73)."""

    def __init__(self, in_channels, out_channels, num_levels, num_things_classes, num_classes, ignore_label,
                 loss_weight, conv_cfg=None, norm_cfg=None, return_feat_levels=4):
        super().__init__()
        self.in_channels, self.out_channels, self.num_levels = in_channels, out_channels, num_levels
        self.num_things_classes, self.num_classes = num_things_classes, num_classes
        self.ignore_label, self.loss_weight = ignore_label, loss_weight
        self.deform_convs = nn.ModuleList([nn.Sequential(
            DeformConvWithOffset(in_channels, in_channels, kernel_size=3, padding=1), nn.GroupNorm(32, in_channels),
            nn.ReLU(inplace=True),
            DeformConvWithOffset(in_channels, out_channels, kernel_size=3, padding=1), nn.GroupNorm(32, out_channels),
            nn.ReLU(inplace=True),
            DeformConvWithOffset(out_channels, out_channels, kernel_size=3, padding=1), nn.GroupNorm(32, out_channels),
            nn.ReLU(inplace=True))])
        self.return_feat_levels = return_feat_levels
        self.conv_pred = _ConvModule(out_channels * 4, num_classes, 1)
        self.upsample = nn.Upsample(scale_factor=4, mode="bilinear", align_corners=False)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m is not None:
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for m in self.modules():                       # the offset convs stay zero (deform_conv_with_offset.py:25-26)
            if isinstance(m, DeformConvWithOffset):
                m.conv_offset.weight.data.zero_()
                m.conv_offset.bias.data.zero_()

    fuse_offset = True      # the offset convolutions of the tower on its pixel-major rows (csrc/offset_conv.hip: LDS halo tiles, 1.8 ms against the
    #                         framework's 3.0 per clip) - and no NCHW copy of the intermediate layers any more
    fuse_pred = True        # the prediction layer (three upsamplings + concat + 1x1 conv) as one kernel (csrc/semantic_pred.hip)
    fuse_norm = True        # pixel-major tower: K7' -> GroupNorm + ReLU (csrc/gn_relu.hip) without layout copies between the layers
    emit_pm16 = None        # torch.bfloat16 / torch.float16 / "hl" (two fp16 planes hi + lo): the last layer also writes its output as 16-bit pixel-major rows, kept in
    #                         `last_pm16` (per returned level, coarse -> fine; None where the fused tower did not run): what K4 reads
    #                         when the detector folds conv_trans into K4's weights (VPS_Temporal_Slots.fold_trans)
    last_pm16 = None

    def _tower(self, x):
        """The shared tower on one level. Where every layer runs K7' (fp32-class, C % 64 == 0, 128 / 256 output channels) the
        activations stay pixel-major between the layers: K7' reads and writes that layout, GroupNorm + ReLU is the library's kernel
        on it (which also emits the NCHW copy the next layer's offset convolution and the tower's consumers take). Otherwise: the
        module sequence as it stands (reference structure, upsnetFPN.py:36-49)."""
        from . import ops
        from .dcn import deform_conv_fused_pm, fused_applicable
        seq = list(self.deform_convs[0])
        ok = (self.fuse_norm and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and len(seq) % 3 == 0)
        if ok:
            for i in range(0, len(seq), 3):
                dc, gn, act = seq[i], seq[i + 1], seq[i + 2]
                ok = ok and (isinstance(dc, DeformConvWithOffset) and isinstance(gn, nn.GroupNorm) and isinstance(act, nn.ReLU)
                             and dc.conv.fused and not dc.conv.bf16_operands
                             and fused_applicable(x, dc.conv.weight, dc.conv.stride, dc.conv.padding, dc.conv.dilation, dc.conv.groups,
                                                  dc.conv.deformable_groups)
                             and dc.conv.stride == (1, 1) and dc.conv.padding == (1, 1) and dc.conv.dilation == (1, 1)
                             and gn.num_channels % 4 == 0 and 256 % (gn.num_channels // 4) == 0)
        if not ok:
            return self.deform_convs[0](x), None
        N, _, H, W = x.shape
        cur_nchw = x.contiguous()
        cur_pm = ops.nchw_to_pixel_major(cur_nchw) if (x.shape[1] % 4 == 0 and 256 % (x.shape[1] // 4) == 0) \
            else x.permute(0, 2, 3, 1).contiguous()                       # [N, H, W, C]
        for i in range(0, len(seq), 3):
            dc, gn = seq[i], seq[i + 1]
            O = dc.conv.weight.shape[0]
            co = dc.conv_offset
            if self.fuse_offset and co.weight.shape[0] <= 32 and co.weight.shape[1] % 32 == 0:
                # the offset convolution on the pixel-major rows themselves (csrc/offset_conv.hip): no NCHW copy of the layer's input
                if getattr(co, "_svps_pack_key", None) != (co.weight.data_ptr(), co.weight._version):
                    co._svps_pack, co._svps_pack_key = ops.pack_conv3x3_small(co.weight), (co.weight.data_ptr(), co.weight._version)
                off = ops.conv3x3_pm_small(cur_pm, co._svps_pack, co.bias, co.weight.shape[0])
            else:
                if cur_nchw is None:
                    cur_nchw = cur_pm.permute(0, 3, 1, 2).contiguous()
                off = co(cur_nchw)                                        # framework 3 x 3 convolution (NCHW in, [N, 18, H, W] out)
            # [N, HW, O] + the GroupNorm's per-channel sums from the kernel's epilogue (no moments pass over y)
            y, st = deform_conv_fused_pm(cur_pm, off, dc.conv._weight_pack(), O, 1, 1, 1, gn_stats=True)
            if i + 3 >= len(seq) and self.emit_pm16 is not None:          # last layer: NCHW for the semantic logits, 16-bit rows for K4
                _, y_nchw, y16 = ops.group_norm_relu_pm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, want_nchw=True,
                                                        want_16=self.emit_pm16, want_pm=False, stats=st)
                return y_nchw.view(N, O, H, W), y16
            last = i + 3 >= len(seq)
            need_nchw = last or not self.fuse_offset                      # the NCHW copy only fed the framework's offset convolution
            y_pm, y_nchw = ops.group_norm_relu_pm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, want_nchw=need_nchw, stats=st)
            cur_pm, cur_nchw = y_pm.view(N, H, W, O), (y_nchw.view(N, O, H, W) if y_nchw is not None else None)
        return cur_nchw, None

    def forward(self, inputs):
        assert len(inputs) == self.num_levels
        both = [self._tower(inputs[i]) for i in range(self.num_levels)]
        px = [b[0] for b in both]
        order = [3, 2, 1, 0] if self.return_feat_levels == 4 else [2, 1, 0]
        feat_before = [px[i] for i in order]
        self.last_pm16 = [both[i][1] for i in order] if all(both[i][1] is not None for i in order) else None
        conv = self.conv_pred.conv
        if (self.fuse_pred and self.num_levels == 4 and px[0].is_cuda and px[0].dtype == torch.float32 and not torch.is_autocast_enabled()
                and conv.weight.shape[0] <= 32 and px[0].shape[-2] % 8 == 0 and px[0].shape[-1] % 8 == 0
                and all(px[i].shape[-2:] == (px[0].shape[-2] >> i, px[0].shape[-1] >> i) and px[i].is_contiguous() for i in range(4))):
            from . import ops
            fcn_score = ops.semantic_pred(px, conv.weight, conv.bias)      # upsampling x 3 + concat + 1x1 conv in one kernel
        else:
            ups = [px[0]] + [F.interpolate(px[i], None, 2 ** i, mode="bilinear", align_corners=False) for i in (1, 2, 3)]
            fcn_score = self.conv_pred(torch.cat(ups, dim=1))
        return self.upsample(fcn_score), fcn_score, feat_before
