"""Plain-PyTorch fp32 restatement of the reference TowerUNet hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). This is the CPU checker and
the ``cpu_baseline`` ("port") that travels to the GPU box; the reference's own
Python never does. It is validated against the stub-imported reference in this
container by tests/test_oracle_vs_reference.py (max |diff| <= 1e-6 on outputs,
loss and gradients) and against the committed fixtures in tests/golden.

Each class cites the reference file:line it restates (paths relative to
/root/reference/src/cultionet). Module attribute names are chosen so that
``state_dict()`` keys equal the reference's (442 tensors at the default
config), which is what lets both sides be filled from the same key-seeded
generator (``seeded_state_dict``).
"""
from __future__ import annotations

import math
import typing as T
import zlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .na2d_ref import NeighborhoodAttention2D

# unet_parts.py:19-40
NATTEN_PARAMS = {
    "a": dict(heads=4, kernel=3, dilation=2),
    "b": dict(heads=4, kernel=3, dilation=1),
    "c": dict(heads=8, kernel=3, dilation=1),
}


class _ToNHWC(nn.Module):
    def forward(self, x):
        return x.permute(0, 2, 3, 1)


class _ToNCHW(nn.Module):
    def forward(self, x):
        return x.permute(0, 3, 1, 2)


class _SqueezeT(nn.Module):
    def forward(self, x):
        return x.squeeze(2)


class Act(nn.Module):
    """nn/modules/activations.py:5-24 (SetActivation)."""

    def __init__(self, activation_type: str = "SiLU"):
        super().__init__()
        self.activation = getattr(nn, activation_type)()

    def forward(self, x):
        return self.activation(x)


def resize_to(x: torch.Tensor, size) -> torch.Tensor:
    """nn/functional.py:72-81 (check_upsample)."""
    if tuple(x.shape[-2:]) != tuple(size):
        x = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)
    return x


class TimeConv(nn.Module):
    """models/nunet.py:18-57 (Conv3d)."""

    def __init__(self, in_channels, in_time, out_channels, kernel_size, act="SiLU"):
        super().__init__()
        rest = in_time - kernel_size + 1
        self.seq = nn.Sequential(
            nn.Conv3d(in_channels, in_channels, (kernel_size, 1, 1), bias=False),
            nn.BatchNorm3d(in_channels),
            Act(act),
            nn.Conv3d(in_channels, out_channels, (rest, 1, 1), bias=False),
            _SqueezeT(),
            nn.BatchNorm2d(out_channels),
            Act(act),
        )

    def forward(self, x):
        return self.seq(x)


class PreTimeReduction(nn.Module):
    """models/nunet.py:60-105."""

    def __init__(self, in_channels, in_time, out_channels, act="SiLU"):
        super().__init__()
        self.conv3 = TimeConv(in_channels, in_time, out_channels, 3, act)
        self.conv5 = TimeConv(in_channels, in_time, out_channels, 5, act)
        self.layer_norm = nn.Sequential(_ToNHWC(), nn.LayerNorm(out_channels), _ToNCHW())

    def forward(self, x):
        return self.layer_norm(self.conv3(x) + self.conv5(x))


class ConvBlock2d(nn.Module):
    """nn/modules/convolution.py:71-120."""

    def __init__(self, cin, cout, kernel_size, padding=0, dilation=1, stride=1, add_activation=True, act="SiLU",
                 batchnorm_first=False):
        super().__init__()
        if batchnorm_first:
            layers = [nn.BatchNorm2d(cin), Act(act),
                      nn.Conv2d(cin, cout, kernel_size, padding=padding, dilation=dilation, stride=stride)]
        else:
            layers = [
                nn.Conv2d(cin, cout, kernel_size, padding=padding, dilation=dilation, stride=stride, bias=False),
                nn.BatchNorm2d(cout),
            ]
            if add_activation:
                layers.append(Act(act))
        self.seq = nn.Sequential(*layers)

    def forward(self, x):
        return self.seq(x)


class ResConvBlock2d(nn.Module):
    """nn/modules/convolution.py:123-176."""

    def __init__(self, cin, cout, kernel_size=3, dilation=1, act="SiLU", num_blocks=2, batchnorm_first=False):
        super().__init__()
        bf = batchnorm_first
        blocks = [ConvBlock2d(cin, cout, kernel_size, padding=0 if kernel_size == 1 else kernel_size // 2, act=act,
                              batchnorm_first=bf)]
        for _ in range(num_blocks - 1):
            d = 1 if kernel_size == 1 else max(1, dilation - 1)
            blocks.append(ConvBlock2d(cout, cout, kernel_size, padding=0 if kernel_size == 1 else d, dilation=d, act=act,
                                      batchnorm_first=bf))
        self.block = nn.ModuleList(blocks)

    def forward(self, x):
        for layer in self.block:
            x = layer(x)
        return x


class ChannelAttention(nn.Module):
    """nn/modules/attention.py:12-62."""

    def __init__(self, cin, act):
        super().__init__()
        mk = lambda: nn.Sequential(nn.Conv2d(cin, cin // 2, 1, bias=False), Act(act), nn.Conv2d(cin // 2, cin, 1, bias=False))
        self.fc1 = mk()
        self.fc2 = mk()

    def forward(self, x):
        avg = self.fc1(x.mean(dim=(2, 3), keepdim=True))
        mx = self.fc2(x.amax(dim=(2, 3), keepdim=True))
        return torch.sigmoid(avg + mx).expand(-1, -1, x.shape[2], x.shape[3])


class SpatialAttention(nn.Module):
    """nn/modules/attention.py:65-86."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, 3, padding=1, bias=False)

    def forward(self, x):
        a = torch.cat([x.mean(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)], dim=1)
        return torch.sigmoid(self.conv(a)).expand(-1, x.shape[1], -1, -1)


class SpatialChannelAttention(nn.Module):
    """nn/modules/attention.py:89-126."""

    def __init__(self, cin, act):
        super().__init__()
        self.channel_attention = ChannelAttention(cin, act)
        self.spatial_attention = SpatialAttention()
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return 1.0 + self.gamma * ((self.channel_attention(x) + self.spatial_attention(x)) * 0.5)


class ResidualConv(nn.Module):
    """nn/modules/convolution.py:179-247 (attention None)."""

    def __init__(self, cin, cout, kernel_size=3, num_blocks=2, attention_weights=None, act="SiLU",
                 batchnorm_first=False, **_unused):
        super().__init__()
        assert attention_weights is None
        self.seq = ResConvBlock2d(cin, cout, kernel_size, act=act, num_blocks=num_blocks, batchnorm_first=batchnorm_first)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        out = self.skip(x) if self.skip is not None else x
        return out + self.seq(x)


class ResidualAConv(nn.Module):
    """nn/modules/convolution.py:250-395 (attention None | natten)."""

    def __init__(self, cin, cout, kernel_size=3, num_blocks=2, dilations=None, attention_weights=None,
                 act="SiLU", heads=8, kernel=3, dilation=1, batchnorm_first=False, na_drop=0.0):
        super().__init__()
        if dilations is None:
            dilations = [1, 2]
        self.attention_weights = attention_weights
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()
        if attention_weights is not None:
            assert attention_weights in ("natten", "spatial_channel"), "The attention method is not supported."
        if attention_weights == "spatial_channel":
            self.attention_conv = SpatialChannelAttention(cout, act)
        elif attention_weights is not None:
            self.attention_conv = nn.Sequential(
                _ToNHWC(),
                nn.LayerNorm(cout),
                NeighborhoodAttention2D(cout, heads, kernel, dilation, attn_drop=na_drop, proj_drop=na_drop),
                nn.LayerNorm(cout),
                _ToNCHW(),
            )
        self.res_modules = nn.ModuleList(
            [ResConvBlock2d(cin, cout, kernel_size, d, act, num_blocks, batchnorm_first) for d in dilations]
        )

    def forward(self, x):
        out = self.skip(x)
        skip = out
        for layer in self.res_modules:
            out = out + layer(x)
        if self.attention_weights == "natten":
            out = out + self.attention_conv(skip)
        elif self.attention_weights is not None:
            out = out * self.attention_conv(skip)
        return out


class PoolResidualConv(nn.Module):
    """nn/modules/convolution.py:398-513 (RESA path)."""

    def __init__(self, cin, cout, dropout=0.0, kernel_size=3, num_blocks=2, attention_weights=None, act="SiLU",
                 dilations=None, pool_first=True, pool_by_max=False, batchnorm_first=False, res_block_type="resa", **na):
        super().__init__()
        self.pool_first = pool_first
        self.pool_by_max = pool_by_max
        if pool_first and not pool_by_max:
            if batchnorm_first:
                self.pool_conv = nn.Conv2d(cin, cout, 3, padding=1, stride=2)
            else:
                self.pool_conv = ConvBlock2d(cin, cout, 3, padding=1, stride=2, add_activation=False)
            cin = cout
        if res_block_type == "res":
            self.res_conv = ResidualConv(cin, cout, kernel_size, num_blocks, attention_weights, act, batchnorm_first)
        else:
            self.res_conv = ResidualAConv(cin, cout, kernel_size, num_blocks, dilations, attention_weights, act,
                                          batchnorm_first=batchnorm_first, **na)
        self.dropout_layer = nn.Dropout2d(p=dropout)

    def forward(self, x):
        h, w = x.shape[-2:]
        if self.pool_first:
            if self.pool_by_max:
                x = F.adaptive_max_pool2d(x, output_size=(h // 2, w // 2))
            else:
                x = self.pool_conv(x)
        return self.dropout_layer(self.res_conv(x))


class UpConv(nn.Module):
    """nn/modules/convolution.py:45-68 (ConvTranspose2d wrapper + check_upsample)."""

    def __init__(self, cin, cout, kernel_size=3, stride=2, padding=1):
        super().__init__()
        self.up_conv = nn.ConvTranspose2d(cin, cout, kernel_size, stride=stride, padding=padding)

    def forward(self, x, size):
        return resize_to(self.up_conv(x), size)


class TowerUNetEncoder(nn.Module):
    """nn/modules/unet_parts.py:377-449."""

    def __init__(self, channels, dilations, act, dropout, attention_weights, pool_by_max, batchnorm_first=False,
                 res_block_type="resa"):
        super().__init__()
        kw = dict(dropout=dropout, act=act, pool_by_max=pool_by_max, attention_weights=attention_weights,
                  batchnorm_first=batchnorm_first, res_block_type=res_block_type)
        na = (lambda k: NATTEN_PARAMS[k]) if attention_weights else (lambda k: {})
        self.down_a = PoolResidualConv(channels[0], channels[0], dilations=dilations, pool_first=False, **kw, **na("a"))
        self.down_b = PoolResidualConv(channels[0], channels[1], dilations=dilations[:3], **kw, **na("b"))
        self.down_c = PoolResidualConv(channels[1], channels[2], dilations=dilations[:2], **kw, **na("c"))
        kw["attention_weights"] = None
        self.down_d = PoolResidualConv(channels[2], channels[3], kernel_size=1, num_blocks=1, dilations=[1], **kw)

    def forward(self, x):
        x_a = self.down_a(x)
        x_b = self.down_b(x_a)
        x_c = self.down_c(x_b)
        x_d = self.down_d(x_c)
        return {"x_a": x_a, "x_b": x_b, "x_c": x_c, "x_d": x_d}


class UNetUpBlock(nn.Module):
    """nn/modules/unet_parts.py:312-374 (num_blocks is NOT forwarded: :355-368)."""

    def __init__(self, cin, cout, kernel_size=3, attention_weights=None, act="SiLU", dilations=None,
                 resample_up=True, num_blocks=2, batchnorm_first=False, res_block_type="resa", **na):
        super().__init__()
        if resample_up:
            self.up_conv = UpConv(cin, cin)
        if res_block_type == "res":  # unet_parts.py:343-353: ResidualConv DOES receive num_blocks
            self.res_conv = ResidualConv(cin, cout, kernel_size, num_blocks, attention_weights, act, batchnorm_first)
        else:
            self.res_conv = ResidualAConv(cin, cout, kernel_size, 2, dilations, attention_weights, act,
                                          batchnorm_first=batchnorm_first, **na)

    def forward(self, x, size):
        if tuple(x.shape[-2:]) != tuple(size):
            x = self.up_conv(x, size=size)
        return self.res_conv(x)


class TowerUNetDecoder(nn.Module):
    """nn/modules/unet_parts.py:452-525."""

    def __init__(self, channels, up_channels, dilations, act, attention_weights, dropout=0.0, batchnorm_first=False,
                 res_block_type="resa"):
        super().__init__()
        na = (lambda k: dict(NATTEN_PARAMS[k], na_drop=dropout)) if attention_weights else (lambda k: {})
        kw = dict(act=act, batchnorm_first=batchnorm_first, res_block_type=res_block_type)
        self.over_d = UNetUpBlock(channels[3], up_channels, kernel_size=1, dilations=[1], resample_up=False, num_blocks=1, **kw)
        self.up_cu = UNetUpBlock(up_channels, up_channels, dilations=dilations[:2], attention_weights=attention_weights, **kw, **na("c"))
        self.up_bu = UNetUpBlock(up_channels, up_channels, dilations=dilations[:3], attention_weights=attention_weights, **kw, **na("b"))
        self.up_au = UNetUpBlock(up_channels, up_channels, dilations=dilations, attention_weights=attention_weights, **kw, **na("a"))

    def forward(self, x):
        x_du = self.over_d(x["x_d"], size=x["x_d"].shape[-2:])
        x_cu = self.up_cu(x_du, size=x["x_c"].shape[-2:])
        x_bu = self.up_bu(x_cu, size=x["x_b"].shape[-2:])
        x_au = self.up_au(x_bu, size=x["x_a"].shape[-2:])
        return {"x_au": x_au, "x_bu": x_bu, "x_cu": x_cu, "x_du": x_du}


class TowerUNetBlock(nn.Module):
    """nn/modules/unet_parts.py:615-760 (use_latlon=False)."""

    def __init__(self, side_channels, down_channels, up_channels, out_channels, tower=False, dilations=None,
                 attention_weights=None, act="SiLU", batchnorm_first=False, res_block_type="resa", **na):
        super().__init__()
        cin = side_channels + down_channels + up_channels * 2
        self.backbone_down_conv = UpConv(down_channels, down_channels)
        self.decode_down_conv = UpConv(up_channels, up_channels)
        if tower:
            self.tower_conv = UpConv(up_channels, up_channels)
            cin += up_channels
        if res_block_type == "res":
            self.res_conv = ResidualConv(cin, out_channels, 3, 2, attention_weights, act, batchnorm_first)
        else:
            self.res_conv = ResidualAConv(cin, out_channels, 3, 2, dilations, attention_weights, act,
                                          batchnorm_first=batchnorm_first, **na)

    def forward(self, backbone_side, backbone_down, decode_side, decode_down, tower_down=None):
        size = decode_side.shape[-2:]
        x = torch.cat(
            (backbone_side, self.backbone_down_conv(backbone_down, size=size), decode_side,
             self.decode_down_conv(decode_down, size=size)), dim=1)
        if tower_down is not None:
            x = torch.cat((x, self.tower_conv(tower_down, size=size)), dim=1)
        return self.res_conv(x)


class TowerUNetFusion(nn.Module):
    """nn/modules/unet_parts.py:528-612."""

    def __init__(self, channels, up_channels, dilations, act, attention_weights, batchnorm_first=False,
                 res_block_type="resa"):
        super().__init__()
        na = (lambda k: NATTEN_PARAMS[k]) if attention_weights else (lambda k: {})
        kw = dict(up_channels=up_channels, out_channels=up_channels, act=act, attention_weights=attention_weights,
                  batchnorm_first=batchnorm_first, res_block_type=res_block_type)
        self.tower_c = TowerUNetBlock(channels[2], channels[3], dilations=dilations[:2], **kw, **na("c"))
        self.tower_b = TowerUNetBlock(channels[1], channels[2], tower=True, dilations=dilations, **kw, **na("b"))
        self.tower_a = TowerUNetBlock(channels[0], channels[1], tower=True, dilations=dilations, **kw, **na("a"))

    def forward(self, encoded, decoded):
        c = self.tower_c(encoded["x_c"], encoded["x_d"], decoded["x_cu"], decoded["x_du"])
        b = self.tower_b(encoded["x_b"], encoded["x_c"], decoded["x_bu"], decoded["x_cu"], tower_down=c)
        a = self.tower_a(encoded["x_a"], encoded["x_b"], decoded["x_au"], decoded["x_bu"], tower_down=b)
        return {"x_tower_a": a, "x_tower_b": b, "x_tower_c": c}


class StreamConv2d(nn.Module):
    """nn/modules/unet_parts.py:196-224."""

    def __init__(self, cin, hidden, cout, act):
        super().__init__()
        self.conv = nn.Sequential(ConvBlock2d(cin, hidden, 3, padding=1, act=act), nn.Conv2d(hidden, cout, 3, padding=1))

    def forward(self, x):
        return self.conv(x)


class TowerUNetFinal(nn.Module):
    """nn/modules/unet_parts.py:227-309."""

    def __init__(self, in_channels, num_classes, act="SiLU", resample_factor=0):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        if resample_factor > 1:
            self.up_conv = UpConv(in_channels, in_channels, 3, stride=resample_factor, padding=1)
        self.dist_conv = StreamConv2d(in_channels, 3, 1, act)
        self.edge_conv = StreamConv2d(in_channels, 3, 1, act)
        self.crop_conv = StreamConv2d(in_channels, 3, 1, act)
        self.fuse_conv = ConvBlock2d(3, 3, 3, padding=1, act=act)

    def forward(self, x, size=None):
        if size is not None:
            x = self.up_conv(x, size=size)
        h = torch.cat([self.dist_conv(x), self.edge_conv(x), self.crop_conv(x)], dim=1)
        h = self.fuse_conv(h)
        return torch.chunk(h, 3, dim=1)  # distance, edge, crop


class SigmoidCrisp(nn.Module):
    """nn/modules/unet_parts.py:43-98."""

    def __init__(self, smooth: float = 1e-2):
        super().__init__()
        self.smooth = smooth
        self.gamma = nn.Parameter(torch.ones(1))

    def forward(self, x):
        return torch.sigmoid(x * torch.reciprocal(self.smooth + torch.sigmoid(self.gamma)))


class TowerUNetFinalCombine(nn.Module):
    """nn/modules/unet_parts.py:101-193."""

    def __init__(self, num_classes=1, edge_activation=True, mask_activation=True):
        super().__init__()
        self.final_dist = nn.Sequential(nn.Conv2d(1, 1, 1), nn.Sigmoid())
        self.dist_gamma1 = nn.Parameter(torch.ones(1))
        self.dist_gamma2 = nn.Parameter(torch.ones(1))
        self.dist_gamma3 = nn.Parameter(torch.ones(1))
        self.final_edge = nn.Sequential(nn.Conv2d(1, 1, 1), SigmoidCrisp() if edge_activation else nn.Identity())
        self.edge_gamma1 = nn.Parameter(torch.ones(1))
        self.edge_gamma2 = nn.Parameter(torch.ones(1))
        self.edge_gamma3 = nn.Parameter(torch.ones(1))
        self.final_crop = nn.Sequential(
            nn.Conv2d(num_classes, num_classes, 1), nn.Sigmoid() if mask_activation else nn.Identity())
        self.crop_gamma1 = nn.Parameter(torch.ones(1))
        self.crop_gamma2 = nn.Parameter(torch.ones(1))
        self.crop_gamma3 = nn.Parameter(torch.ones(1))

    def forward(self, out_a, out_b, out_c):
        def mix(i, g1, g2, g3):
            return torch.reciprocal(g1) * out_a[i] + torch.reciprocal(g2) * out_b[i] + torch.reciprocal(g3) * out_c[i]

        return {
            "distance": self.final_dist(mix(0, self.dist_gamma1, self.dist_gamma2, self.dist_gamma3)),
            "edge": self.final_edge(mix(1, self.edge_gamma1, self.edge_gamma2, self.edge_gamma3)),
            "crop": self.final_crop(mix(2, self.crop_gamma1, self.crop_gamma2, self.crop_gamma3)),
        }


def init_conv_weights(module: nn.Module) -> None:
    """layers/weights.py:24-39: Kaiming-normal weights, N(0,1) biases, BN gamma~N(1,.02)."""
    if isinstance(module, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear)):
        nn.init.kaiming_normal_(module.weight.data, a=0, mode="fan_in")
        if module.bias is not None:
            nn.init.normal_(module.bias.data)
    elif isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
        nn.init.normal_(module.weight.data, 1.0, 0.02)
        nn.init.constant_(module.bias.data, 0.0)


class TowerUNet(nn.Module):
    """models/nunet.py:108-265."""

    def __init__(self, in_channels, in_time, hidden_channels=64, num_classes=1, dilations=None,
                 activation_type="SiLU", dropout=0.0, attention_weights="natten", pool_by_max=False,
                 edge_activation=True, mask_activation=True, batchnorm_first=False, res_block_type="resa"):
        super().__init__()
        if dilations is None:
            dilations = [1, 2]
        channels = [hidden_channels, hidden_channels * 2, hidden_channels * 4, hidden_channels * 8]
        up_channels = int(hidden_channels * len(channels))
        self.pre_unet = PreTimeReduction(in_channels, in_time, channels[0], activation_type)
        self.encoder = TowerUNetEncoder(channels, dilations, activation_type, dropout, None, pool_by_max,
                                        batchnorm_first, res_block_type)
        self.decoder = TowerUNetDecoder(channels, up_channels, dilations, activation_type, attention_weights, dropout,
                                        batchnorm_first, res_block_type)
        self.tower_fusion = TowerUNetFusion(channels, up_channels, dilations, activation_type, None, batchnorm_first,
                                            res_block_type)
        self.final_a = TowerUNetFinal(up_channels, num_classes, activation_type)
        self.final_b = TowerUNetFinal(up_channels, num_classes, activation_type, resample_factor=2)
        self.final_c = TowerUNetFinal(up_channels, num_classes, activation_type, resample_factor=4)
        self.final_combine = TowerUNetFinalCombine(num_classes, edge_activation, mask_activation)
        self.apply(init_conv_weights)

    def forward(self, x, latlon_coords=None, return_stages: bool = False):
        emb = self.pre_unet(x)
        enc = self.encoder(emb)
        dec = self.decoder(enc)
        tow = self.tower_fusion(enc, dec)
        size = tow["x_tower_a"].shape[-2:]
        out_a = self.final_a(tow["x_tower_a"])
        out_b = self.final_b(tow["x_tower_b"], size=size)
        out_c = self.final_c(tow["x_tower_c"], size=size)
        out = self.final_combine(out_a, out_b, out_c)
        if return_stages:
            stages = {"embeddings": emb, **enc, **dec, **tow}
            for sfx, o in (("a", out_a), ("b", out_b), ("c", out_c)):
                for name, t in zip(("distance", "edge", "crop"), o):
                    stages[f"{name}_{sfx}"] = t
            return out, stages
        return out


# --------------------------------------------------------------------------
# Losses (losses/losses.py) and the LitModel loss glue (models/lightning.py)
# --------------------------------------------------------------------------

def loss_preprocess(inputs, targets, mask=None, one_hot_targets=True):
    """losses/losses.py:9-59 (transform_logits=False)."""
    if one_hot_targets and inputs.shape[1] > 1:
        targets = F.one_hot(targets, num_classes=inputs.shape[1]).permute(0, 3, 1, 2)
    elif targets.dim() == 3:
        targets = targets.unsqueeze(1)
    if mask is not None:
        if mask.dim() == 3:
            mask = mask.unsqueeze(1)
        inputs = inputs * mask
        targets = targets * mask
    return inputs, targets


def tanimoto_complement_distance(y, yhat, smooth=1e-5, depth=5):
    """losses/losses.py:152-186 (dim=(1,2,3))."""
    tpl = (y * yhat).sum(dim=(1, 2, 3))
    sq = (y**2 + yhat**2).sum(dim=(1, 2, 3))
    den = 0.0
    for d in range(depth):
        a = 2.0**d
        b = -(2.0 * a - 1.0)
        den = den + torch.reciprocal(((a * sq) + (b * tpl)) + smooth)
    return 1.0 - ((tpl + smooth) * den) * (1.0 / depth)


def tanimoto_complement_loss(inputs, targets, mask=None, one_hot_targets=True):
    """losses/losses.py:188-218."""
    inputs, targets = loss_preprocess(inputs, targets, mask, one_hot_targets)
    l1 = tanimoto_complement_distance(targets, inputs)
    l2 = tanimoto_complement_distance(1.0 - targets, 1.0 - inputs)
    return ((l1 + l2) * 0.5).mean()


def _tanimoto_dist(ypred, ytrue, smooth=1e-5):
    """losses/losses.py:221-248."""
    ytrue = ytrue.to(dtype=ypred.dtype)
    tpl = (ypred * ytrue).sum(dim=(1, 2, 3))
    sq = (ypred**2 + ytrue**2).sum(dim=(1, 2, 3))
    return 1.0 - (tpl + smooth) / ((sq - tpl) + smooth)


def tanimoto_dist_loss(inputs, targets, mask=None, one_hot_targets=True):
    """losses/losses.py:300-340."""
    inputs, targets = loss_preprocess(inputs, targets, mask, one_hot_targets)
    return ((_tanimoto_dist(inputs, targets) + _tanimoto_dist(1.0 - inputs, 1.0 - targets)) * 0.5).mean()


def tanimoto_combined_loss(inputs, targets, mask=None, one_hot_targets=True):
    """losses/losses.py:62-100 with LOSS_DICT[TanimotoCombined] (lightning.py:62-80)."""
    return 0.5 * (tanimoto_dist_loss(inputs, targets, mask, one_hot_targets)
                  + tanimoto_complement_loss(inputs, targets, mask, one_hot_targets))


LOSSES = {
    "TanimotoComplementLoss": tanimoto_complement_loss,
    "TanimotoDistLoss": tanimoto_dist_loss,
    "TanimotoCombined": tanimoto_combined_loss,
}


def true_labels(y: torch.Tensor, edge_class: int = 2):
    """models/lightning.py:161-207 (the three tensors calc_loss consumes)."""
    true_edge = (y == edge_class).long()
    true_crop = ((y > 0) & (y < edge_class)).long()
    mask = None
    if y.min() == -1:
        mask = (y != -1).long().unsqueeze(1)
    return true_edge, true_crop, mask


def calc_loss(pred: T.Dict[str, torch.Tensor], y, bdist, loss_name="TanimotoComplementLoss", edge_class=2):
    """models/lightning.py:209-354 with classes_l2/l3 = None."""
    fn = LOSSES[loss_name]
    true_edge, true_crop, mask = true_labels(y, edge_class)
    dloss = fn(pred["distance"], bdist, mask, one_hot_targets=False)
    eloss = fn(pred["edge"], true_edge, mask)
    closs = fn(pred["crop"], true_crop, mask)
    return (dloss + eloss + closs) / 3.0, {"dloss": dloss, "eloss": eloss, "closs": closs}


# --------------------------------------------------------------------------
# Key-seeded weights + seeded inputs
# --------------------------------------------------------------------------

def seeded_state_dict(template: T.Dict[str, torch.Tensor], salt: int = 0) -> T.Dict[str, torch.Tensor]:
    """Fill every tensor of a state dict from a generator seeded by crc32(key).

    Keys are normalised first: a leading ``cultionet_TowerUNet.mask_model.`` and
    torch.compile's ``_orig_mod.`` are dropped, so the reference LitModel, the
    reference TowerUNet, this oracle and the HIP build all draw identical values.
    Distributions: conv/linear weights N(0, sqrt(2/fan_in)) (the reference's
    Kaiming-normal), biases N(0, 0.5), norm weights N(1, 0.1), norm biases
    N(0, 0.1), running_mean N(0, 0.1), running_var U(0.5, 1.5), gammas U(0.8, 1.25).
    """
    out = {}
    for key, t in template.items():
        k = key.replace("cultionet_TowerUNet.mask_model.", "").replace("_orig_mod.", "")
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + salt) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            v = torch.zeros(shape, dtype=t.dtype)
        elif k.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif "gamma" in k:
            v = torch.rand(shape, generator=g) * 0.45 + 0.8
        elif t.dim() >= 2:
            fan_in = math.prod(shape[1:])
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif k.endswith("weight"):
            v = 1.0 + torch.randn(shape, generator=g) * 0.1
        else:
            v = torch.randn(shape, generator=g) * (0.5 if _is_conv_bias(k) else 0.1)
        out[key] = v.to(t.dtype)
    return out


def _is_conv_bias(k: str) -> bool:
    # norm-layer biases sit right after a weight of the same module with dim 1;
    # conv / linear biases are the ones whose module is skip/up_conv/qkv/proj/conv.1/final_*.0
    return any(s in k for s in ("skip.", "up_conv.", "qkv.", "proj.", "conv.1.bias", "final_dist.0", "final_edge.0", "final_crop.0"))


def seeded_batch(batch: int, channels: int = 3, time: int = 12, height: int = 100, width: int = 100,
                 seed: int = 7, with_mask: bool = False):
    """Synthetic inputs as tests/conftest.py:19-55 of the reference, but seeded."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, channels, time, height, width, generator=g)
    bdist = torch.rand(batch, height, width, generator=g)
    y = torch.randint(-1 if with_mask else 0, 3, (batch, height, width), generator=g)
    return x, y, bdist


# ---------------------------------------------------------------------------
# gradient probes (fixtures): element-level evidence without shipping 42 MB of gradients
# ---------------------------------------------------------------------------
GRAD_PROBE_FIRST = 64


def grad_probe(name: str, grad: torch.Tensor) -> T.Tuple["torch.Tensor", float]:
    """(first GRAD_PROBE_FIRST elements zero-padded, fixed random projection) of one parameter's gradient.

    The projection is ``sum(g * r) / sqrt(numel)`` with ``r ~ N(0, 1)`` drawn from a CPU generator seeded by
    ``crc32(name)``: any permutation / swap / sign error inside the tensor moves it by O(rms(g)), while its value for
    the right gradient is reproducible on every box. Generator (oracle/make_golden.py, real reference) and tests use
    THIS function, so both sides project onto the same vector."""
    g = grad.detach().double().flatten().cpu()
    first = torch.zeros(GRAD_PROBE_FIRST, dtype=torch.float64)
    k = min(GRAD_PROBE_FIRST, g.numel())
    first[:k] = g[:k]
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    r = torch.randn(g.numel(), generator=gen, dtype=torch.float64)
    return first, float((g * r).sum() / math.sqrt(g.numel()))

