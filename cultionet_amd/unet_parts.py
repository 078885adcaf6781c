"""TowerUNet encoder / decoder / tower fusion / heads on the HIP engine.

Host-side mirror of /root/reference/src/cultionet/nn/modules/unet_parts.py (same class names, attribute
names => state-dict keys, constructor arguments). Forwards take and return engine ``Var`` buffers.
"""
from __future__ import annotations

import typing as T

import torch
import torch.nn as nn

from . import engine as E
from .convolution import ConvBlock2d, ConvTranspose2d, PoolResidualConv, ResidualAConv, ResidualConv, _Marker
from .enums import AttentionTypes, InferenceNames, ResBlockTypes

# unet_parts.py:19-40
NATTEN_PARAMS = {
    "a": {"natten_num_heads": 4, "natten_kernel_size": 3, "natten_dilation": 2},
    "b": {"natten_num_heads": 4, "natten_kernel_size": 3, "natten_dilation": 1},
    "c": {"natten_num_heads": 8, "natten_kernel_size": 3, "natten_dilation": 1},
    "d": {"natten_num_heads": 8, "natten_kernel_size": 1, "natten_dilation": 1},
}


class SigmoidCrisp(nn.Module):
    """unet_parts.py:43-98; evaluated inside the fused final-combine kernel."""

    def __init__(self, smooth: float = 1e-2):
        super().__init__()
        self.smooth = smooth
        self.gamma = nn.Parameter(torch.ones(1, requires_grad=True))


class TowerUNetFinalCombine(nn.Module):
    """unet_parts.py:101-193: per task sum_t (1/gamma_t) out_t -> Conv2d(1,1,1) -> Sigmoid / SigmoidCrisp,
    one fused HIP kernel (cn_final_combine_*)."""

    def __init__(self, num_classes: int, edge_activation: bool = True, mask_activation: bool = True):
        super().__init__()
        if num_classes != 1 or not edge_activation or not mask_activation:
            raise NotImplementedError("the fused final-combine kernel covers num_classes=1 with both activations")
        self.num_classes, self.edge_activation, self.mask_activation = num_classes, edge_activation, mask_activation
        self.final_dist = nn.Sequential(nn.Conv2d(1, 1, kernel_size=1, padding=0), nn.Sigmoid())
        self.dist_gamma1 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.dist_gamma2 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.dist_gamma3 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.final_edge = nn.Sequential(nn.Conv2d(1, 1, kernel_size=1, padding=0), SigmoidCrisp())
        self.edge_gamma1 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.edge_gamma2 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.edge_gamma3 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.final_crop = nn.Sequential(nn.Conv2d(num_classes, num_classes, kernel_size=1, padding=0), nn.Sigmoid())
        self.crop_gamma1 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.crop_gamma2 = nn.Parameter(torch.ones(1, requires_grad=True))
        self.crop_gamma3 = nn.Parameter(torch.ones(1, requires_grad=True))

    def forward(self, out_a: E.Var, out_b: E.Var, out_c: E.Var, suffixes: T.Sequence[str] = ("_a", "_b", "_c")):
        dist, edge, crop = E.final_combine(out_a, out_b, out_c, final_combine_params(self),
                                           self.final_edge[1].smooth)
        return {InferenceNames.DISTANCE: dist, InferenceNames.EDGE: edge, InferenceNames.CROP: crop}


def final_combine_params(fc) -> T.List[nn.Parameter]:
    """The 16 scalars in the order cn_final_combine_* expects (works for the oracle's module too)."""
    return [
        fc.dist_gamma1, fc.dist_gamma2, fc.dist_gamma3,
        fc.edge_gamma1, fc.edge_gamma2, fc.edge_gamma3,
        fc.crop_gamma1, fc.crop_gamma2, fc.crop_gamma3,
        fc.final_dist[0].weight, fc.final_edge[0].weight, fc.final_crop[0].weight,
        fc.final_dist[0].bias, fc.final_edge[0].bias, fc.final_crop[0].bias,
        fc.final_edge[1].gamma,
    ]


class StreamConv2d(nn.Module):
    """unet_parts.py:196-224: ConvBlock2d(C -> 3, 3x3, BN, SiLU) -> Conv2d(3 -> 1, 3x3, bias)."""

    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, activation_type: str):
        super().__init__()
        self.conv = nn.Sequential(
            ConvBlock2d(in_channels, hidden_channels, kernel_size=3, padding=1, add_activation=True,
                        activation_type=activation_type),
            nn.Conv2d(hidden_channels, out_channels, kernel_size=3, padding=1),
        )

    def forward(self, x: E.Var, out: T.Optional[torch.Tensor] = None) -> E.Var:
        h = self.conv[0](x)
        return E.conv2d(h, self.conv[1], 1, 1, 1, out=out)


class TowerUNetFinal(nn.Module):
    """unet_parts.py:227-309. The three stream outputs are written straight into the channel slices of the
    [B,3,H,W] buffer that feeds fuse_conv (no torch.cat); the chunked outputs stay one [B,3,H,W] buffer
    (channel = task) because the fused final-combine kernel reads it that way."""

    def __init__(self, in_channels: int, num_classes: int, activation_type: str = "SiLU", resample_factor: int = 0):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        if resample_factor > 1:
            self.up_conv = ConvTranspose2d(in_channels, in_channels, kernel_size=3, stride=resample_factor, padding=1)
        self.dist_conv = StreamConv2d(in_channels, 3, 1, activation_type)
        self.edge_conv = StreamConv2d(in_channels, 3, 1, activation_type)
        self.crop_conv = StreamConv2d(in_channels, 3, 1, activation_type)
        self.fuse_conv = ConvBlock2d(3, 3, kernel_size=3, padding=1, add_activation=True,
                                     activation_type=activation_type)

    def cn_contiguous_params(self):
        """The three 128 -> 3 stream weights, kept adjacent in the engine's flat parameter store: the mixed-precision path
        runs them as one 128 -> 9 convolution on that view (engine._thin_conv3x3_bf16)."""
        return [[s.conv[0].seq[0].weight for s in (self.dist_conv, self.edge_conv, self.crop_conv)]]

    def forward(self, x: E.Var, size=None, suffix: str = "") -> E.Var:
        if size is not None:
            x = self.up_conv(x, size=size)
        streams = (self.dist_conv, self.edge_conv, self.crop_conv)
        if self.num_classes != 1:  # wider crop stream: generic kernels, stream by stream
            h = E.cat_channels([s(x) for s in streams])
            return self.fuse_conv(h)
        # The three streams as ONE pass per layer (direct thin-conv kernels, x read once):
        #   128 -> 3 (x3, shared input) -> BN+SiLU per stream on channel slices -> 3 -> 1 (x3, grouped) -> fuse 3 -> 3
        heads = [s.conv[0] for s in streams]
        h9 = E.thin_conv3x3(x, [h.seq[0] for h in heads], grouped=False)
        B, _, H, W = h9.shape
        buf = E.alloc((B, 9, H, W), torch.float32, h9.t.device)
        acts = E.bn_act_group(E.split_channels(h9, [3, 3, 3]), [h.seq[1] for h in heads], heads[0].act,
                              training=self.training, outs=[buf[:, 3 * i:3 * i + 3] for i in range(3)])
        a9 = E.join_channels(acts, buf)
        h3 = E.thin_conv3x3(a9, [s.conv[1] for s in streams], grouped=True)
        f = self.fuse_conv
        y = E.thin_conv3x3(h3, [f.seq[0]], grouped=False)
        return E.bn_act(y, f.seq[1], f.act, training=f.training)


class UNetUpBlock(nn.Module):
    """unet_parts.py:312-374 (``num_blocks`` is not forwarded to ResidualAConv upstream: :355-368)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, num_blocks: int = 2,
                 attention_weights: T.Optional[str] = None, activation_type: str = "SiLU",
                 res_block_type: str = ResBlockTypes.RESA, dilations: T.Sequence[int] = None,
                 batchnorm_first: bool = False, resample_up: bool = True, natten_num_heads: int = 8,
                 natten_kernel_size: int = 3, natten_dilation: int = 1, natten_attn_drop: float = 0.0,
                 natten_proj_drop: float = 0.0):
        super().__init__()
        assert res_block_type in (ResBlockTypes.RES, ResBlockTypes.RESA)
        if resample_up:
            self.up_conv = ConvTranspose2d(in_channels, in_channels)
        if res_block_type == ResBlockTypes.RES:
            self.res_conv = ResidualConv(in_channels, out_channels, kernel_size=kernel_size, num_blocks=num_blocks,
                                         attention_weights=attention_weights, activation_type=activation_type,
                                         batchnorm_first=batchnorm_first)
        else:
            self.res_conv = ResidualAConv(in_channels, out_channels, kernel_size=kernel_size, dilations=dilations,
                                          attention_weights=attention_weights, activation_type=activation_type,
                                          batchnorm_first=batchnorm_first, natten_num_heads=natten_num_heads,
                                          natten_kernel_size=natten_kernel_size, natten_dilation=natten_dilation,
                                          natten_attn_drop=natten_attn_drop, natten_proj_drop=natten_proj_drop)

    def forward(self, x: E.Var, size, out: T.Optional[torch.Tensor] = None) -> E.Var:
        if tuple(x.shape[-2:]) != tuple(size):
            x = self.up_conv(x, size=size)
        if out is not None and isinstance(self.res_conv, ResidualAConv):
            return self.res_conv(x, out=out)
        return self.res_conv(x)


class TowerUNetEncoder(nn.Module):
    """unet_parts.py:377-449."""

    def __init__(self, channels: T.Sequence[int], dilations: T.Sequence[int] = None, activation_type: str = "SiLU",
                 dropout: float = 0.0, res_block_type: str = ResBlockTypes.RESA,
                 attention_weights: str = AttentionTypes.NATTEN, pool_by_max: bool = False,
                 batchnorm_first: bool = False):
        super().__init__()
        kw = dict(dropout=dropout, activation_type=activation_type, res_block_type=res_block_type,
                  batchnorm_first=batchnorm_first, pool_by_max=pool_by_max, natten_attn_drop=dropout,
                  natten_proj_drop=dropout)
        self.down_a = PoolResidualConv(channels[0], channels[0], dilations=dilations, pool_first=False,
                                       attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["a"]})
        self.down_b = PoolResidualConv(channels[0], channels[1], dilations=dilations[:3],
                                       attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["b"]})
        self.down_c = PoolResidualConv(channels[1], channels[2], dilations=dilations[:2],
                                       attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["c"]})
        self.down_d = PoolResidualConv(channels[2], channels[3], kernel_size=1, num_blocks=1, dilations=[1],
                                       attention_weights=None, **kw)

    def forward(self, x: E.Var, outs: T.Optional[T.Dict[str, torch.Tensor]] = None) -> T.Dict[str, E.Var]:
        """``outs``: channel slices of the towers' concat buffers for x_a / x_b / x_c (TowerUNet.forward_vars)."""
        o = outs or {}
        x_a = self.down_a(x, out=o.get("x_a"))
        x_b = self.down_b(x_a, out=o.get("x_b"))
        x_c = self.down_c(x_b, out=o.get("x_c"))
        x_d = self.down_d(x_c)
        return {"x_a": x_a, "x_b": x_b, "x_c": x_c, "x_d": x_d}


class TowerUNetDecoder(nn.Module):
    """unet_parts.py:452-525."""

    def __init__(self, channels: T.Sequence[int], up_channels: int, dilations: T.Sequence[int] = None,
                 activation_type: str = "SiLU", dropout: float = 0.0, res_block_type: str = ResBlockTypes.RESA,
                 attention_weights: str = AttentionTypes.NATTEN, batchnorm_first: bool = False):
        super().__init__()
        kw = dict(activation_type=activation_type, res_block_type=res_block_type, batchnorm_first=batchnorm_first,
                  natten_attn_drop=dropout, natten_proj_drop=dropout)
        self.over_d = UNetUpBlock(channels[3], up_channels, kernel_size=1, num_blocks=1, dilations=[1],
                                  resample_up=False, attention_weights=None, **kw)
        self.up_cu = UNetUpBlock(up_channels, up_channels, dilations=dilations[:2],
                                 attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["c"]})
        self.up_bu = UNetUpBlock(up_channels, up_channels, dilations=dilations[:3],
                                 attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["b"]})
        self.up_au = UNetUpBlock(up_channels, up_channels, dilations=dilations,
                                 attention_weights=attention_weights, **{**kw, **NATTEN_PARAMS["a"]})

    def forward(self, x: T.Dict[str, E.Var], outs: T.Optional[T.Dict[str, torch.Tensor]] = None) -> T.Dict[str, E.Var]:
        o = outs or {}
        x_du = self.over_d(x["x_d"], size=x["x_d"].shape[-2:])
        x_cu = self.up_cu(x_du, size=x["x_c"].shape[-2:], out=o.get("x_cu"))
        x_bu = self.up_bu(x_cu, size=x["x_b"].shape[-2:], out=o.get("x_bu"))
        x_au = self.up_au(x_bu, size=x["x_a"].shape[-2:], out=o.get("x_au"))
        return {"x_au": x_au, "x_bu": x_bu, "x_cu": x_cu, "x_du": x_du}


class TowerUNetBlock(nn.Module):
    """unet_parts.py:615-760 (use_latlon=False branch)."""

    def __init__(self, backbone_side_channels: int, backbone_down_channels: int, up_channels: int, out_channels: int,
                 tower: bool = False, kernel_size: int = 3, num_blocks: int = 2,
                 attention_weights: T.Optional[str] = None, res_block_type: str = ResBlockTypes.RESA,
                 dilations: T.Sequence[int] = None, activation_type: str = "SiLU", batchnorm_first: bool = False,
                 natten_num_heads: int = 8, natten_kernel_size: int = 3, natten_dilation: int = 1,
                 natten_attn_drop: float = 0.0, natten_proj_drop: float = 0.0, use_latlon: bool = False):
        super().__init__()
        if use_latlon:
            raise NotImplementedError("use_latlon is never enabled by CultionetLitModel (cultionet.py:55)")
        self.use_latlon = use_latlon
        assert res_block_type in (ResBlockTypes.RES, ResBlockTypes.RESA)
        in_channels = backbone_side_channels + backbone_down_channels + up_channels * 2
        self.cat_channels = [backbone_side_channels, backbone_down_channels, up_channels, up_channels] \
            + ([up_channels] if tower else [])
        self.backbone_down_conv = ConvTranspose2d(backbone_down_channels, backbone_down_channels, 3, 2, 1)
        self.decode_down_conv = ConvTranspose2d(up_channels, up_channels, 3, 2, 1)
        if tower:
            self.tower_conv = ConvTranspose2d(up_channels, up_channels, 3, 2, 1)
            in_channels += up_channels
        if res_block_type == ResBlockTypes.RES:
            self.res_conv = ResidualConv(in_channels, out_channels, kernel_size=kernel_size, num_blocks=num_blocks,
                                         attention_weights=attention_weights, activation_type=activation_type,
                                         batchnorm_first=batchnorm_first)
        else:
            self.res_conv = ResidualAConv(in_channels, out_channels, kernel_size=kernel_size, num_blocks=num_blocks,
                                          dilations=dilations, attention_weights=attention_weights,
                                          activation_type=activation_type, batchnorm_first=batchnorm_first,
                                          natten_num_heads=natten_num_heads, natten_kernel_size=natten_kernel_size,
                                          natten_dilation=natten_dilation, natten_attn_drop=natten_attn_drop,
                                          natten_proj_drop=natten_proj_drop)

    def make_buffer(self, B: int, size, like: torch.Tensor) -> T.Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The concat buffer of this block for a [B, ., size] level, allocated BEFORE the encoder / decoder run, and its
        channel slices for ``backbone_side`` and ``decode_side``: their producers (the last op of down_x / up_xu) write
        there directly and torch.cat (unet_parts.py:700-720 of the reference) copies nothing."""
        cs = self.cat_channels
        buf = E.new_buffer((B, sum(cs), size[0], size[1]), like)
        return buf, buf[:, :cs[0]], buf[:, cs[0] + cs[1]:cs[0] + cs[1] + cs[2]]

    def forward(self, backbone_side: E.Var, backbone_down: E.Var, decode_side: E.Var, decode_down: E.Var,
                tower_down: T.Optional[E.Var] = None, latlon_coords=None, buf: T.Optional[torch.Tensor] = None) -> E.Var:
        size = decode_side.shape[-2:]
        # the resized up-convolutions write straight into their channel slices of the concat buffer
        cs = [backbone_side.shape[1], backbone_down.shape[1], decode_side.shape[1], decode_down.shape[1]]
        if tower_down is not None:
            cs.append(tower_down.shape[1])
        B = decode_side.shape[0]
        want = (B, sum(cs), size[0], size[1])
        if buf is None or tuple(buf.shape) != want or buf.dtype != decode_side.t.dtype:
            buf = E.new_buffer(want, decode_side.t)
        offs = [sum(cs[:i]) for i in range(len(cs))]
        sl = lambda i: buf[:, offs[i]:offs[i] + cs[i]]
        parts = [backbone_side, self.backbone_down_conv(backbone_down, size=size, out=sl(1)), decode_side,
                 self.decode_down_conv(decode_down, size=size, out=sl(3))]
        if tower_down is not None:
            parts.append(self.tower_conv(tower_down, size=size, out=sl(4)))
        return self.res_conv(E.cat_channels(parts, buf=buf))


class TowerUNetFusion(nn.Module):
    """unet_parts.py:528-612."""

    def __init__(self, channels: T.Sequence[int], up_channels: int, dilations: T.Sequence[int] = None,
                 activation_type: str = "SiLU", dropout: float = 0.0, res_block_type: str = ResBlockTypes.RESA,
                 attention_weights: str = AttentionTypes.NATTEN, batchnorm_first: bool = False,
                 use_latlon: bool = False):
        super().__init__()
        kw = dict(up_channels=up_channels, out_channels=up_channels, activation_type=activation_type,
                  res_block_type=res_block_type, batchnorm_first=batchnorm_first, attention_weights=attention_weights,
                  natten_attn_drop=dropout, natten_proj_drop=dropout, use_latlon=use_latlon)
        self.tower_c = TowerUNetBlock(channels[2], channels[3], dilations=dilations[:2], **{**kw, **NATTEN_PARAMS["c"]})
        self.tower_b = TowerUNetBlock(channels[1], channels[2], tower=True, dilations=dilations,
                                      **{**kw, **NATTEN_PARAMS["b"]})
        self.tower_a = TowerUNetBlock(channels[0], channels[1], tower=True, dilations=dilations,
                                      **{**kw, **NATTEN_PARAMS["a"]})

    def make_buffers(self, B: int, sizes: T.Dict[str, T.Tuple[int, int]], like: torch.Tensor):
        """(bufs, encoder outs, decoder outs): the three concat buffers and the slices x_a/x_b/x_c, x_au/x_bu/x_cu go to."""
        bufs, enc, dec = {}, {}, {}
        for k, blk in (("a", self.tower_a), ("b", self.tower_b), ("c", self.tower_c)):
            bufs[k], enc["x_" + k], dec["x_" + k + "u"] = blk.make_buffer(B, sizes[k], like)
        return bufs, enc, dec

    def forward(self, encoded: T.Dict[str, E.Var], decoded: T.Dict[str, E.Var], latlon_coords=None,
                bufs: T.Optional[T.Dict[str, torch.Tensor]] = None,
                after: T.Optional[T.Dict[str, T.Callable[[E.Var], None]]] = None):
        """``after["c"]`` / ``after["b"]``: called with x_tower_c / x_tower_b as soon as they exist (TowerUNet starts their
        heads there, beside the next tower's convolutions)."""
        bf, af = bufs or {}, after or {}
        c = self.tower_c(encoded["x_c"], encoded["x_d"], decoded["x_cu"], decoded["x_du"], buf=bf.get("c"))
        if "c" in af:
            af["c"](c)
        b = self.tower_b(encoded["x_b"], encoded["x_c"], decoded["x_bu"], decoded["x_cu"], tower_down=c, buf=bf.get("b"))
        if "b" in af:
            af["b"](b)
        a = self.tower_a(encoded["x_a"], encoded["x_b"], decoded["x_au"], decoded["x_bu"], tower_down=b, buf=bf.get("a"))
        return {"x_tower_a": a, "x_tower_b": b, "x_tower_c": c}
