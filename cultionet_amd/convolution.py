"""Convolution blocks of the TowerUNet hot path, executed by HIP kernels through cultionet_amd.engine.

Host-side mirror of /root/reference/src/cultionet/nn/modules/convolution.py: same class names, constructor
arguments, sub-module attribute names (=> identical state-dict keys) and error behaviour. torch.nn
layers are used as *parameter containers only* (so initialisation and checkpoints are those of the
reference); every forward runs engine ops on ``Var`` buffers:

  * Conv2d -> BatchNorm2d -> SiLU is two engine ops (implicit-GEMM conv, fused BN+SiLU(+residual));
  * the ResUNet-a sum ``skip(x) + sum_d branch_d(x)`` is fused into the last BN+SiLU of each branch;
  * ``+ LayerNorm(NA(LayerNorm(skip)))`` keeps tensors NCHW: the two LayerNorms normalise over the
    channel stride, qkv / proj run as 1x1 convs and the residual add is fused into the second LayerNorm.
"""
from __future__ import annotations

import os
import typing as T

import torch
import torch.nn as nn

from . import engine as E
from .enums import AttentionTypes, ResBlockTypes


class SetActivation(nn.Module):
    """nn/modules/activations.py:5-24. Only SiLU (default everywhere upstream) has a fused kernel."""

    def __init__(self, activation_type: str):
        super().__init__()
        if activation_type != "SiLU":
            raise NotImplementedError(f"activation {activation_type!r}: only 'SiLU' has a HIP kernel")
        self.activation = nn.SiLU()


class _Marker(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices equal to the reference's."""


class ConvTranspose2d(nn.Module):
    """convolution.py:45-68: nn.ConvTranspose2d(k, stride, padding) then bilinear resize iff size differs."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 2, padding: int = 1):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.up_conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)

    def forward(self, x: E.Var, size, out: T.Optional[torch.Tensor] = None) -> E.Var:
        y = E.conv_transpose2d(x, self.up_conv, self.stride, self.padding, size=tuple(size), out=out)
        if y.valid is None and tuple(y.shape[-2:]) == tuple(size) and (out is None or y.t.data_ptr() == out.data_ptr()):
            return y  # already at ``size`` (the stride >= kernel path resizes in its own pointwise pass), or no resize needed
        natural = y.valid if y.valid is not None else tuple(y.shape[-2:])
        if out is not None and tuple(natural) == tuple(size):
            out = None  # no resize to write through: the caller copies
        return E.resize_bilinear(y, tuple(size), out=out)


class ConvBlock2d(nn.Module):
    """convolution.py:71-120: Conv2d(bias=False) -> BatchNorm2d -> [SiLU], or with ``batchnorm_first``
    BatchNorm2d(in) -> SiLU -> Conv2d(bias=True)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, padding: int = 0, dilation: int = 1,
                 stride: int = 1, add_activation: bool = True, activation_type: str = "SiLU",
                 batchnorm_first: bool = False):
        super().__init__()
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.batchnorm_first = batchnorm_first
        self.act = E.ACT_SILU if add_activation else E.ACT_NONE
        if batchnorm_first:
            layers = [
                nn.BatchNorm2d(in_channels),
                SetActivation(activation_type),
                nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, stride=stride),
            ]
        else:
            layers = [
                nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, stride=stride,
                          bias=False),
                nn.BatchNorm2d(out_channels),
            ]
            if add_activation:
                layers.append(SetActivation(activation_type))
        self.seq = nn.Sequential(*layers)

    def forward(self, x: E.Var, residual: T.Optional[E.Var] = None) -> E.Var:
        if not self.batchnorm_first and E.can_fuse_eval(x, self.seq[1], self.training):
            # inference on the mixed-precision path: conv + BatchNorm(running statistics) + SiLU (+ residual), one launch
            return E.conv_bn_act_eval(x, self.seq[0], self.seq[1], self.act, self.stride, self.padding, self.dilation,
                                      residual)
        if self.batchnorm_first:
            h = E.bn_act(x, self.seq[0], E.ACT_SILU, training=self.training)
            y = E.conv2d(h, self.seq[2], self.stride, self.padding, self.dilation)
            return E.add(residual, y) if residual is not None else y
        # mixed precision: the conv epilogue hands BatchNorm its batch statistics (no statistics pass over y)
        y = E.conv2d(x, self.seq[0], self.stride, self.padding, self.dilation, want_stats=self.training,
                     bn=self.seq[1] if self.training else None)
        return E.bn_act(y, self.seq[1], self.act, residual=residual, training=self.training)


class ResConvBlock2d(nn.Module):
    """convolution.py:123-176. Block 0: dilation 1; blocks 1..n-1: pad = dil = max(1, d-1)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, dilation: int = 1,
                 activation_type: str = "SiLU", num_blocks: int = 2, batchnorm_first: bool = False):
        super().__init__()
        assert num_blocks > 0, "There must be at least one block."
        layers = [ConvBlock2d(in_channels, out_channels, kernel_size, padding=0 if kernel_size == 1 else kernel_size // 2,
                              dilation=1, activation_type=activation_type, batchnorm_first=batchnorm_first)]
        for _ in range(num_blocks - 1):
            d = 1 if kernel_size == 1 else max(1, dilation - 1)
            layers.append(ConvBlock2d(out_channels, out_channels, kernel_size, padding=0 if kernel_size == 1 else d,
                                      dilation=d, activation_type=activation_type, batchnorm_first=batchnorm_first))
        self.block = nn.ModuleList(layers)

    def forward(self, x: E.Var, residual: T.Optional[E.Var] = None) -> E.Var:
        last = len(self.block) - 1
        for i, layer in enumerate(self.block):
            x = layer(x, residual if i == last else None)
        return x


class ChannelAttention(nn.Module):
    """nn/modules/attention.py:12-62 (parameter container; arithmetic in cn_sca_mlp_*)."""

    def __init__(self, in_channels: int, activation_type: str):
        super().__init__()
        mk = lambda: nn.Sequential(
            nn.Conv2d(in_channels, in_channels // 2, kernel_size=1, padding=0, bias=False),
            SetActivation(activation_type),
            nn.Conv2d(in_channels // 2, in_channels, kernel_size=1, padding=0, bias=False),
        )
        self.fc1 = mk()
        self.fc2 = mk()


class SpatialAttention(nn.Module):
    """nn/modules/attention.py:65-86."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size=3, padding=1, bias=False)


class SpatialChannelAttention(nn.Module):
    """nn/modules/attention.py:89-126: attention = 1 + gamma * 0.5 * (channel + spatial)."""

    def __init__(self, in_channels: int, activation_type: str):
        super().__init__()
        self.channel_attention = ChannelAttention(in_channels=in_channels, activation_type=activation_type)
        self.spatial_attention = SpatialAttention()
        self.gamma = nn.Parameter(torch.zeros(1, requires_grad=True))


class NeighborhoodAttention2D(nn.Module):
    """natten.NeighborhoodAttention2D(dim, heads, k, dilation, qkv_bias=True, rel_pos_bias=False) parameters
    (``qkv``, ``proj`` Linear layers, names as in natten 0.17.1) driving the HIP NA kernel on NCHW buffers."""

    def __init__(self, dim: int, num_heads: int, kernel_size: int, dilation: int = 1, rel_pos_bias: bool = False,
                 qkv_bias: bool = True, attn_drop: float = 0.0, proj_drop: float = 0.0):
        super().__init__()
        if rel_pos_bias:
            raise NotImplementedError("rel_pos_bias is not used by cultionet")
        self.attn_drop, self.proj_drop = float(attn_drop), float(proj_drop)
        self.num_heads, self.kernel_size, self.dilation = num_heads, kernel_size, dilation
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: E.Var) -> E.Var:
        qkv = E.conv2d(x, _as_conv(self.qkv))
        o = E.na2d(qkv, self.num_heads, self.kernel_size, self.dilation,
                   attn_drop=self.attn_drop if self.training else 0.0)
        o = E.conv2d(o, _as_conv(self.proj))
        return E.dropout(o, self.proj_drop, channelwise=False, training=self.training)


class _LinearAsConv:
    """View of an nn.Linear as a 1x1 conv for engine.conv2d (weight [out][in] -> [out][in][1][1])."""

    __slots__ = ("lin", "__dict__")

    def __init__(self, lin: nn.Linear):
        self.lin = lin

    @property
    def weight(self):
        return self.lin.weight

    @property
    def bias(self):
        return self.lin.bias


def _as_conv(lin: nn.Linear):
    v = lin.__dict__.get("_cn_conv_view")
    if v is None:
        v = _LinearAsConv(lin)
        lin.__dict__["_cn_conv_view"] = v
    return v


class ResidualConv(nn.Module):
    """convolution.py:179-247: skip(x) + ResConvBlock2d(x) (res_block_type='res'; attention None)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, num_blocks: int = 2,
                 attention_weights: T.Optional[str] = None, activation_type: str = "SiLU",
                 batchnorm_first: bool = False):
        super().__init__()
        self.attention_weights = attention_weights
        if self.attention_weights is not None:
            assert self.attention_weights in [AttentionTypes.SPATIAL_CHANNEL], "The attention method is not supported."
            # upstream constructs SpatialChannelAttention(out_channels=...) here, a keyword its ctor does not have
            # (convolution.py:203-205): res blocks with attention fail the same way in the reference
            raise TypeError("SpatialChannelAttention.__init__() got an unexpected keyword argument 'out_channels'")
        self.seq = ResConvBlock2d(in_channels, out_channels, kernel_size, num_blocks=num_blocks,
                                  activation_type=activation_type, batchnorm_first=batchnorm_first)
        self.skip = None
        if in_channels != out_channels:
            self.skip = nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, x: E.Var) -> E.Var:
        out = E.conv2d(x, self.skip) if self.skip is not None else x
        return self.seq(x, residual=out)


_ATT_STREAM = os.environ.get("CN_ATT_STREAM", "1") != "0"  # A/B switch: the attention chain beside the conv branches


class ResidualAConv(nn.Module):
    """convolution.py:250-395: out = skip(x) + sum_d ResConvBlock2d_d(x) [+ LN(NA(LN(skip(x))))]."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, num_blocks: int = 2,
                 dilations: T.Optional[T.List[int]] = None, attention_weights: T.Optional[str] = None,
                 activation_type: str = "SiLU", batchnorm_first: bool = False, natten_num_heads: int = 8,
                 natten_kernel_size: int = 3, natten_dilation: int = 1, natten_attn_drop: float = 0.0,
                 natten_proj_drop: float = 0.0):
        super().__init__()
        if dilations is None:
            dilations = [1, 2]
        self.attention_weights = attention_weights
        self.skip = nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0) if in_channels != out_channels \
            else nn.Identity()
        if self.attention_weights is not None:
            assert self.attention_weights in [AttentionTypes.NATTEN, AttentionTypes.SPATIAL_CHANNEL], \
                "The attention method is not supported."
            if self.attention_weights != AttentionTypes.NATTEN:
                self.attention_conv = SpatialChannelAttention(in_channels=out_channels,
                                                              activation_type=activation_type)
        if self.attention_weights == AttentionTypes.NATTEN:
            self.attention_conv = nn.Sequential(
                _Marker(),
                nn.LayerNorm(out_channels),
                NeighborhoodAttention2D(out_channels, natten_num_heads, natten_kernel_size, natten_dilation,
                                        attn_drop=natten_attn_drop, proj_drop=natten_proj_drop),
                nn.LayerNorm(out_channels),
                _Marker(),
            )
        self.res_modules = nn.ModuleList([
            ResConvBlock2d(in_channels, out_channels, kernel_size, dilation=d, activation_type=activation_type,
                           num_blocks=num_blocks, batchnorm_first=batchnorm_first) for d in dilations
        ])

    def forward(self, x: E.Var, out: T.Optional["torch.Tensor"] = None) -> E.Var:
        """``out``: where the caller would like the result -- its channel slice of a tower's concat buffer (TowerUNet
        hands it down so that torch.cat needs no copy). Honoured by the final op of the default paths (the summed
        BatchNorm of the grouped branches, or the second LayerNorm of the attention branch); every other path returns a
        buffer of its own and the concat copies as before."""
        G = len(self.res_modules)
        blocks0 = [m.block[0] for m in self.res_modules]
        sum_out = out if self.attention_weights is None else None  # (with attention the branch sum is an intermediate)
        natten = self.attention_weights == AttentionTypes.NATTEN
        att_br = att = None
        if natten and _ATT_STREAM and not isinstance(self.skip, nn.Conv2d) and E.is16(x.t):
            # LayerNorm -> neighbourhood attention (qkv, windows, projection) reads nothing but x: on an auxiliary stream
            # beside the MFMA-bound convolution branches (engine.spawn); joined before the closing LayerNorm + add.
            # Mixed precision only: there the chain is bandwidth-bound (same box: bf16 step +1.0 %, reference-default
            # point +1.0 %, predict +3 %); in fp32 its 1x1 convolutions are matrix-pipe-bound like the branches (-0.5 %)
            att_br, att = E.spawn(lambda v: self.attention_conv[2](E.layer_norm_c(v, self.attention_conv[1])), [x], 0)
        fused_eval = (not blocks0[0].batchnorm_first
                      and all(E.can_fuse_eval(x, b.seq[1], self.training) for m in self.res_modules for b in m.block))
        if 2 <= G <= 4 and all(len(m.block) == 2 for m in self.res_modules) and not blocks0[0].batchnorm_first \
                and not fused_eval:
            # the G dilation branches run level by level: one grouped launch for their first convs (shared input),
            # one for their second convs; out + SiLU(BN(.)) is fused into the last BN of each branch
            blocks1 = [m.block[1] for m in self.res_modules]
            bn0 = [b.seq[1] for b in blocks0] if self.training else None
            bn1 = [b.seq[1] for b in blocks1] if self.training else None
            ys = E.conv2d_group([x] * G, [b.seq[0] for b in blocks0], [b.padding for b in blocks0],
                                [b.dilation for b in blocks0], blocks0[0].stride, bns=bn0)
            hs = E.bn_act_group(ys, [b.seq[1] for b in blocks0], blocks0[0].act, training=self.training)
            ys = E.conv2d_group(hs, [b.seq[0] for b in blocks1], [b.padding for b in blocks1],
                                [b.dilation for b in blocks1], blocks1[0].stride, bns=bn1)
            # skip(x) is recorded AFTER the branches, so in backward its bwd-data is the FIRST writer of dx (plain
            # stores) and the branches' shared-dx launch accumulates onto it: no zero-fill of dx, and no
            # read-modify-write epilogue in the 1x1 GEMM (that ordering cost 165 us at 8 x 480 x 100^2)
            skip = E.conv2d(x, self.skip) if isinstance(self.skip, nn.Conv2d) else x
            res = E.bn_act_group(ys, [b.seq[1] for b in blocks1], blocks1[0].act, residual=skip, sum_outputs=True,
                                 training=self.training, outs=[sum_out] if sum_out is not None else None)
        else:
            skip = E.conv2d(x, self.skip) if isinstance(self.skip, nn.Conv2d) else x
            res = skip
            for layer in self.res_modules:
                res = layer(x, residual=res)  # out + SiLU(BN(conv(...))) fused in the last block
        if natten:
            if att is None:
                att = self.attention_conv[2](E.layer_norm_c(skip, self.attention_conv[1]))
            E.join([att_br])
            res = E.layer_norm_c(att, self.attention_conv[3], residual=res, out=out)
        elif self.attention_weights is not None:  # spatial_channel: out *= attention(skip)
            res = E.spatial_channel_attention(skip, res, self.attention_conv)
        return res


class PoolResidualConv(nn.Module):
    """convolution.py:398-513: [stride-2 conv | adaptive max pool] -> Residual(A)Conv -> Dropout2d."""

    def __init__(self, in_channels: int, out_channels: int, dropout: float = 0.0, kernel_size: int = 3,
                 num_blocks: int = 2, attention_weights: T.Optional[str] = None, activation_type: str = "SiLU",
                 res_block_type: str = ResBlockTypes.RESA, dilations: T.Sequence[int] = None, pool_first: bool = True,
                 pool_by_max: bool = False, batchnorm_first: bool = False, natten_num_heads: int = 8,
                 natten_kernel_size: int = 3, natten_dilation: int = 1, natten_attn_drop: float = 0.0,
                 natten_proj_drop: float = 0.0):
        super().__init__()
        assert res_block_type in (ResBlockTypes.RES, ResBlockTypes.RESA)
        self.pool_first, self.pool_by_max = pool_first, pool_by_max
        self.batchnorm_first = batchnorm_first
        if self.pool_first and not self.pool_by_max:
            if batchnorm_first:
                self.pool_conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, stride=2)
            else:
                self.pool_conv = ConvBlock2d(in_channels, out_channels, kernel_size=3, padding=1, stride=2,
                                             add_activation=False, batchnorm_first=False)
            in_channels = out_channels
        if res_block_type == ResBlockTypes.RES:
            self.res_conv = ResidualConv(in_channels, out_channels, kernel_size=kernel_size,
                                         attention_weights=attention_weights, num_blocks=num_blocks,
                                         activation_type=activation_type, batchnorm_first=batchnorm_first)
        else:
            self.res_conv = ResidualAConv(in_channels, out_channels, kernel_size=kernel_size, dilations=dilations,
                                          num_blocks=num_blocks, attention_weights=attention_weights,
                                          activation_type=activation_type, batchnorm_first=batchnorm_first,
                                          natten_num_heads=natten_num_heads, natten_kernel_size=natten_kernel_size,
                                          natten_dilation=natten_dilation, natten_attn_drop=natten_attn_drop,
                                          natten_proj_drop=natten_proj_drop)
        self.dropout_layer = nn.Dropout2d(p=dropout)

    def forward(self, x: E.Var, out: T.Optional["torch.Tensor"] = None) -> E.Var:
        if self.pool_first:
            if self.pool_by_max:
                h, w = x.shape[-2:]
                x = E.adaptive_max_pool2d(x, (h // 2, w // 2))
            elif self.batchnorm_first:
                x = E.conv2d(x, self.pool_conv, 2, 1, 1)
            else:
                x = self.pool_conv(x)
        drops = self.training and self.dropout_layer.p > 0.0  # (then Dropout2d is the final op and takes `out`)
        if isinstance(self.res_conv, ResidualAConv):
            x = self.res_conv(x, out=None if drops else out)
        else:
            x = self.res_conv(x)
        return E.dropout(x, self.dropout_layer.p, channelwise=True, training=self.training, out=out)
