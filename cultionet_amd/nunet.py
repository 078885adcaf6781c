"""TowerUNet on the HIP engine.

Host-side mirror of /root/reference/src/cultionet/models/nunet.py: ``TowerUNet(in_channels, in_time,
hidden_channels, ...)`` with the reference's constructor arguments, sub-module names (state-dict keys,
442 tensors at the default config) and output dictionary. ``forward(x)`` takes/returns torch tensors and
bridges to torch.autograd when gradients are enabled; ``forward_vars`` is the engine-level entry used by
the native training step.

There is no Transformer temporal encoder in this snapshot of the reference: the time axis
is collapsed by PreTimeReduction (two Conv3d stacks + BN + SiLU, summed, LayerNorm over channels).
"""
from __future__ import annotations

import os
import typing as T

import torch
import torch.nn as nn

from . import engine as E
from .convolution import SetActivation, _Marker
from .enums import AttentionTypes, InferenceNames, ResBlockTypes
from .unet_parts import (TowerUNetDecoder, TowerUNetEncoder, TowerUNetFinal, TowerUNetFinalCombine,
                         TowerUNetFusion)


def init_conv_weights(module: nn.Module) -> None:
    """layers/weights.py:24-39: Kaiming-normal(fan_in) weights, N(0,1) biases, BN gamma ~ N(1, .02), beta = 0."""
    if isinstance(module, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear)):
        nn.init.kaiming_normal_(module.weight.data, a=0, mode="fan_in")
        if module.bias is not None:
            nn.init.normal_(module.bias.data)
    elif isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
        nn.init.normal_(module.weight.data, 1.0, 0.02)
        nn.init.constant_(module.bias.data, 0.0)


class _Conv3dAs1x1:
    """View of the (T',1,1) Conv3d weight [hid, C, T', 1, 1] as a 1x1 conv weight [hid, C*T']."""

    def __init__(self, conv: nn.Conv3d):
        self.conv = conv
        self.bias = None

    @property
    def weight(self):
        w = self.conv.weight
        return w.view(w.shape[0], -1)


class Conv3d(nn.Module):
    """nunet.py:18-57: Conv3d(C->C,(k,1,1)) -> BN3d -> SiLU -> Conv3d(C->hid,(T-k+1,1,1)) -> BN2d -> SiLU.
    Both Conv3d run as 1x1 contractions over the [B, C*T, H, W] view (cn_pack_timeconv / cn_conv2d)."""

    def __init__(self, in_channels: int, in_time: int, out_channels: int, kernel_size: int, activation_type: str):
        super().__init__()
        remaining_time = in_time - kernel_size + 1
        self.in_channels, self.in_time, self.remaining_time = in_channels, in_time, remaining_time
        self.seq = nn.Sequential(
            nn.Conv3d(in_channels, in_channels, kernel_size=(kernel_size, 1, 1), padding=0, bias=False),
            nn.BatchNorm3d(in_channels),
            SetActivation(activation_type=activation_type),
            nn.Conv3d(in_channels, out_channels, kernel_size=(remaining_time, 1, 1), padding=0, bias=False),
            _Marker(),
            nn.BatchNorm2d(out_channels),
            SetActivation(activation_type=activation_type),
        )
        self._reduce = _Conv3dAs1x1(self.seq[3])

    def forward(self, x: E.Var, residual: T.Optional[E.Var] = None) -> E.Var:
        h = E.time_conv(x, self.seq[0], self.in_time)
        h = E.bn_act(h, self.seq[1], E.ACT_SILU, channels=self.in_channels, training=self.training)
        h = E.conv2d(h, self._reduce)
        return E.bn_act(h, self.seq[5], E.ACT_SILU, residual=residual, training=self.training)


class PreTimeReduction(nn.Module):
    """nunet.py:60-105: conv3(x) + conv5(x) -> LayerNorm over channels."""

    def __init__(self, in_channels: int, in_time: int, out_channels: int, activation_type: str):
        super().__init__()
        self.conv3 = Conv3d(in_channels, in_time, out_channels, 3, activation_type)
        self.conv5 = Conv3d(in_channels, in_time, out_channels, 5, activation_type)
        self.layer_norm = nn.Sequential(_Marker(), nn.LayerNorm(out_channels), _Marker())

    def forward(self, x: E.Var) -> E.Var:
        # the whole stage as one fused kernel family (3 launches forward, 1 in inference, 3 backward on the side stream);
        # writes bf16 NHWC directly inside the mixed-precision region
        y = E.pretime_reduction(x, self, self.conv3.in_channels, self.conv3.in_time)
        if y is not None:
            return y
        x3 = self.conv3(x)
        s = self.conv5(x, residual=x3)  # x3 + x5 fused into conv5's last BN+SiLU
        y = E.layer_norm_c(s, self.layer_norm[1])
        # mixed precision: the time reduction (0.1 % of the FLOPs, fp32 input chips) stays fp32; its output enters the
        # bf16 NHWC region here and the tower heads leave it again (engine._thin_conv3x3_bf16)
        return E.to_bf16(y) if E.bf16_enabled() else y


_CAT_IN_PLACE = os.environ.get("CN_CAT_IN_PLACE", "1") != "0"  # A/B switch: concat inputs produced in place


class TowerUNet(nn.Module):
    """nunet.py:108-265."""

    def __init__(self, in_channels: int, in_time: int, hidden_channels: int = 64, num_classes: int = 1,
                 dilations: T.Optional[T.Sequence[int]] = None, activation_type: str = "SiLU", dropout: float = 0.0,
                 res_block_type: str = ResBlockTypes.RESA, attention_weights: str = AttentionTypes.NATTEN,
                 pool_by_max: bool = False, batchnorm_first: bool = False, edge_activation: bool = True,
                 mask_activation: bool = True, use_latlon: bool = False):
        super().__init__()
        if dilations is None:
            dilations = [1, 2]
        channels = [hidden_channels, hidden_channels * 2, hidden_channels * 4, hidden_channels * 8]
        up_channels = int(hidden_channels * len(channels))
        self.in_channels, self.in_time = in_channels, in_time
        # the bf16 NHWC kernels move channels in 16-byte groups: widths that are not multiples of 8 train / predict in fp32
        self.mixed_precision_ok = hidden_channels % 8 == 0
        # the reference wraps pre_unet in torch.compile (nunet.py:141), which renames its checkpoint keys to
        # pre_unet._orig_mod.*; both spellings are accepted on load (see _load_from_state_dict).
        self.pre_unet = PreTimeReduction(in_channels, in_time, channels[0], activation_type)
        self.encoder = TowerUNetEncoder(channels=channels, dilations=dilations, activation_type=activation_type,
                                        dropout=dropout, res_block_type=res_block_type, attention_weights=None,
                                        pool_by_max=pool_by_max, batchnorm_first=batchnorm_first)
        self.decoder = TowerUNetDecoder(channels=channels, up_channels=up_channels, dilations=dilations,
                                        activation_type=activation_type, dropout=dropout,
                                        res_block_type=res_block_type, attention_weights=attention_weights,
                                        batchnorm_first=batchnorm_first)
        self.tower_fusion = TowerUNetFusion(channels=channels, up_channels=up_channels, dilations=dilations,
                                            activation_type=activation_type, dropout=dropout,
                                            res_block_type=res_block_type, attention_weights=None,
                                            batchnorm_first=batchnorm_first, use_latlon=use_latlon)
        self.final_a = TowerUNetFinal(up_channels, num_classes, activation_type)
        self.final_b = TowerUNetFinal(up_channels, num_classes, activation_type, resample_factor=2)
        self.final_c = TowerUNetFinal(up_channels, num_classes, activation_type, resample_factor=4)
        self.final_combine = TowerUNetFinalCombine(num_classes, edge_activation, mask_activation)
        self.apply(init_conv_weights)
        self.__dict__["_cn_store"] = None

    # ---- checkpoint compatibility -------------------------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith(prefix + "pre_unet._orig_mod."):
                state_dict[k.replace("pre_unet._orig_mod.", "pre_unet.")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    #: explicit precision of the drop-in forward: None = follow the ambient torch.autocast region (what
    #: lightning.Trainer(precision=...) opens); "32-true" / "bf16-mixed" pin it (see autograd_bridge._autocast_bf16)
    precision: T.Optional[str] = None

    #: inference through a recorded launch plan (cultionet_amd/replay.py): set by the sliding-window predictor; the
    #: outputs of a replayed forward are the plan's own buffers (consume them before the next forward)
    replay = False

    #: write checkpoints with the reference's key spelling ``pre_unet._orig_mod.*`` (upstream wraps pre_unet in
    #: torch.compile, nunet.py:141, so ITS strict load expects that prefix). Off by default: plain keys.
    upstream_checkpoint_keys = False

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if self.upstream_checkpoint_keys:
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            for k in [k for k in sd if k.startswith(prefix + "pre_unet.")]:
                sd[k.replace(prefix + "pre_unet.", prefix + "pre_unet._orig_mod.", 1)] = sd.pop(k)
        return sd

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        st = self.__dict__.get("_cn_store")
        if st is not None:
            st.bump()
        return out

    # ---- engine plumbing ------------------------------------------------------------------------
    def param_store(self) -> E.ParamStore:
        """The flat parameter store of this model (created on first use, after .to('cuda'))."""
        st = self.__dict__.get("_cn_store")
        if st is None or not st.owns(self):
            st = E.ParamStore(self)
            self.__dict__["_cn_store"] = st
            self.__dict__["_cn_nbt"] = [m.num_batches_tracked for m in self.modules()
                                        if isinstance(m, nn.modules.batchnorm._BatchNorm)
                                        and m.num_batches_tracked is not None]
        return st

    def has_dropout(self) -> bool:
        """Any Dropout2d / Dropout / natten attn_drop / proj_drop with p > 0 (fixed at construction: cached)."""
        hd = self.__dict__.get("_cn_has_dropout")
        if hd is None:
            hd = any(isinstance(m, (nn.Dropout, nn.Dropout2d)) and m.p > 0 for m in self.modules()) or \
                any(getattr(m, "attn_drop", 0.0) > 0 or getattr(m, "proj_drop", 0.0) > 0 for m in self.modules())
            self.__dict__["_cn_has_dropout"] = hd
        return hd

    def forward_vars(self, x: E.Var) -> T.Dict[str, E.Var]:
        """Engine-level forward: x is a Var over [B, C*T, H, W]; returns {distance, edge, crop} Vars."""
        try:
            return self._forward_vars(x)
        except BaseException:
            E.release_branches()  # a forward that died between spawn() and join() must not leave frees deferred
            raise

    def _forward_vars(self, x: E.Var) -> T.Dict[str, E.Var]:
        E.current_store().refresh()  # torch optimizers / checkpoint loads since the last pack (drop-in mode)
        if self.training and self.has_dropout():
            E.begin_rng_step(x.t.device)  # fresh dropout masks for this step (device step word; see engine.manual_seed)
        emb = self.pre_unet(x)
        # the towers' concat buffers exist before the encoder / decoder run: x_a / x_b / x_c and x_au / x_bu / x_cu are
        # produced in their channel slices (no concat copies of the six "side" inputs)
        bufs = eouts = douts = None
        if _CAT_IN_PLACE and not self.encoder.down_b.pool_by_max:
            H, W = emb.shape[-2:]
            half = lambda n: (n - 1) // 2 + 1  # 3x3, stride 2, padding 1
            sizes = {"a": (H, W), "b": (half(H), half(W)), "c": (half(half(H)), half(half(W)))}
            bufs, eouts, douts = self.tower_fusion.make_buffers(emb.shape[0], sizes, emb.t)
        enc = self.encoder(emb, outs=eouts)
        dec = self.decoder(enc, outs=douts)
        # final_c / final_b start the moment their tower exists, on auxiliary streams beside the next tower's convolutions
        # (engine.spawn); final_a follows tower_a on the compute stream
        size = enc["x_a"].shape[-2:]
        heads = {}

        def start(key, final):
            def go(xt):
                # both heads on auxiliary stream 1: stream 0 belongs to the attention chains of the ResidualAConv blocks
                # (convolution.py), which would queue behind final_c's ~15 launches inside tower_b / tower_a
                heads[key] = E.spawn(lambda v: final(v, size=size, suffix="_" + key), [xt], 1)
            return go

        tow = self.tower_fusion(encoded=enc, decoded=dec, bufs=bufs,
                                after={"c": start("c", self.final_c), "b": start("b", self.final_b)})
        out_a = self.final_a(tow["x_tower_a"], suffix="_a")
        (br_c, out_c), (br_b, out_b) = heads["c"], heads["b"]
        E.join([br_c, br_b])
        if self.training:
            E._py_op(torch._foreach_add_, self.__dict__["_cn_nbt"], 1)  # BatchNorm bookkeeping (not arithmetic on the path)
        return self.final_combine(out_a, out_b, out_c, suffixes=["_a", "_b", "_c"])

    def input_var(self, x: torch.Tensor) -> E.Var:
        if x.dim() != 5:
            raise ValueError("x must be shaped (B, C, T, H, W)")
        B, C, Tn, H, W = x.shape
        if C != self.in_channels or Tn != self.in_time:
            raise ValueError(f"expected C={self.in_channels}, T={self.in_time}; got {C}, {Tn}")
        x = E._check(x).contiguous()
        return E.Var(x.view(B, C * Tn, H, W), False)

    def forward(self, x: torch.Tensor, latlon_coords: T.Optional[torch.Tensor] = None) -> T.Dict[str, torch.Tensor]:
        """x: (B, C, T, H, W) fp32 on the GPU -> {distance, edge, crop}: (B, 1, H, W) probabilities."""
        from .autograd_bridge import run_towerunet

        return run_towerunet(self, x)
