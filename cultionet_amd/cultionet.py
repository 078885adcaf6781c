"""CultioNet wrapper (mirror of /root/reference/src/cultionet/models/cultionet.py:12-110)."""
from __future__ import annotations

import typing as T

import torch
import torch.nn as nn

from .data import Data
from .enums import AttentionTypes, InferenceNames, ModelTypes, ResBlockTypes
from .nunet import TowerUNet


class CultioNet(nn.Module):
    def __init__(self, in_channels: int, in_time: int, hidden_channels: int = 32,
                 model_type: str = ModelTypes.TOWERUNET, activation_type: str = "SiLU", dropout: float = 0.1,
                 dilations: T.Union[int, T.Sequence[int]] = None, res_block_type: str = ResBlockTypes.RESA,
                 attention_weights: str = AttentionTypes.NATTEN, pool_by_max: bool = False,
                 batchnorm_first: bool = False, use_latlon: bool = False):
        super().__init__()
        self.in_channels, self.in_time, self.hidden_channels = in_channels, in_time, hidden_channels
        assert model_type in (ModelTypes.TOWERUNET), "The model type is not supported."
        self.mask_model = TowerUNet(in_channels=in_channels, in_time=in_time, hidden_channels=hidden_channels,
                                    num_classes=1, attention_weights=attention_weights,
                                    res_block_type=res_block_type, dropout=dropout, dilations=dilations,
                                    activation_type=activation_type, edge_activation=True, mask_activation=True,
                                    pool_by_max=pool_by_max, batchnorm_first=batchnorm_first, use_latlon=use_latlon)

    def forward(self, batch: Data) -> T.Dict[str, T.Optional[torch.Tensor]]:
        # cultionet.py:88-94 builds latlon_coords from batch.lon/lat; unused because use_latlon=False
        out = self.mask_model(batch.x)
        out.update({InferenceNames.CROP_TYPE: None, InferenceNames.CLASSES_L2: None, InferenceNames.CLASSES_L3: None})
        return out
