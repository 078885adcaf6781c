"""CultionetLitModel on the HIP engine.

Host-side mirror of /root/reference/src/cultionet/models/lightning.py for the hot path: the constructor
keeps the reference's 24 keyword arguments (lightning.py:822-847), the model lives under the attribute
``f"{model_name}_{model_type}"`` (state-dict prefix ``cultionet_TowerUNet.mask_model.``), and
``forward / predict_step / get_true_labels / calc_loss / training_step / configure_optimizers`` keep their
signatures and return types, so ``cultionet.fit`` / ``predict_lightning`` can drive it unchanged.

Two ways to train:
  * **drop-in** (lightning.Trainer): ``training_step`` returns a torch scalar whose ``backward()`` replays
    the HIP tape through cultionet_amd.autograd_bridge; torch optimizers / DDP work as usual.
  * **native** (``HipTrainer``, used by bench.py): forward + Tanimoto + backward + global-norm clip + fused
    AdamW entirely in HIP kernels on flat parameter/gradient buffers, RCCL all-reduce of the flat gradient
    overlapped with the backward tape (cultionet_amd.ddp).
"""
from __future__ import annotations

import typing as T
from pathlib import Path

import torch

from . import engine as E
from .cultionet import CultioNet
from .data import Data
from .enums import (AttentionTypes, InferenceNames, LearningRateSchedulers, LossTypes, ModelTypes, ResBlockTypes,
                    ValidationNames)
from .losses import CombinedLoss, TanimotoComplementLoss, TanimotoDistLoss

try:  # the real LightningModule when lightning is installed (it is not in the build image)
    from lightning import LightningModule as _Base  # type: ignore

    if not hasattr(_Base, "load_from_checkpoint"):  # a test double left in sys.modules is not Lightning
        raise ImportError("lightning.LightningModule lacks load_from_checkpoint")
except Exception:  # pragma: no cover - exercised in the build image
    class _Base(torch.nn.Module):
        """Minimal stand-in providing the LightningModule methods this file uses."""

        def __init__(self):
            super().__init__()
            self.hparams: T.Dict[str, T.Any] = {}
            self.trainer = None

        def save_hyperparameters(self, *args, **kwargs):
            import inspect

            frame = inspect.currentframe().f_back
            av = inspect.getargvalues(frame)
            self.hparams = {k: av.locals[k] for k in av.args if k != "self"}

        def log(self, *args, **kwargs):
            pass

        def log_dict(self, *args, **kwargs):
            pass

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
            ckpt = torch.load(str(checkpoint_path), map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"])
            return model


# lightning.py:38-92 restricted to the losses selectable from the CLI (args.yml:436-442)
LOSS_DICT = {
    LossTypes.TANIMOTO_COMPLEMENT: {
        "classification": TanimotoComplementLoss(),
        "regression": TanimotoComplementLoss(transform_logits=False, one_hot_targets=False),
    },
    LossTypes.TANIMOTO: {
        "classification": TanimotoDistLoss(),
        "regression": TanimotoDistLoss(transform_logits=False, one_hot_targets=False),
    },
    LossTypes.TANIMOTO_COMBINED: {
        "classification": CombinedLoss(losses=[TanimotoDistLoss(), TanimotoComplementLoss()]),
        "regression": CombinedLoss(losses=[TanimotoDistLoss(transform_logits=False, one_hot_targets=False),
                                           TanimotoComplementLoss(transform_logits=False, one_hot_targets=False)]),
    },
}


class LightningModuleMixin(_Base):
    def __init__(self):
        super().__init__()

    def forward(self, batch: Data, batch_idx: int = None) -> T.Dict[str, torch.Tensor]:
        """distance / edge / crop probabilities, each (B, 1, H, W); crop_type, classes_l2, classes_l3 = None."""
        return self.cultionet_model(batch)

    @property
    def cultionet_model(self) -> CultioNet:
        return getattr(self, self.model_attr)

    @property
    def hip_precision(self) -> T.Optional[str]:
        """Explicit precision of the drop-in forward ("32-true" | "bf16-mixed" | "16-mixed"), or None (default): follow
        the torch.autocast region lightning.Trainer(precision=...) opens (model.py:168-186)."""
        return self.cultionet_model.mask_model.precision

    @hip_precision.setter
    def hip_precision(self, value: T.Optional[str]) -> None:
        if value not in (None, "32-true", "32", "bf16-mixed", "16-mixed"):
            raise ValueError(f"unsupported precision {value!r}")
        self.cultionet_model.mask_model.precision = value

    # ---- device-side input prologue ----------------------------------------------------------
    def set_norm_values(self, mean: T.Optional[torch.Tensor], std: T.Optional[torch.Tensor]) -> None:
        """Per-channel z-score statistics (NormValues, utils/normalize.py:63-82) applied by the device prologue."""
        self._norm_mean, self._norm_std = mean, std

    def on_after_batch_transfer(self, batch: Data, dataloader_idx: int = 0) -> Data:
        """Lightning hook, called once the batch is on the GPU. A batch whose ``x`` is still RAW (integer
        reflectances, as stored by the reference's .pt files) is scaled / clipped / z-scored here by one HIP pass
        (cn_prepare_chips_f32) instead of per sample on the CPU in EdgeDataset.get (data/datasets.py:443-446):
        collate the raw samples with cultionet_amd.data.collate_fn and let the device do the arithmetic. Float
        batches (already prepared upstream) pass through untouched."""
        x = getattr(batch, "x", None)
        if x is None or not x.is_cuda or x.dtype == torch.float32:
            return batch
        from .edges import SCALE_FACTOR, prepare_chips

        batch.x = prepare_chips(x, getattr(self, "_norm_mean", None), getattr(self, "_norm_std", None))
        bd = getattr(batch, "bdist", None)
        if bd is not None and bd.dtype != torch.float32:
            B = bd.shape[0]
            batch.bdist = prepare_chips(bd.reshape(B, 1, 1, *bd.shape[1:])).reshape(bd.shape)
        return batch

    def predict_step(self, batch: Data, batch_idx: int = None) -> T.Dict[str, torch.Tensor]:
        return self.forward(batch, batch_idx=batch_idx)

    @torch.no_grad()
    def get_true_labels(self, batch: Data, crop_type: torch.Tensor = None) -> T.Dict[str, T.Optional[torch.Tensor]]:
        """lightning.py:161-207. Kept for API compatibility (validation code reads these tensors); the training
        loss does not call it: the HIP loss kernel derives edge / crop / mask from ``batch.y`` on the fly."""
        y = batch.y
        ec = self.edge_class
        true_edge = (y == ec).long()
        true_crop = ((y > 0) & (y < ec)).long()
        mask = None
        if y.min() == -1:
            mask = (y != -1).long().unsqueeze(1)
        return {
            ValidationNames.TRUE_EDGE: true_edge,
            ValidationNames.TRUE_CROP: true_crop,
            ValidationNames.TRUE_CROP_AND_EDGE: (y > 0).long(),
            ValidationNames.TRUE_CROP_OR_EDGE: torch.where((y > 0) & (y < ec), 1, torch.where(y == ec, 2, 0)).long(),
            ValidationNames.TRUE_CROP_TYPE: torch.where(y == ec, 0, y).long() if crop_type is not None else None,
            ValidationNames.MASK: mask,
        }

    def _loss_terms(self, batch: Data):
        """(prediction key, kernel kwargs) of the three main losses of calc_loss (lightning.py:307-339).

        The weak-supervision mask ``y != -1`` is always applied in-kernel: when no -1 is present it is the
        identity, which removes the reference's device->host sync (``batch.y.min() == -1``, lightning.py:194).
        """
        y = batch.y
        if y.dtype != torch.int64:
            y = y.long()
        y = y.contiguous()
        ec = int(self.edge_class)
        return (
            (InferenceNames.DISTANCE, dict(target_f=batch.bdist.contiguous(), labels=y, target_mode=E.TGT_FLOAT,
                                           mask_mode=E.MSK_LABEL)),
            (InferenceNames.EDGE, dict(labels=y, target_mode=E.TGT_EQ, mask_mode=E.MSK_LABEL, klass=ec)),
            (InferenceNames.CROP, dict(labels=y, target_mode=E.TGT_RANGE, mask_mode=E.MSK_LABEL, klass=ec)),
        )

    def calc_loss(self, batch: T.Union[Data, T.List], predictions: T.Dict[str, torch.Tensor]):
        """lightning.py:209-354: (dist + edge + crop) / 3 and the report dict {dloss, eloss, closs}."""
        from .autograd_bridge import tanimoto_autograd

        kind = E.LOSS_KINDS[str(self.loss_name)]
        terms = []
        for key, kw in self._loss_terms(batch):
            terms.append(tanimoto_autograd(predictions[key], loss_kind=kind, **kw))
        loss = (terms[0] + terms[1] + terms[2]) / 3.0
        return loss, {"dloss": terms[0], "eloss": terms[1], "closs": terms[2]}

    def training_step(self, batch: Data, batch_idx: int = None):
        predictions = self(batch)
        loss, _ = self.calc_loss(batch, predictions)
        self.log("loss", loss, on_step=False, on_epoch=True, prog_bar=True, batch_size=batch.num_samples)
        return loss

    def probas_to_labels(self, x: torch.Tensor, thresh: float = 0.5) -> torch.Tensor:
        """lightning.py:126-136."""
        if x.shape[1] == 1:
            return x.gt(thresh).squeeze(dim=1).long()
        return x.argmax(dim=1).long()

    def _shared_eval_step(self, batch: Data, batch_idx: int = None) -> dict:
        """lightning.py:374-481: loss + the torchmetrics scores, computed on the device by ONE fused kernel
        (masked MAE / MSE, two 2x2 confusion matrices, micro F-beta = accuracy, MCC, checkpoint score): no
        torchmetrics round trips, no masked_select copies, no host sync."""
        from . import _lib

        with torch.no_grad():
            predictions = self(batch)
            loss, loss_report = self.calc_loss(batch, predictions)
            dist = predictions[InferenceNames.DISTANCE].float().contiguous()
            edge = predictions[InferenceNames.EDGE].float().contiguous()
            crop = predictions[InferenceNames.CROP].float().contiguous()
            if edge.shape[1] != 1 or crop.shape[1] != 1:
                raise NotImplementedError("the fused metrics kernel covers single-channel edge / crop maps")
            y = batch.y if batch.y.dtype == torch.int64 else batch.y.long()
            y = y.contiguous()
            bdist = batch.bdist.float().contiguous()
            dev = dist.device
            counts = E.alloc(11, torch.float64, dev)
            out = E.alloc(7, torch.float32, dev)
            lossd = loss.detach().float().reshape(1).contiguous()
            _lib.call("cn_eval_metrics_f32", dist.data_ptr(), edge.data_ptr(), crop.data_ptr(), bdist.data_ptr(),
                      y.data_ptr(), int(self.edge_class), 0.5, y.numel(), lossd.data_ptr(), counts.data_ptr(),
                      out.data_ptr(), E._stream())
        metrics = {
            "loss": loss,
            "dist_mae": out[0],
            "dist_mse": out[1],
            "edge_f1": out[2],
            "crop_f1": out[3],
            "edge_mcc": out[4],
            "crop_mcc": out[5],
            "score": out[6],
        }
        metrics.update(loss_report)
        return metrics

    def validation_step(self, batch: Data, batch_idx: int = None) -> dict:
        """lightning.py:483-508 (``val_score`` is what callbacks.py:246 checkpoints on)."""
        eval_metrics = self._shared_eval_step(batch, batch_idx)
        metrics = {
            "vef1": eval_metrics["edge_f1"],
            "vcf1": eval_metrics["crop_f1"],
            "vmae": eval_metrics["dist_mae"],
            "val_score": eval_metrics["score"],
            "val_loss": eval_metrics["loss"],
            "val_dloss": eval_metrics["dloss"],
            "val_eloss": eval_metrics["eloss"],
            "val_closs": eval_metrics["closs"],
        }
        self.log_dict(metrics, on_step=False, on_epoch=True, prog_bar=True, batch_size=batch.num_samples)
        if self.save_batch_val_metrics:
            self._save_batch_metrics(metrics, getattr(self, "current_epoch", 0), batch)
        return metrics

    def _save_batch_metrics(self, metrics: T.Dict[str, torch.Tensor], epoch: int, batch: Data) -> None:
        """lightning.py:510-533: appends the batch metrics to <logger.save_dir>/batch_metrics.parquet."""
        import pandas as pd

        trainer = getattr(self, "trainer", None)
        if trainer is not None and getattr(trainer, "sanity_checking", False):
            return
        logger = getattr(self, "logger", None)
        save_dir = getattr(logger, "save_dir", None) if logger is not None else None
        if save_dir is None:
            return
        ids = list(getattr(batch, "train_id", []) or [])
        write_metrics = {"epoch": [epoch] * len(ids), "train_ids": ids}
        for k, v in metrics.items():
            write_metrics[k] = [float(v)] * len(ids)
        metrics_file = Path(save_dir) / "batch_metrics.parquet"
        df = pd.DataFrame(write_metrics)
        if metrics_file.is_file():
            df = pd.concat((pd.read_parquet(metrics_file), df), axis=0)
        df.to_parquet(metrics_file)

    def test_step(self, batch: Data, batch_idx: int = None) -> dict:
        """lightning.py:535-560. Upstream also reads ``edge_dice`` / ``crop_dice`` / ``edge_jaccard`` / ``crop_jaccard``
        there, keys its _shared_eval_step never produces (a KeyError in the reference): the keys that exist are
        logged."""
        eval_metrics = self._shared_eval_step(batch, batch_idx)
        metrics = {
            "test_loss": eval_metrics["loss"],
            "tmae": eval_metrics["dist_mae"],
            "tmse": eval_metrics["dist_mse"],
            "tef1": eval_metrics["edge_f1"],
            "tcf1": eval_metrics["crop_f1"],
            "temcc": eval_metrics["edge_mcc"],
            "tcmcc": eval_metrics["crop_mcc"],
            "test_score": eval_metrics["score"],
        }
        self.log_dict(metrics, on_step=False, on_epoch=True, prog_bar=True)
        return metrics

    def configure_scorer(self):
        """lightning.py:562-577. The four torchmetrics scorers (MAE, MSE, micro F-beta(2), MCC) are evaluated by the
        fused HIP metrics kernel in _shared_eval_step; the attributes are kept (torchmetrics objects when that package
        is installed) for code that reaches for them."""
        try:
            import torchmetrics  # type: ignore

            self.mae_scorer = torchmetrics.MeanAbsoluteError()
            self.mse_scorer = torchmetrics.MeanSquaredError()
            self.f_beta_scorer = torchmetrics.FBetaScore(task="multiclass", num_classes=2, beta=2.0)
            self.mcc_scorer = torchmetrics.MatthewsCorrCoef(task="multiclass", num_classes=2)
        except Exception:
            self.mae_scorer = self.mse_scorer = self.f_beta_scorer = self.mcc_scorer = None

    def configure_loss(self):
        """lightning.py:589-609."""
        if str(self.loss_name) not in [str(k) for k in LOSS_DICT]:
            raise NameError(f"loss {self.loss_name!r} has no HIP kernel; choose one of {[str(k) for k in LOSS_DICT]}")
        entry = {str(k): v for k, v in LOSS_DICT.items()}[str(self.loss_name)]
        self.reg_loss = entry.get("regression")
        self.cls_loss = entry.get("classification")

    # The reference's optimizer / scheduler choices as data (lightning.py:611-683): name -> (torch class, which of the
    # module's hyper-parameters it takes, fixed keyword arguments). AdamW's betas (0.9, 0.98) and eps are what
    # cultionet_amd.lightning.HipTrainer's fused kernel implements natively.
    _OPTIMIZERS = {
        "Adam": ("Adam", ("lr", "eps"), {}),
        "AdamW": ("AdamW", ("lr", "weight_decay", "eps"), {"betas": (0.9, 0.98)}),
        "RAdam": ("RAdam", ("lr", "weight_decay", "eps"), {"betas": (0.9, 0.99), "decoupled_weight_decay": True}),
        "SGD": ("SGD", ("lr", "weight_decay"), {"momentum": 0.9}),
    }

    def _make_scheduler(self, optimizer):
        """(scheduler, stepping interval) for ``self.lr_scheduler``; OneCycleLR alone steps per batch."""
        from torch.optim import lr_scheduler as sched

        name = str(self.lr_scheduler)
        if name == str(LearningRateSchedulers.ONE_CYCLE_LR):
            tr = self.trainer
            return sched.OneCycleLR(optimizer, max_lr=self.learning_rate, epochs=tr.max_epochs,
                                    steps_per_epoch=tr.estimated_stepping_batches), "step"
        per_epoch = {
            str(LearningRateSchedulers.COSINE_ANNEALING_LR): lambda: sched.CosineAnnealingLR(optimizer, T_max=20, eta_min=1e-5,
                                                                                             last_epoch=-1),
            str(LearningRateSchedulers.EXPONENTIAL_LR): lambda: sched.ExponentialLR(optimizer, gamma=0.5),
            str(LearningRateSchedulers.STEP_LR): lambda: sched.StepLR(optimizer, step_size=self.steplr_step_size, gamma=0.5),
        }
        if name not in per_epoch:
            raise NameError("The learning rate scheduler is not implemented in Cultionet.")
        return per_epoch[name](), "epoch"

    def configure_optimizers(self):
        """Lightning hook, same choices and error behaviour as the reference (lightning.py:611-683): a torch optimizer over
        the model's parameters + one scheduler monitored on ``val_score``. (The native path -- HipTrainer -- does not use
        this: its clip + AdamW + OneCycleLR run in cn_optim.hip.)"""
        recipe = self._OPTIMIZERS.get(str(self.optimizer))
        if recipe is None:
            raise NameError("Choose either 'AdamW' or 'SGD'.")
        cls_name, takes, fixed = recipe
        hyper = {"lr": self.learning_rate, "weight_decay": self.weight_decay, "eps": self.eps}
        optimizer = getattr(torch.optim, cls_name)(list(self.cultionet_model.parameters()),
                                                   **{k: hyper[k] for k in takes}, **fixed)
        scheduler, interval = self._make_scheduler(optimizer)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "name": "lr_sch", "monitor": "val_score", "interval": interval,
                                 "frequency": 1}}


class CultionetLitTransferModel(LightningModuleMixin):
    """lightning.py:686-818: transfer learning on a pretrained CultionetLitModel checkpoint.

    ``finetune``: "all" trains everything; "fc" freezes the network and unfreezes the ``mask_model.final_*`` heads;
    anything else (the default None) freezes the network and REPLACES final_a / final_b / final_c / final_combine by
    freshly initialised, trainable heads. Frozen parameters keep ``requires_grad=False`` (torch optimizers skip them);
    the HIP tape still writes their gradient slices, which nothing reads.
    """

    def __init__(
        self,
        pretrained_ckpt_file: T.Union[Path, str],
        in_channels: int,
        in_time: int,
        hidden_channels: int = 64,
        model_type: str = ModelTypes.TOWERUNET,
        dropout: float = 0.2,
        activation_type: str = "SiLU",
        dilations: T.Union[int, T.Sequence[int]] = None,
        res_block_type: str = ResBlockTypes.RESA,
        attention_weights: str = AttentionTypes.NATTEN,
        optimizer: str = "AdamW",
        loss_name: str = LossTypes.TANIMOTO_COMPLEMENT,
        learning_rate: float = 0.01,
        lr_scheduler: str = LearningRateSchedulers.ONE_CYCLE_LR,
        steplr_step_size: int = 5,
        weight_decay: float = 1e-3,
        eps: float = 1e-4,
        ckpt_name: str = "last_transfer",
        model_name: str = "cultionet_transfer",
        pool_by_max: bool = False,
        batchnorm_first: bool = False,
        class_counts: T.Optional[torch.Tensor] = None,
        edge_class: T.Optional[int] = None,
        scale_pos_weight: bool = False,
        save_batch_val_metrics: bool = False,
        finetune: T.Optional[str] = None,
    ):
        super().__init__()
        self.save_hyperparameters()
        from .nunet import init_conv_weights
        from .unet_parts import TowerUNetFinal, TowerUNetFinalCombine

        self.optimizer = optimizer
        self.loss_name = loss_name
        self.learning_rate = learning_rate
        self.lr_scheduler = lr_scheduler
        self.steplr_step_size = steplr_step_size
        self.weight_decay = weight_decay
        self.eps = eps
        self.ckpt_name = ckpt_name
        self.model_name = model_name
        self.in_time = in_time
        self.class_counts = class_counts
        self.scale_pos_weight = scale_pos_weight
        self.save_batch_val_metrics = save_batch_val_metrics
        self.finetune = finetune
        self.edge_class = edge_class if edge_class is not None else 2

        cultionet_model = CultionetLitModel.load_from_checkpoint(
            checkpoint_path=str(pretrained_ckpt_file)).cultionet_model
        if self.finetune != "all":
            self.freeze(cultionet_model)
            if self.finetune == "fc":
                for name, param in cultionet_model.named_parameters():
                    if name.startswith("mask_model.final_"):
                        param.requires_grad = True
            else:
                mm = cultionet_model.mask_model
                for attr, factor in (("final_a", 0), ("final_b", 2), ("final_c", 4)):
                    old = getattr(mm, attr)
                    new = TowerUNetFinal(in_channels=old.in_channels, num_classes=old.num_classes,
                                         activation_type=activation_type, resample_factor=factor)
                    new.apply(init_conv_weights)
                    setattr(mm, attr, new)
                fc = mm.final_combine
                new_fc = TowerUNetFinalCombine(num_classes=fc.num_classes, edge_activation=fc.edge_activation,
                                               mask_activation=fc.mask_activation)
                new_fc.apply(init_conv_weights)
                mm.final_combine = new_fc
                mm.__dict__["_cn_store"] = None  # modules were replaced: re-flatten the parameters on next use
        self.model_attr = f"{model_name}_{model_type}"
        setattr(self, self.model_attr, cultionet_model)
        # Upstream ALSO assigns ``self.cultionet_model = ...`` (lightning.py:742-744), which nn.Module.__setattr__ files
        # under _modules: its transfer checkpoints carry every tensor twice, as ``cultionet_model.*`` and as
        # ``cultionet_transfer_TowerUNet.*``. Registered under both names here too, so those checkpoints load strictly
        # and checkpoints written here hold the keys upstream expects (parameters() de-duplicates the shared module).
        self._modules["cultionet_model"] = cultionet_model
        self.configure_loss()
        self.configure_scorer()

    @property
    def is_transfer_model(self) -> bool:
        return True

    def freeze(self, layer):
        for param in layer.parameters():
            param.requires_grad = False

    def unfreeze(self, layer):
        for param in layer.parameters():
            param.requires_grad = True
        return layer


class CultionetLitModel(LightningModuleMixin):
    def __init__(
        self,
        in_channels: int,
        in_time: int,
        hidden_channels: int = 64,
        model_type: str = ModelTypes.TOWERUNET,
        dropout: float = 0.2,
        activation_type: str = "SiLU",
        dilations: T.Union[int, T.Sequence[int]] = None,
        res_block_type: str = ResBlockTypes.RESA,
        attention_weights: str = AttentionTypes.NATTEN,
        optimizer: str = "AdamW",
        loss_name: str = LossTypes.TANIMOTO_COMPLEMENT,
        learning_rate: float = 0.01,
        lr_scheduler: str = LearningRateSchedulers.ONE_CYCLE_LR,
        steplr_step_size: int = 5,
        weight_decay: float = 1e-3,
        eps: float = 1e-4,
        ckpt_name: str = "last",
        model_name: str = "cultionet",
        pool_by_max: bool = False,
        batchnorm_first: bool = False,
        class_counts: T.Optional[torch.Tensor] = None,
        edge_class: T.Optional[int] = None,
        scale_pos_weight: bool = False,
        save_batch_val_metrics: bool = False,
    ):
        super().__init__()
        self.save_hyperparameters()
        self.optimizer = optimizer
        self.loss_name = loss_name
        self.learning_rate = learning_rate
        self.lr_scheduler = lr_scheduler
        self.steplr_step_size = steplr_step_size
        self.weight_decay = weight_decay
        self.eps = eps
        self.ckpt_name = ckpt_name
        self.model_name = model_name
        self.in_time = in_time
        self.class_counts = class_counts
        self.scale_pos_weight = scale_pos_weight
        self.save_batch_val_metrics = save_batch_val_metrics
        self.edge_class = edge_class if edge_class is not None else 2
        self.model_attr = f"{model_name}_{model_type}"
        setattr(self, self.model_attr, CultioNet(
            in_channels=in_channels, in_time=in_time, hidden_channels=hidden_channels, model_type=model_type,
            dropout=dropout, activation_type=activation_type, dilations=dilations, res_block_type=res_block_type,
            attention_weights=attention_weights, pool_by_max=pool_by_max, batchnorm_first=batchnorm_first))
        self.configure_loss()
        self.configure_scorer()

    @property
    def is_transfer_model(self) -> bool:
        return False


class HipTrainer:
    """Native training step: every FLOP and byte of forward + loss + backward + clip + AdamW in HIP kernels.

    Semantics of lightning.Trainer(gradient_clip_val=1.0) + AdamW(lr, wd, eps, betas=(0.9, 0.98)) as configured
    by the reference (model.py:84,168-186; lightning.py:622-629). Learning-rate schedule: ``total_steps`` given
    => the reference's default OneCycleLR stepped per optimizer step (lightning.py:657-664; it also cycles beta1,
    see cultionet_amd.schedules); otherwise ``lr_fn(step) -> lr`` (or ``-> (lr, beta1)``), constant by default.
    With ``world_size > 1`` the flat gradient is all-reduced over RCCL in buckets overlapped with the backward
    tape (see cultionet_amd.ddp).
    """

    def __init__(self, lit: CultionetLitModel, gradient_clip_val: T.Optional[float] = 1.0,
                 lr_fn: T.Optional[T.Callable[[int], T.Union[float, T.Tuple[float, float]]]] = None, comm=None,
                 total_steps: T.Optional[int] = None, precision: str = "32-true", replay: bool = False):
        self.lit = lit
        # replay=True: forward + loss + backward run from a recorded launch plan after the first steps
        # (cultionet_amd/replay.py): the Python of a step drops from ~8-10 ms to ~1.5 ms, which is what bounds the step at
        # the reference's default batch of 4 in mixed precision. Dropout replays too (the per-step part of the mask
        # seeds is a device word the plan's first launch bumps); a communicator keeps the step eager.
        self.replay = bool(replay)
        self._plan = None
        self._plan_key = None
        self._eager_steps = 0
        self.model = lit.cultionet_model.mask_model
        self.store = self.model.param_store()
        dev = self.store.flat.device
        self.exp_avg = torch.zeros_like(self.store.flat)
        self.exp_avg_sq = torch.zeros_like(self.store.flat)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.total = torch.zeros(1, dtype=torch.float32, device=dev)
        self.clip = gradient_clip_val
        if lr_fn is None:
            from .schedules import ConstantLR, OneCycleLR

            if total_steps is not None:
                if str(lit.lr_scheduler) != str(LearningRateSchedulers.ONE_CYCLE_LR):
                    raise NotImplementedError("the native step schedules OneCycleLR (the reference default); pass lr_fn")
                lr_fn = OneCycleLR(lit.learning_rate, total_steps)
            else:
                lr_fn = ConstantLR(lit.learning_rate)
        self.lr_fn = lr_fn
        self.step_count = 0
        self.comm = comm
        # lightning.Trainer(precision=...) of the reference (model.py:168-186; default "16-mixed"): on MI355X the
        # mixed mode is bf16 activations + fp32 master weights / statistics / accumulation (no loss scaling needed)
        if precision not in ("32-true", "32", "bf16-mixed", "16-mixed"):
            raise ValueError(f"unsupported precision {precision!r}")
        self.bf16 = precision in ("bf16-mixed", "16-mixed")
        if self.bf16 and not getattr(self.model, "mixed_precision_ok", True):
            import warnings

            warnings.warn("cultionet_amd: the mixed-precision path needs channel counts that are multiples of 8 "
                          "(hidden_channels % 8 == 0); this trainer runs in fp32", stacklevel=2)
            self.bf16 = False
        if lit.optimizer != "AdamW":
            raise NotImplementedError("the fused HIP optimizer implements AdamW (the reference default)")
        if any(not p.requires_grad for p in self.store.params):
            raise NotImplementedError("frozen parameters (CultionetLitTransferModel): the fused optimizer updates the "
                                      "whole flat buffer; train transfer models through the drop-in (torch optimizer) mode")
        if comm is not None:
            # torch DDP (the reference's strategy="ddp", model.py:101,184) broadcasts rank 0's parameters and buffers
            # at construction; dropout masks must differ per rank (each rank draws its own torch RNG stream upstream)
            comm.sync_initial_state(self.store, self.model)
            E.manual_seed(E._rng["seed"] + 0x9E37 * comm.rank)

    def forward_backward(self, batch: Data) -> torch.Tensor:
        """Forward + loss + backward; leaves d(loss)/d(params) in store.flat_grad. Returns the loss (1-elem tensor)."""
        if self.replay and self.comm is None and self.model.training:
            from . import replay as R

            key = R.step_key(self, batch)
            if self._plan is not None and self._plan.key == key:
                R.replay_step(self._plan, batch)
                self.last_outputs = self._plan.outputs
                return self.total
            if self._plan_key != key:  # new shapes / stream: two eager steps first (workspaces grow, packs are built)
                self._plan_key, self._eager_steps, self._plan = key, 0, None
            if self._eager_steps >= 2:  # everything lazily created exists by now: record this step
                plan = R.record_step(self, batch, self._forward_backward_eager)
                self._plan = plan if plan.key is not None else None  # (None: a scratch buffer moved while recording)
                return self.total
        self._eager_steps += 1
        return self._forward_backward_eager(batch)

    def _forward_backward_eager(self, batch: Data) -> torch.Tensor:
        from . import _lib

        lit, store = self.lit, self.store
        kind = E.LOSS_KINDS[str(lit.loss_name)]
        with E.using_store(store), E.recording(True) as tape, E.mixed_precision(self.bf16):
            outs = self.model.forward_vars(self.model.input_var(batch.x))
            self.last_outputs = {k: v.t for k, v in outs.items()}  # distance / edge / crop of this step (no copies)
            # the three losses in one launch per pass; self.total = (dist + edge + crop) / 3 is written by the kernel
            terms = lit._loss_terms(batch)
            self.last_losses = E.tanimoto_loss_multi([outs[key] for key, _ in terms], [kw for _, kw in terms],
                                                     loss_kind=kind, weights=[1.0 / 3.0] * len(terms), total=self.total)
            store.zero_grad()
            if self.comm is not None:
                self.comm.backward(tape, store)
            else:
                tape.backward()
        return self.total

    def optimizer_step(self) -> None:
        from . import _lib

        lit, store = self.lit, self.store
        self.step_count += 1
        s = E._stream()
        scale = 1.0 / self.comm.world_size if self.comm is not None else 1.0
        sumsq = None
        if self.clip is not None:
            _lib.call("cn_grad_sumsq_f32", store.flat_grad.data_ptr(), store.numel, self.sumsq.data_ptr(), s)
            sumsq = self.sumsq.data_ptr()
        sched = self.lr_fn(self.step_count)
        lr, beta1 = sched if isinstance(sched, tuple) else (sched, 0.9)
        _lib.call("cn_adamw_step_f32", store.flat.data_ptr(), store.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), store.numel, float(lr), float(beta1), 0.98,
                  float(lit.eps), float(lit.weight_decay), self.step_count, scale, sumsq,
                  float(self.clip) if self.clip is not None else 0.0, s)
        store.bump()

    def training_step(self, batch: Data) -> torch.Tensor:
        loss = self.forward_backward(batch)
        self.optimizer_step()
        return loss
