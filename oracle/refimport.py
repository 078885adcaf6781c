"""Stub-import the real reference hot path from /root/reference (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE ONLY. /root/reference does not exist on the GPU box, so
nothing that runs there may import this module; it is used by
``oracle/make_golden.py`` and by the ``not gpu`` oracle-vs-reference tests
(which skip when /root/reference is absent).

Recipe:
  1. register an empty ``cultionet`` package whose ``__path__`` points at the
     reference sources, so ``cultionet/__init__.py`` (which needs lightning) is
     skipped;
  2. stub the third-party modules that are absent here: ``cv2``, ``lightning``,
     ``torchmetrics``, ``cultionet.data`` (10-line ``Data``) and ``natten``
     (``oracle.na2d_ref``);
  3. disable dynamo so ``torch.compile`` at nunet.py:141 is a no-op.
No reference source is copied: the modules are executed from where they lie.
"""
from __future__ import annotations

import os
import sys
import types

REF_SRC = "/root/reference/src/cultionet"


def available() -> bool:
    return os.path.isdir(REF_SRC)


def import_reference():
    """Returns a namespace with TowerUNet, CultionetLitModel, losses, Data."""
    if "cultionet" in sys.modules and getattr(sys.modules["cultionet"], "_oracle_stub", False):
        return _namespace()
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    import torch
    import torch.nn as nn

    from . import na2d_ref

    pkg = types.ModuleType("cultionet")
    pkg.__path__ = [REF_SRC]
    pkg._oracle_stub = True
    sys.modules["cultionet"] = pkg

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))

    # lightning stub
    lightning = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    lightning.LightningModule = LightningModule
    lightning._oracle_stub = True
    prev_lightning = sys.modules.get("lightning")
    sys.modules["lightning"] = lightning

    try:  # a real torchmetrics pins the validation metrics; absent here => the restatement in oracle/metrics_ref.py
        import torchmetrics  # noqa: F401
    except Exception:
        from . import metrics_ref

        tm = types.ModuleType("torchmetrics")
        for name in ("MeanAbsoluteError", "MeanSquaredError", "FBetaScore", "MatthewsCorrCoef"):
            setattr(tm, name, getattr(metrics_ref, name))
        sys.modules["torchmetrics"] = tm

    # natten: the REAL package pins NA2D whenever it is importable (it is not, in this image); otherwise the stub
    # below routes the reference to our restatement (the only non-torch arithmetic on the path)
    try:
        import natten as _real_natten  # noqa: F401

        have_natten = hasattr(_real_natten, "NeighborhoodAttention2D")
    except Exception:
        have_natten = False
    natten = types.ModuleType("natten")
    natten.NeighborhoodAttention2D = na2d_ref.NeighborhoodAttention2D
    nf = types.ModuleType("natten.functional")
    nf.na2d = na2d_ref.na2d
    nf.na2d_qk = na2d_ref.na2d_qk
    nf.na2d_av = na2d_ref.na2d_av
    natten.functional = nf
    if not have_natten:
        sys.modules["natten"] = natten
        sys.modules["natten.functional"] = nf

    # cultionet.data stub: the in-memory contract of data/data.py:51-139
    data_mod = types.ModuleType("cultionet.data")

    class Data:
        def __init__(self, x, y=None, **kwargs):
            self.x = x
            self.y = y
            for k, v in kwargs.items():
                setattr(self, k, v)

        @property
        def num_samples(self):
            return self.x.shape[0]

    data_mod.Data = Data
    sys.modules["cultionet.data"] = data_mod
    try:
        return _namespace()
    finally:
        # The stub is only needed while the reference modules are being imported (they bind LightningModule at class
        # creation). Leaving it in sys.modules made a later ``import cultionet_amd.lightning`` subclass the STUB
        # (no load_from_checkpoint): the suite passed only in alphabetical order.
        if sys.modules.get("lightning") is lightning:
            if prev_lightning is not None:
                sys.modules["lightning"] = prev_lightning
            else:
                del sys.modules["lightning"]


def _namespace():
    import importlib

    ns = types.SimpleNamespace()
    ns.nunet = importlib.import_module("cultionet.models.nunet")
    ns.lightning = importlib.import_module("cultionet.models.lightning")
    ns.losses = importlib.import_module("cultionet.losses")
    ns.Data = sys.modules["cultionet.data"].Data
    ns.TowerUNet = ns.nunet.TowerUNet
    ns.CultionetLitModel = ns.lightning.CultionetLitModel
    return ns
