"""Drop-in check against the REAL lightning.Trainer (the reference trains through
lightning.Trainer(...).fit(lit_model, datamodule), model.py:168-186).

Lightning is not installed in the build image nor on the GPU box, so this test skips there; it runs wherever a
maintainer has cultionet's own environment (lightning >= 2.1). What it pins: CultionetLitModel is accepted as a
LightningModule by Trainer.fit, the hooks the reference defines (training_step / validation_step /
configure_optimizers incl. the OneCycleLR "step" interval / on_validation_epoch_end) run for two optimizer steps
through the autograd bridge on the HIP kernels, parameters change, packed weights follow the optimizer's in-place
updates (second step's loss differs from the first), and val_score is logged for the checkpoint callback."""
import pytest
import torch

pytestmark = pytest.mark.gpu

L = pytest.importorskip("lightning", reason="lightning is not installed in this image (drop-in test needs the real Trainer)")


def _loader(n_batches: int, batch: int, hw: int):
    from cultionet_amd.data import Data, collate_fn
    from oracle import towerunet_oracle as O

    chips = []
    for i in range(n_batches * batch):
        x, y, bdist = O.seeded_batch(1, height=hw, width=hw, seed=100 + i)
        chips.append(Data(x=x, y=y, bdist=bdist, lon=torch.zeros(1), lat=torch.zeros(1)))
    return torch.utils.data.DataLoader(chips, batch_size=batch, shuffle=False, collate_fn=collate_fn)


def test_trainer_fit_two_steps():
    from cultionet_amd.lightning import CultionetLitModel

    torch.manual_seed(0)
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    before = [p.detach().clone() for p in lit.parameters()]
    losses = []

    class Grab(L.Callback):
        def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
            losses.append(float(outputs["loss"] if isinstance(outputs, dict) else outputs))

    trainer = L.Trainer(accelerator="gpu", devices=1, max_epochs=1, max_steps=2, gradient_clip_val=1.0,
                        gradient_clip_algorithm="norm", precision="32-true", logger=False,
                        enable_checkpointing=False, enable_progress_bar=False, num_sanity_val_steps=0,
                        callbacks=[Grab()])
    trainer.fit(lit, train_dataloaders=_loader(2, 2, 28), val_dataloaders=_loader(1, 2, 28))
    assert trainer.global_step == 2 and len(losses) == 2
    assert all(l == l and 0.0 < l < 2.0 for l in losses)  # finite Tanimoto losses
    assert losses[0] != losses[1]                         # the second step saw the updated (re-packed) weights
    changed = sum(int(not torch.equal(a.cpu(), b.detach().cpu())) for a, b in zip(before, lit.parameters()))
    assert changed > 0.9 * len(before)
    assert "val_score" in trainer.callback_metrics
