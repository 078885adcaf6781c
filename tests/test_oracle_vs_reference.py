"""Oracle vs the REAL reference, stub-imported from /root/reference (this container only;
skipped on the GPU box where /root/reference does not exist)."""
import pytest
import torch

from oracle import refimport
from oracle import towerunet_oracle as O

pytestmark = pytest.mark.skipif(not refimport.available(), reason="/root/reference not present")


@pytest.mark.parametrize("with_mask", [False, True])
def test_oracle_equals_reference(with_mask):
    ns = refimport.import_reference()
    torch.set_float32_matmul_precision("highest")
    ref = ns.CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    mine = O.TowerUNet(3, 12, hidden_channels=8)
    keys_ref = [k.replace("cultionet_TowerUNet.mask_model.", "") for k in ref.state_dict()]
    assert keys_ref == list(mine.state_dict().keys())
    ref.load_state_dict(O.seeded_state_dict(ref.state_dict()))
    mine.load_state_dict(O.seeded_state_dict(mine.state_dict()))
    x, y, bd = O.seeded_batch(2, height=28, width=28, with_mask=with_mask)
    batch = ns.Data(x=x, y=y, bdist=bd, lon=torch.zeros(2), lat=torch.zeros(2))
    pr = ref(batch)
    lr, _ = ref.calc_loss(batch, pr)
    lr.backward()
    pm = mine(x)
    lm, _ = O.calc_loss(pm, y, bd)
    lm.backward()
    assert abs(lr.item() - lm.item()) <= 1e-6
    for k in ("distance", "edge", "crop"):
        assert (pr[k] - pm[k]).abs().max() <= 1e-6
    for (_, p1), (_, p2) in zip(ref.named_parameters(), mine.named_parameters()):
        assert (p1.grad - p2.grad).abs().max() <= 1e-6
