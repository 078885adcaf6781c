"""cultionet_amd.synthetic (product side) must stay bit-identical to the oracle's seeded helpers."""
import torch

from cultionet_amd import synthetic as S
from oracle import towerunet_oracle as O


def test_seeded_helpers_identical():
    m = O.TowerUNet(3, 12, hidden_channels=8)
    a, b = S.seeded_state_dict(m.state_dict()), O.seeded_state_dict(m.state_dict())
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for kw in (dict(batch=2, height=28, width=28), dict(batch=1, with_mask=True, seed=9)):
        for u, v in zip(S.seeded_batch(**kw), O.seeded_batch(**kw)):
            assert torch.equal(u, v)
