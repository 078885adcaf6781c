"""Checkpoint-key compatibility with the reference (CPU only, no kernels).

tests/golden/state_dict_keys_h32.npz holds the state-dict keys and shapes of the REAL reference's CultionetLitModel
(default configuration, hidden 32) exactly as upstream writes them: upstream wraps ``pre_unet`` in torch.compile
(models/nunet.py:141), so its 26 time-reduction tensors are spelt ``...pre_unet._orig_mod.*``. A drop-in must
  * load such a checkpoint strictly (and the plain spelling too),
  * be able to WRITE that spelling (``TowerUNet.upstream_checkpoint_keys``) for upstream's strict load,
  * keep values intact on the round trip.
"""
import os

import numpy as np
import torch


def _fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "state_dict_keys_h32.npz"))
    return [str(k) for k in g["keys"]], [tuple(int(v) for v in s.split(",")) if s else () for s in g["shapes"]]


def _lit():
    from cultionet_amd.lightning import CultionetLitModel

    return CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)


def test_key_set_and_shapes_match_the_reference(golden_dir):
    keys, shapes = _fixture(golden_dir)
    assert len(keys) == 442 and sum("_orig_mod" in k for k in keys) == 26
    lit = _lit()
    sd = lit.state_dict()
    plain = [k.replace("pre_unet._orig_mod.", "pre_unet.") for k in keys]
    assert list(sd.keys()) == plain  # same tensors, same ORDER (optimizer state in checkpoints is positional)
    for k, shp in zip(plain, shapes):
        assert tuple(sd[k].shape) == shp, k


def test_loads_upstream_spelling_strictly_and_round_trips(golden_dir):
    from cultionet_amd import synthetic as S

    keys, shapes = _fixture(golden_dir)
    # an "upstream checkpoint": upstream's keys, key-seeded values
    up = {k: torch.empty(shp) for k, shp in zip(keys, shapes)}
    up = {k: v.to(up[k].dtype) for k, v in S.seeded_state_dict(up).items()}
    for k in up:
        if k.endswith("num_batches_tracked"):
            up[k] = torch.tensor(3, dtype=torch.int64)
    lit = _lit()
    res = lit.load_state_dict(dict(up), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = lit.cultionet_model.mask_model
    # values landed in the plain-named parameters
    got = lit.state_dict()
    for k, v in up.items():
        assert torch.equal(got[k.replace("pre_unet._orig_mod.", "pre_unet.")], v), k
    # the plain spelling loads too
    lit2 = _lit()
    res2 = lit2.load_state_dict(dict(got), strict=True)
    assert not res2.missing_keys and not res2.unexpected_keys
    # writing upstream's spelling: exactly the reference's key list, in order, values intact
    model.upstream_checkpoint_keys = True
    try:
        out = lit.state_dict()
    finally:
        model.upstream_checkpoint_keys = False
    assert sorted(out.keys()) == sorted(keys)
    for k, v in up.items():
        assert torch.equal(out[k], v), k
    # and a checkpoint file round trip through load_from_checkpoint (the reference's predict entry, model.py:396-398)
    import tempfile

    from cultionet_amd.lightning import CultionetLitModel

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "last.ckpt")
        torch.save({"state_dict": out, "hyper_parameters": {"in_channels": 3, "in_time": 12, "hidden_channels": 32,
                                                            "dropout": 0.0}}, path)
        lit3 = CultionetLitModel.load_from_checkpoint(path)
    got3 = lit3.state_dict()
    for k, v in up.items():
        assert torch.equal(got3[k.replace("pre_unet._orig_mod.", "pre_unet.")], v), k
