"""Robustness sweep (round 6): one native training step + one eval forward of TowerUNet over widths, batches, chip sizes,
time lengths and both precisions that the fixtures and the benchmark do not cover -- every launch must either run or be
refused by an engine-level error, never fail inside the library (a CN_ERR_LDS on a non-default width was found this way).
fp32 losses are compared with the CPU oracle at 1e-4 where the oracle is cheap (hidden <= 16).

    python tools/shape_sweep.py            (GPU box)
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cultionet_amd  # noqa: E402

cultionet_amd.configure_runtime()
import torch  # noqa: E402

from cultionet_amd import synthetic as S  # noqa: E402
from cultionet_amd.data import Data  # noqa: E402
from cultionet_amd.lightning import CultionetLitModel, HipTrainer  # noqa: E402

CASES = [
    # hidden, batch, channels, time, height, width
    (8, 1, 3, 12, 100, 100), (8, 3, 3, 12, 100, 100), (16, 2, 3, 12, 100, 100), (64, 1, 3, 12, 100, 100),
    (32, 1, 3, 12, 100, 100), (32, 5, 3, 12, 100, 100), (16, 2, 4, 25, 64, 64), (32, 2, 3, 12, 50, 50),
    (32, 2, 3, 12, 120, 100), (16, 2, 3, 12, 75, 110), (8, 2, 5, 6, 36, 36), (48, 2, 3, 12, 100, 100),
    (32, 1, 4, 25, 256, 256), (24, 2, 3, 12, 100, 100), (40, 2, 3, 12, 60, 60), (96, 1, 3, 12, 50, 50),
    (12, 2, 3, 12, 64, 64), (8, 2, 3, 12, 28, 28),
]


def main():
    dev = torch.device("cuda:0")
    bad = 0
    for prec in ("32-true", "bf16-mixed"):
        for (hid, B, C, Tn, H, W) in CASES:
            # (hidden % 8 != 0 under "bf16-mixed": the trainer warns once and runs fp32)
            tag = f"{prec} hidden {hid} B {B} [{C},{Tn},{H},{W}]"
            try:
                torch.manual_seed(0)
                lit = CultionetLitModel(in_channels=C, in_time=Tn, hidden_channels=hid, dropout=0.0)
                model = lit.cultionet_model.mask_model
                model.load_state_dict(S.seeded_state_dict(model.state_dict()))
                lit = lit.to(dev).train()
                tr = HipTrainer(lit, gradient_clip_val=1.0, precision=prec)
                x, y, bd = S.seeded_batch(B, channels=C, time=Tn, height=H, width=W, seed=5, with_mask=True)
                batch = Data(x=x.to(dev), y=y.to(dev), bdist=bd.to(dev))
                l1 = float(tr.training_step(batch).item())
                l2 = float(tr.training_step(batch).item())
                torch.cuda.synchronize()
                ok = l1 == l1 and l2 == l2 and abs(l1) < 1e3
                note = ""
                if prec == "32-true" and hid <= 16 and H * W <= 10000:
                    from oracle import towerunet_oracle as O

                    m = O.TowerUNet(C, Tn, hidden_channels=hid)
                    m.load_state_dict(O.seeded_state_dict(m.state_dict()))
                    m.train()
                    lo, _ = O.calc_loss(m(x), y, bd)
                    d = abs(float(lo) - l1)
                    note = f" |loss - oracle| {d:.2e}"
                    ok = ok and d <= 1e-4
                lit.eval()
                with torch.no_grad():
                    out = lit(batch)
                torch.cuda.synchronize()
                ok = ok and all(torch.isfinite(v.float()).all().item() for v in out.values() if torch.is_tensor(v))
                print(("ok   " if ok else "BAD  ") + tag + f" loss {l1:.5f} -> {l2:.5f}" + note, flush=True)
                bad += 0 if ok else 1
                del tr, lit, model
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("FAIL " + tag + f": {type(e).__name__}: {str(e)[:300]}", flush=True)
                traceback.print_exc(limit=3)
    print(f"{bad} failing configuration(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
