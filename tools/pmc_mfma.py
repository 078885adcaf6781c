"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass of the bench command.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o m -- python3 bench.py ...
    python tools/pmc_mfma.py DIR/m_counter_collection.csv profiles/r02_pmc_mfma_f32.json

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the fraction of SIMD-cycles with the
matrix pipe busy while the kernel ran (MI355X_MICROARCH.md: the counter counts cycles, 64 per v_mfma_f32_32x32x2_f32,
32 per v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE sums the 8 XCDs). valu_per_mfma = non-MFMA VALU instructions per MFMA.
wait / issue_stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0].replace("void ", "").strip()


agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0 or mf <= 0:
        continue
    n = max(cnt[k].values())
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    im = v.get("SQ_INSTS_MFMA", 0.0)
    out[k] = {
        "launches": n,
        "mfma_busy": mf / (gui / 8.0 * 1024.0),
        "valu_per_mfma": (v.get("SQ_INSTS_VALU", 0.0) - im) / im if im else None,
        "gui_active_cycles_per_launch_per_xcd": gui / 8.0 / n,
        "wait": v.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
        "issue_stall": v.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
        "active": v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
    }
res = {"note": __doc__.split("\n\n")[1], "kernels": out}
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        json.dump(res, f, indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_launch_per_xcd"] * kv[1]["launches"])[:16]:
    print(f"{k[:52]:52s} n={v['launches']:5d} mfma_busy {v['mfma_busy']:.3f} valu/mfma "
          f"{(v['valu_per_mfma'] or 0):5.2f} wait {(v['wait'] or 0):.2f} stall {(v['issue_stall'] or 0):.2f}")
