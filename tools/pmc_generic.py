"""Per-kernel raw PMC sums (per launch) from a rocprofv3 --pmc ... --output-format csv counter_collection.csv:
python tools/pmc_generic.py DIR/x_counter_collection.csv [name-substring]"""
import csv
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "").strip()
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    n = max(cnt[k].values())
    print(k, "launches", n)
    for c, val in sorted(v.items()):
        print(f"   {c:34s} {val / n:16.1f} per launch")
