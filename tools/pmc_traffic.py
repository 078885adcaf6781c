"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv \
                                gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so reads are DOUBLED;
WRITE_SIZE is exact for 16-byte streaming stores and float atomics. Infinity-Cache hits are counted as traffic.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.split("(")[0].replace("void ", "").strip()
    return name


def collect(path, counter):
    tot = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = tot[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return tot


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fk, fn = fetch.get(k, [0.0, 0])
        wk, wn = write.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        rd = 2.0 * fk * 1024.0 / n      # gfx950 correction: x2
        wr = wk * 1024.0 / n
        out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    res = {"note": "FETCH_SIZE (x2 on gfx950) + WRITE_SIZE, KiB -> bytes, averaged per launch over separate PMC passes",
           "kernels": out}
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(res, f, indent=1)
    rows = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    for k, v in rows[:25]:
        print(f"{k[:60]:60s} n={v['launches']:5d} rd {v['read_bytes_per_launch'] / 1e6:9.2f} MB  wr {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
