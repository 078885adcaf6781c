"""The default `python bench.py` line on a real GPU: ONE JSON line on stdout carrying the contract's fields, the roofline
and cpu_baseline objects, and the extra blocks (bf16 = BASELINE configs[2], predict = configs[4], feed) without errors.
Short run (3 timed steps, a 2-step CPU leg); the numbers themselves are the driver's business."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_has_every_block():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--cpu-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "train_chips_per_sec" and d["unit"] == "chips/s" and d["n_gpus"] == 1 and d["steps"] == 3
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["kernel_launches_per_step"] > 100
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    assert r["kernel"].startswith("cn_") and r["launches_per_step"] >= 1 and 0 < r["end_to_end_frac"] < 1
    # provenance: the PMC traffic comes from the newest committed passes that contain THIS dominant
    # kernel, next to the algorithmic bytes of its launches and the rocprof-average launch time frac can be recomputed from
    for rr in (r, d["bf16"]["roofline"]):
        assert rr["traffic"] is not None and rr["traffic_error"] is None, rr["traffic_error"]
        assert rr["algorithmic_bytes"] > 0 and 0.5 < rr["traffic_vs_algorithmic"] < 4.0, rr["traffic_vs_algorithmic"]
        assert rr["rocprof_avg_launch_us"] > 0 and 0 < rr["rocprof_frac"] < 1
        assert os.path.exists(os.path.join(ROOT, rr["traffic_source"].split(" ")[0]))
        assert os.path.exists(os.path.join(ROOT, rr["rocprof_source"].split(" ")[0]))
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "chips/s"
    assert d["value"] > 10 * c["value"]  # north_star: >= 10x the host-CPU reference on one MI355X
    assert d["loss_delta_vs_cpu"]["abs_delta"] <= d["loss_delta_vs_cpu"]["tolerance"]
    b = d["bf16"]
    assert "error" not in b and b["dtype"] == "bf16" and b["value"] > d["value"] and b["roofline"]["peak"] == 2500.0
    assert b["loss_delta_vs_cpu"]["abs_delta"] <= b["loss_delta_vs_cpu"]["tolerance"]
    pr = d["predict"]
    assert "error" not in pr and pr["unit"] == "pixels/s" and pr["value"] == pr["bf16_mixed"]["value"] > pr["fp32"]["value"]
    assert 0 < pr["bf16_mixed"]["roofline"]["frac"] < 1 and pr["bf16_mixed"]["batch4"]["value"] > 0
    assert d["feed"]["value"] > 0 and d["feed"]["host_bytes_per_step"] > 0
    # the data-parallel stream set on one GPU (one-rank RCCL group): both precisions, really through RCCL, buckets launched
    d1 = d["ddp1"]
    for dt in ("f32", "bf16"):
        assert "error" not in d1[dt], d1[dt]
        assert d1[dt]["rccl_ranks"] == 1 and d1[dt]["buckets_per_step"] >= 1 and d1[dt]["value"] > 0
        assert d1[dt]["comm_ms_exposed"] is not None
        assert d["config"][f"ddp1_{dt}_chips_per_s"] == d1[dt]["value"]
