"""Tuning aid: time one conv shape (fwd / dgrad) under CN_DBG_SPLITS=k for several k (each in a fresh process)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
shape = sys.argv[1:4] if len(sys.argv) > 3 else ["128", "128", "25"]
for k in (1, 2, 3, 4, 6, 8, 12, 16):
    env = dict(os.environ, CN_DBG_SPLITS=str(k))
    out = subprocess.run([sys.executable, os.path.join(here, "kone_time.py"), *shape], env=env, capture_output=True, text=True)
    print(f"splits={k:2d}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
