"""Tuning aid: time one conv shape for every (tile config, K split) pair (fresh process each)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
shape = sys.argv[1:4]
for cfg in (0, 1, 2):
    for k in (1, 2, 3, 4, 6, 8):
        env = dict(os.environ, CN_DBG_SPLITS=str(k), CN_DBG_CFG=str(cfg))
        out = subprocess.run([sys.executable, os.path.join(here, "kone_time.py"), *shape], env=env, capture_output=True, text=True)
        print(f"cfg={cfg} splits={k:2d}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:]}", flush=True)
