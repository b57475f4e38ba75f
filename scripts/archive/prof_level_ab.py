"""What the profiling events inside the timed region cost: bench.py's shard step with the default record level (every
contraction and every phase) against level 1 (contractions of at least 2 Gflop only).  python scripts/prof_level_ab.py [bench args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["--samples-total", "64"]
for rep in range(2):
    for lvl in ("2", "1"):   # 2 = every contraction and phase inside the timed region (the old way), 1 = the default
        env = dict(os.environ, HFMI_BENCH_PROF_LEVEL=lvl)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-cpu-baseline", "--no-check", "--no-literal"],
                             env=env, capture_output=True, text=True).stdout
        j = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        print("prof_level %s: %.3f ms/step, median %.3f, min %.3f" % (lvl, j["ms_per_step"], j["median_ms_per_step"], j["step_ms_min_max"][0]), flush=True)
