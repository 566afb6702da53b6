"""Where the drop-in pays off: `solve_system(L, r)` on host vectors against the reference's own solve
step (tocsc + spsolve + residual, solver.py:772-775) at ~11 k, ~100 k and ~1 M unknowns, every CPU time measured live on
this host (the bench's `small` block times the 1 M CPU solve only here: it takes minutes of one core).
    python scripts/small_sizes.py  ->  gpurun_out/r06_small.json  (copied to profiles/r06_small.json)"""
import json, os, platform, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from padne_amd import _hip

ctx = _hip.Context(0)
rows = bench.small_block(ctx, live_cpu_limit=10 ** 9)
ctx.close()
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    cpu = platform.processor()
doc = {"host": {"cpu": cpu, "cores": os.cpu_count()}, "small": rows,
       "what": "bench.small_block(live_cpu_limit=inf): hip = padne_amd.solver.solve_system(L, r), host r in / host v out, plan, "
               "reduced matrix, hierarchy rebuilt per call (median of 5); cpu = oracle assembly, then the reference's tocsc + "
               "spsolve + residual on one core (SuperLU is single-threaded)"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(doc, open("gpurun_out/r06_small.json", "w"), indent=1)
for r in rows:
    print(r)
