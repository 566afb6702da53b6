"""What the strip numbering costs a cold solve_system on systems too small for an x-window plan: bench.small_block with the
plan's strip_order switched off (monkeypatched) against the default.  python scripts/lab/exp_small_strip.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from padne_amd import _hip
ctx = _hip.Context(0)
base = bench.small_block(ctx, live_cpu_limit=0)
orig = _hip.KktPlan.__init__
def no_strips(self, L, n_potential, elim, tied, n_free, index_map=None, strip_order=False):
    orig(self, L, n_potential, elim, tied, n_free, index_map=index_map, strip_order=False)
_hip.KktPlan.__init__ = no_strips
off = bench.small_block(ctx, live_cpu_limit=0)
for a, b in zip(base, off):
    print(f"n={a['n']:8d}  strips: cold {a['hip_solve_system_ms']:.2f} ms cached {a['hip_solve_system_cached_plan_ms']:.2f} ms it {a['iterations']}"
          f"   |  as numbered: cold {b['hip_solve_system_ms']:.2f} ms cached {b['hip_solve_system_cached_plan_ms']:.2f} ms it {b['iterations']}")
