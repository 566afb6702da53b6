"""Config C5 (8 right-hand sides in lockstep on the N = 5 M matrix) as a profiling target: rocprofv3 -- python3 scripts/c5_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip
if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])      # another build of the library (same-box A/B)
ctx = _hip.Context(0)
for k in range(2):
    rec = bench.c5_block(ctx, 2)
    print({k2: (round(v, 2) if isinstance(v, float) else v) for k2, v in rec.items() if k2 != "workload"}, flush=True)
