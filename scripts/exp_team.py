"""Iteration counts of the layer-partitioned solver at world = 1..8 on ONE GPU (in-process team)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from padne_amd import _hip, distributed, synthetic

block = "--block" in sys.argv
sys.argv = [a for a in sys.argv if a != "--block"]
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
sysm = synthetic.config(name)
for world in [int(w) for w in (sys.argv[2:] or ["2", "4", "8"])]:
    team = _hip.LocalTeam(world)
    out = [None] * world
    def rank_main(rank):
        c = _hip.Context(0)
        plan = distributed.build_layer_partition(sysm, rank, world)
        ds = distributed.DistributedSolver(c, plan, team=team, block_preconditioner=block)
        for _ in range(3):          # like bench.py: every step rebuilds everything derived from the matrix
            c0 = c.comm_call_counts()
            t = time.perf_counter(); res = ds.solve(rtol=1e-12, precond="amg", rebuild=True); w = time.perf_counter() - t
            c1 = c.comm_call_counts()
        # the counters are per process: all ranks of a team add to them, so divide by the world size
        calls = [(b - a) / world for a, b in zip(c0[0], c1[0])]; nbytes = [(b - a) / world for a, b in zip(c0[1], c1[1])]
        out[rank] = (res, w, plan.m, calls, nbytes)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    res, w, m, calls, nbytes = out[0]
    print(f"[{name}] world={world}: {'block' if block else 'global'}-AMG iterations={res.iterations} levels={res.levels} relres={res.rel_residual:.2e} "
          f"halo m={m} setup {res.setup_seconds*1e3:.0f} ms solve {res.seconds*1e3:.0f} ms (shared-GPU wall {w*1e3:.0f} ms, not a timing); "
          f"collectives per rank per solve: all-reduce {calls[0]:.0f}, all-gather f64 {calls[1]:.0f} ({nbytes[1]/1e6:.1f} MB), "
          f"f32 {calls[2]:.0f} ({nbytes[2]/1e6:.1f} MB)", flush=True)
