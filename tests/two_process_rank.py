"""One rank of the two-process GPU tests (started by tests/rank_launcher.py, one process per rank, ALL ON GPU 0): the
row-partitioned solver with its collectives through gloo (RCCL refuses two ranks on one device) and its halo exchanges as
peer-to-peer stores into mailboxes shared through hipIpc -- the path one process per GPU takes on a multi-GPU node.

    python two_process_rank.py compare | die
Prints one line "RESULT {json}" (mode compare, every rank).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from padne_amd import _hip, distributed, synthetic  # noqa: E402


class Dist:
    """torch.distributed as DistributedSolver uses it, with a counter on the all-gathers of the library's collectives; in
    mode `die` rank 1 leaves the job without a word after the N-th one -- in the middle of a solve."""

    def __init__(self, die_after=None):
        self.calls = 0
        self.die_after = die_after

    def get_backend(self):
        return dist.get_backend()

    def all_gather(self, parts, t):
        self.calls += 1
        if self.die_after is not None and self.calls > self.die_after:
            os._exit(17)
        return dist.all_gather(parts, t)

    def all_gather_object(self, parts, obj):
        return dist.all_gather_object(parts, obj)

    def broadcast(self, t, src=0):
        return dist.broadcast(t, src=src)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
    ctx = _hip.Context(0)                                      # every rank on the same device
    n_layers = max(4, world)
    sysm = synthetic.layered_system(n_layers, 150, 120, via_lattice=6)
    plan = distributed.build_layer_partition(sysm, rank, world)
    d = Dist(die_after=40 if (mode == "die" and rank == 1) else None)
    ds = distributed.DistributedSolver(ctx, plan, dist=d)
    out = {"rank": rank, "p2p": bool(ds.p2p)}
    if mode == "die":
        res = ds.solve(rtol=1e-12)                             # rank 1 exits inside; rank 0 must come back with an error
        print("RESULT " + json.dumps({"rank": rank, "unexpected": "the solve returned", "iterations": int(res.iterations)}), flush=True)
        return 0
    res = ds.solve(rtol=1e-12)                                 # builds the hierarchy
    sol = ds.solution().copy()
    c0 = ctx.comm_call_counts()[0]
    res2 = ds.solve(rtol=1e-12)                                # the same solve with the hierarchy in place: what the iterations cost
    c1 = ctx.comm_call_counts()[0]
    assert res2.iterations == res.iterations and np.array_equal(ds.solution(), sol), "the second solve differs from the first"
    out["iterations"] = int(res.iterations)
    out["rel_residual"] = float(res.rel_residual)
    out["calls_per_solve_p2p"] = [int(b - a) for a, b in zip(c0, c1)]
    out["split_tiles"] = [int(x) for x in ds.A.split_tiles(-1)]
    # the same with the exchanges as all-gathers (the mailboxes stay mapped, the library does not use them)
    os.environ["PADNE_NO_P2P"] = "1"
    ctx.reload_options()                                       # (the switches are read once per context)
    c0 = ctx.comm_call_counts()[0]
    res3 = ds.solve(rtol=1e-12)
    c1 = ctx.comm_call_counts()[0]
    del os.environ["PADNE_NO_P2P"]
    ctx.reload_options()
    out["calls_per_solve_allgather"] = [int(b - a) for a, b in zip(c0, c1)]
    out["iterations_allgather"] = int(res3.iterations)
    sol_ag = ds.solution().copy()
    # what one exchange costs on the device: peer-to-peer stores + flags, and the all-gather (through gloo and the host here)
    out["p2p_exchange_us"] = ctx.halo_exchange_time(300) * 1e6
    out["exchange_slots_per_rank"] = int(plan.m)
    os.environ["PADNE_NO_P2P"] = "1"
    ctx.reload_options()
    out["allgather_exchange_us_gloo"] = ctx.halo_exchange_time(30) * 1e6
    del os.environ["PADNE_NO_P2P"]
    ctx.reload_options()
    out["max_abs_difference"] = float(np.abs(sol - sol_ag).max())
    out["bit_identical"] = bool(np.array_equal(sol, sol_ag))
    # potentials of all ranks against the reference's direct solve on the oracle-assembled system (rank 0)
    parts = [None] * world
    dist.all_gather_object(parts, (ds.owned_reduced_global, sol))
    if rank == 0:
        from oracle import padne_oracle as O
        els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
        els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
        Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
        v_ref = O.solve_system(Lo, ro)[0][:sysm.n_vertices]
        v = np.zeros(sysm.n_vertices)
        for idx, xs in parts:
            v[idx] = xs
        out["rel_error_vs_direct_solve"] = float(np.abs(v - v_ref).max() / np.abs(v_ref).max())
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    ctx.p2p_close()
    ctx.close()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
