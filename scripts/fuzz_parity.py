"""Randomised parity sweep (GPU): random multi-layer systems -- jittered grids of random sizes and conductances, random
via resistors, internal nodes, current sources, a forest of voltage sources, sometimes a regulator -- assembled by
the oracle in the reference's KKT layout, solved by the reference's direct solve and by the product path
(padne_amd.solver.solve_system on the same matrix: index reduction + device multigrid-PCG + multiplier recovery).
Prints the worst deviation.  python scripts/fuzz_parity.py [n_cases] [seed]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import padne_oracle as O
from padne_amd import solver, synthetic

def run(n_cases=40, seed0=0, verbose=True):
    worst = (0.0, None)
    worst_m = 0.0
    t_start = time.perf_counter()
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 * 1000 + case)
        n_layers = int(rng.integers(1, 5))
        meshes, offs = [], [0]
        for l in range(n_layers):
            nx, ny = int(rng.integers(12, 140)), int(rng.integers(12, 110))
            xy, tri = synthetic.jittered_grid(nx, ny, seed=int(rng.integers(1 << 30)))
            meshes.append((xy, tri, float(rng.choice([2082.5, 1041.25, 4165.0, 520.0]))))
            offs.append(offs[-1] + len(xy))
        n_vert = offs[-1]
        n_internal = int(rng.integers(0, 4))
        node = lambda: int(rng.integers(0, n_vert + n_internal))
        vert = lambda l: int(rng.integers(offs[l], offs[l + 1]))
        els = []
        for l in range(n_layers - 1):                               # vias: every layer is tied to the next one
            for _ in range(int(rng.integers(1, 30))):
                els.append(("R", vert(l), vert(l + 1), float(10 ** rng.uniform(-4, 0))))
        for k in range(n_internal):                                 # internal nodes hang on at least two resistors
            for _ in range(2):
                els.append(("R", n_vert + k, vert(int(rng.integers(n_layers))), float(10 ** rng.uniform(-3, 1))))
        for _ in range(int(rng.integers(0, 6))):
            a, b = node(), node()
            if a != b:
                els.append(("R", a, b, float(10 ** rng.uniform(-3, 2))))
        for _ in range(int(rng.integers(1, 5))):
            f, t = node(), node()
            if f != t:
                els.append(("I", f, t, float(rng.uniform(0.1, 5.0))))
        # voltage sources: a forest (no loops) -- each new source ties a node not yet touched by a source to any node
        n_extra = 0
        base = n_vert + n_internal
        tied = set()
        vs = []
        for _ in range(int(rng.integers(0, 4))):
            p, n = node(), node()
            if p == n or p in tied:
                continue
            tied.add(p); tied.add(n) if not vs else None
            vs.append((p, n, float(rng.uniform(0.5, 12.0))))
        uf = {}
        def find(x):
            while uf.get(x, x) != x:
                x = uf[x]
            return x
        for p, n, volt in vs:
            if find(p) == find(n):
                continue
            uf[find(p)] = find(n)
            els.append(("V", p, n, volt, base + n_extra)); n_extra += 1
        if rng.uniform() < 0.3 and n_layers >= 2:
            vp, vn, sf, st = vert(0), vert(1), vert(0), vert(1)
            if len({vp, vn, sf, st}) == 4 and find(vp) != find(vn):
                uf[find(vp)] = find(vn)
                els.append(("REG", vp, vn, sf, st, float(rng.uniform(1.0, 5.0)), float(rng.uniform(0.5, 1.5)), base + n_extra)); n_extra += 1
                els.append(("R", vp, vn, float(rng.uniform(0.5, 5.0))))     # a load, so that the regulator delivers something
        v_el = [e for e in els if e[0] == "V"]
        ground = max(v_el, key=lambda e: e[3])[2] if v_el else 0          # n of the largest source (solver.py:671-686)
        Lo, ro = O.assemble_system(meshes, n_internal, els, ground)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            try:
                v_ref, gc_ref, _ = O.solve_system(Lo, ro)
            except Exception as exc:                                         # singular for the reference too: skip
                print(f"case {case}: reference solve failed ({type(exc).__name__}), skipped", flush=True)
                continue
        if not np.all(np.isfinite(v_ref)):
            print(f"case {case}: reference returned non-finite values, skipped", flush=True)
            continue
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            v, info = solver.solve_system(Lo, ro)
        n_pot = n_vert + n_internal
        scale = max(np.abs(v_ref[:n_pot]).max(), 1e-300)
        dev = np.abs(v[:n_pot] - v_ref[:n_pot]).max() / scale
        # multipliers are currents: judged against the largest current in the system (a ground current of 1e-16 A next
        # to 0 A is not a deviation)
        i_scale = max(np.abs(v_ref[n_pot:]).max(), max([abs(e[3]) for e in els if e[0] == "I"], default=0.0), 1e-300)
        dev_m = np.abs(v[n_pot:] - v_ref[n_pot:]).max() / i_scale
        worst_m = max(worst_m, dev_m)
        kinds = "+".join(sorted({e[0] for e in els}))
        if verbose:
            print(f"case {case:3d}: layers {n_layers} N {Lo.shape[0]:6d} elements {len(els):3d} [{kinds}] iterations {info.iterations:3d} "
                  f"potentials {dev:.1e} multipliers {dev_m:.1e} residual {info.residual_norm:.1e}" + (f"  WARN {w[0].message}" if w else ""), flush=True)
        if dev > worst[0]:
            worst = (dev, case)
    if verbose:
        print(f"worst relative deviation of the potentials: {worst[0]:.2e} (case {worst[1]}), of the multipliers {worst_m:.2e}, "
              f"over {n_cases} cases in {time.perf_counter()-t_start:.0f} s")
    return worst[0], worst_m


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
