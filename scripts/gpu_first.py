"""First-contact GPU check: SpMV / assembly / PCG parity against the oracle + quick timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from padne_amd import _hip, synthetic as S
from oracle import padne_oracle as O

ctx = _hip.Context(0)
print("devices", _hip.device_count(), flush=True)

def system_arrays(sysm):
    xy = np.concatenate([m[0] for m in sysm.meshes]); tri = np.concatenate([m[1] for m in sysm.meshes])
    mvo = sysm.mesh_offsets; mto = np.concatenate([[0], np.cumsum([m[1].shape[0] for m in sysm.meshes])])
    sig = np.array([m[2] for m in sysm.meshes])
    return xy, tri, mvo, mto, sig

def stamps(sysm, N):
    a, b, r = sysm.resistors
    g = 1.0 / r
    rows = np.stack([a, a, b, b], 1).reshape(-1); cols = np.stack([a, b, b, a], 1).reshape(-1)
    vals = np.stack([-g, g, -g, g], 1).reshape(-1)
    gi = sysm.ground
    rows = np.concatenate([rows, [N - 1, gi]]); cols = np.concatenate([cols, [gi, N - 1]]); vals = np.concatenate([vals, [1.0, 1.0]])
    rhs = np.zeros(N); f, t, i = sysm.current_sources
    np.add.at(rhs, f, i); np.add.at(rhs, t, -i)
    return rows, cols, vals, rhs

for (nl, nx) in [(1, 60), (2, 64), (1, 300)]:
    sysm = S.layered_system(nl, nx, nx, via_lattice=4)
    nv = sysm.n_vertices; N = nv + 1
    xy, tri, mvo, mto, sig = system_arrays(sysm)
    rows, cols, vals, rhs = stamps(sysm, N)
    t = time.time(); Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals); ctx.synchronize(); t_asm = time.time() - t
    Lg = Ld.to_scipy()
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t_), float(i)) for f, t_, i in zip(*sysm.current_sources)]
    Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    Lo.sort_indices()
    same_struct = np.array_equal(Lo.indptr, Lg.indptr) and np.array_equal(Lo.indices, Lg.indices)
    print(f"[asm {nl}x{nx}^2] N={N} nnz gpu={Lg.nnz} oracle={Lo.nnz} struct_equal={same_struct} "
          f"bitwise_equal={same_struct and np.array_equal(Lo.data, Lg.data)} maxabs={abs(Lo-Lg).max():.3e} rhs_equal={np.array_equal(ro, rhs)} t={t_asm:.3f}s", flush=True)
    # spmv parity (bitwise vs scipy csr)
    x = np.random.default_rng(1).uniform(-1, 1, N)
    yg = Ld.matvec(x); yo = Lo @ x
    print("   spmv bitwise", np.array_equal(yg, yo), "maxabs", np.abs(yg - yo).max(), flush=True)
    # reduce + solve
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    A = Ld.reduce(imap, nv - 1, -1.0)
    Ao = (-Lo[:nv, :nv]).tocsr(); keep = np.setdiff1d(np.arange(nv), [sysm.ground]); Ao = Ao[keep][:, keep]
    Ag = A.to_scipy(); Ao.sort_indices()
    print("   reduce bitwise", np.array_equal(Ag.indptr, Ao.indptr) and np.array_equal(Ag.indices, Ao.indices) and np.array_equal(Ag.data, Ao.data), flush=True)
    b = -rhs[keep]
    res = A.solve_spd(b, rtol=1e-12)
    v, gc, rn = O.solve_system(Lo, ro)
    err = np.abs(res.x - v[keep]).max() / np.abs(v[:nv]).max()
    print(f"   pcg iters={res.iterations} restarts={res.restarts} relres={res.rel_residual:.2e} t={res.seconds:.3f}s relerr_vs_spsolve={err:.2e}", flush=True)
    pd_g = ctx.power_density(xy, tri, mvo, mto, sig, v[:nv])
    pd_o = np.concatenate([O.power_density(m[0], m[1], v[o:o + m[0].shape[0]], m[2]) for m, o in zip(sysm.meshes, mvo[:-1])])
    print("   power density bitwise", np.array_equal(pd_g, pd_o), np.abs(pd_g - pd_o).max(), flush=True)

# timings at scale: SpMV N=1M and N=10M (8 layers)
for name in ["C2", "C4"]:
    t = time.time(); sysm = S.config(name); nv = sysm.n_vertices; N = nv + 1
    xy, tri, mvo, mto, sig = system_arrays(sysm); rows, cols, vals, rhs = stamps(sysm, N)
    print(name, "generated", time.time() - t, flush=True)
    t = time.time(); Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals); ctx.synchronize(); print("  assemble", time.time() - t, Ld.shape, Ld.nnz, flush=True)
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    t = time.time(); A = Ld.reduce(imap, nv - 1, -1.0); ctx.synchronize(); print("  reduce", time.time() - t, A.shape, A.nnz, flush=True)
    x = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1])); y = ctx.empty(A.shape[0])
    for rep in range(3):
        ts = A.spmv_time(x, y, 5, 50)
        print(f"  spmv {ts*1e6:.1f} us  {A.spmv_bytes/ts/1e9:.1f} GB/s  ({A.spmv_bytes/ts/8e12*100:.1f}% of 8 TB/s)", flush=True)
    keep = np.setdiff1d(np.arange(nv), [sysm.ground])
    b = ctx.to_device(-rhs[keep]); xs = ctx.empty(A.shape[0])
    t = time.time(); res = A.solve_spd_dev(b, xs, rtol=1e-12, raise_on_fail=False); wall = time.time() - t
    print(f"  pcg iters={res.iterations} restarts={res.restarts} relres={res.rel_residual:.2e} dev={res.seconds:.3f}s wall={wall:.3f}s  us/iter={res.seconds/max(res.iterations,1)*1e6:.1f} status={res.status}", flush=True)
    del A, Ld, x, y, b, xs
