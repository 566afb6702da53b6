"""scipy prototype: what rounding the multigrid cycle's operator VALUES to 16 bits does to the PCG iteration count
(f32 as shipped / f16 with a power-of-two scale per operator / bf16).  python scripts/exp_half_cycle.py [layers] [nx]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import amg_proto as AP
from oracle import padne_oracle as O
from padne_amd import synthetic as S

def quant(M, kind):
    M = M.tocsr().copy()
    d = M.data
    if kind == "f32":
        M.data = d.astype(np.float32).astype(np.float64)
    elif kind == "f16":
        s = 2.0 ** np.floor(np.log2(32768.0 / np.abs(d).max()))
        M.data = (d * s).astype(np.float16).astype(np.float64) / s
    elif kind == "bf16":
        u = d.astype(np.float32).view(np.uint32)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        M.data = u.view(np.float32).astype(np.float64)
    return M

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 160
sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
n = sysm.n_vertices
A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
lv = AP.build_hierarchy(A, coarse_n=600, omega=1.5 / 2.0)
for kind in ("f64", "f32", "f16", "f16+W", "A16", "A16+Wc", "A16+Wc+diag", "bf16"):
    W0 = None
    if kind.startswith("A16"):
        # only the level operators A_l in 16 bits (P / R exact); "+Wc": the fused up-leg W built from the ROUNDED A_0
        # (consistent with the down-leg); "+diag": the diagonal absorbs the rounding so that row sums are those of A
        def qa(a):
            aq = quant(a, "f16")
            if kind.endswith("diag"):
                aq = aq.tolil() if False else aq
                d = np.asarray(a.sum(1)).ravel() - np.asarray(aq.sum(1)).ravel()
                aq = (aq + sp.diags(d)).tocsr()
            return aq
        q = [(qa(a), p) for a, p in lv]
        q[-1] = (lv[-1][0], None)
        if "Wc" in kind:
            A0q, P0 = q[0]
            W0 = (P0 - 0.75 * sp.diags(1.0 / lv[0][0].diagonal()) @ (A0q @ P0)).tocsr()
        def M(rr, q=q, W0=W0):
            def cyc(l, bb):
                a, p = q[l]
                if p is None:
                    import scipy.sparse.linalg as sla
                    return sla.spsolve(a.tocsc(), bb)
                dinv = 1.0 / lv[l][0].diagonal()
                w = 0.75
                x = w * dinv * bb
                rr1 = bb - a @ x
                if l == 0 and W0 is not None:
                    return x + w * dinv * rr1 + W0 @ cyc(l + 1, p.T @ rr1)
                x = x + p @ cyc(l + 1, p.T @ rr1)
                return x + w * dinv * (bb - a @ x)
            return cyc(0, rr)
        x, it = AP.pcg(A, b, M)
        print(f"{kind}: PCG iterations {it}, true relres {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.2e}", flush=True)
        continue
    if kind == "f16+W":
        A0, P0 = lv[0]
        W0 = quant((P0 - 0.75 * sp.diags(1.0 / A0.diagonal()) @ (A0 @ P0)).tocsr(), "f16")
        kind = "f16"
    if kind == "f64":
        q = lv
    else:
        q = [(quant(a, kind), None if p is None else quant(p, kind)) for a, p in lv]
        q[-1] = (lv[-1][0], None)          # the coarsest level stays exact (the dense inverse)
    def M(rr, q=q, W0=W0):
        # V(1,1) with an exact coarsest solve
        def cyc(l, bb):
            a, p = q[l]
            if p is None:
                import scipy.sparse.linalg as sla
                return sla.spsolve(a.tocsc(), bb)
            dinv = 1.0 / lv[l][0].diagonal()
            w = 0.75
            x = w * dinv * bb
            rr1 = bb - a @ x
            if l == 0 and W0 is not None:      # the fused up-leg of the fine level: W rounded on its own (not symmetric to the down-leg)
                return x + w * dinv * rr1 + W0 @ cyc(l + 1, p.T @ rr1)
            x = x + p @ cyc(l + 1, p.T @ rr1)
            return x + w * dinv * (bb - a @ x)
        return cyc(0, rr)
    x, it = AP.pcg(A, b, M)
    tag = kind + ("+W" if W0 is not None else "")
    print(f"{tag}: PCG iterations {it}, true relres {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.2e}", flush=True)
