"""SpMV on a 1M-node mesh: scan-line numbering vs shuffled (CGAL-like worst case) vs shuffled + Z-order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from padne_amd import _hip, synthetic, reduction
ctx = _hip.Context(0)
xy, tri = synthetic.jittered_grid(1000, 1000)
n = len(xy)
e = np.zeros(0, np.int64)
def spmv_us(xy_, tri_, perm_map=None):
    L = ctx.assemble_system(n, xy_, tri_, [0, n], [0, len(tri_)], [2082.5], e, e, np.zeros(0))
    A = L if perm_map is None else L.reduce(perm_map, n, 1.0)
    x = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, n)); y = ctx.empty(n)
    t = min(A.spmv_time(x, y, 5, 50) for _ in range(3))
    return t * 1e6, A.spmv_bytes / t / 1e9
print("scan-line numbering : %.1f us %.0f GB/s" % spmv_us(xy, tri))
rng = np.random.default_rng(0); perm = rng.permutation(n); inv = np.empty_like(perm); inv[perm] = np.arange(n)
sxy, stri = xy[perm], inv[tri].astype(np.int32)
print("shuffled numbering  : %.1f us %.0f GB/s" % spmv_us(sxy, stri))
rank = np.empty(n, dtype=np.int32); rank[np.argsort(reduction.morton_keys(sxy), kind="stable")] = np.arange(n, dtype=np.int32)
print("shuffled + Z-order  : %.1f us %.0f GB/s" % spmv_us(sxy, stri, rank))
strip = reduction.strip_index(sxy, np.zeros(n, dtype=np.int64))
srank = np.empty(n, dtype=np.int32); srank[np.lexsort((sxy[:, 0], strip))] = np.arange(n, dtype=np.int32)
print("shuffled + strips   : %.1f us %.0f GB/s" % spmv_us(sxy, stri, srank))
# an unstructured mesh (random Delaunay, 1 M points): Z-order vs strips
import scipy.spatial
pts = np.random.default_rng(3).uniform(0, 100, (1000000, 2))
dt = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
a, b, c = pts[dt[:, 0]], pts[dt[:, 1]], pts[dt[:, 2]]
cr = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
dt[cr < 0] = dt[cr < 0][:, [0, 2, 1]]
n = len(pts)
zr = np.empty(n, dtype=np.int32); zr[np.argsort(reduction.morton_keys(pts), kind="stable")] = np.arange(n, dtype=np.int32)
st = reduction.strip_index(pts, np.zeros(n, dtype=np.int64))
sr = np.empty(n, dtype=np.int32); sr[np.lexsort((pts[:, 0], st))] = np.arange(n, dtype=np.int32)
print("Delaunay, as generated: %.1f us %.0f GB/s" % spmv_us(pts, dt))
print("Delaunay + Z-order    : %.1f us %.0f GB/s" % spmv_us(pts, dt, zr))
print("Delaunay + strips     : %.1f us %.0f GB/s" % spmv_us(pts, dt, sr))
