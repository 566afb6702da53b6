"""strip_hull_slivers of scripts/exp_unstructured.py as an importable helper (no side effects)."""
import numpy as np


def strip_hull_slivers(p, t, min_deg=12.0):
    """Drop sliver triangles that sit on the boundary (the hull of jittered grid points carries triangles with angles
    of 0 / 180 degrees; a quality mesher never emits those), repeatedly, interior triangles stay."""
    def min_angle(t):
        out = []
        for k in range(3):
            u = p[t[:, (k + 1) % 3]] - p[t[:, k]]
            v = p[t[:, (k + 2) % 3]] - p[t[:, k]]
            out.append(np.degrees(np.arccos(np.clip((u * v).sum(1) / np.linalg.norm(u, axis=1) / np.linalg.norm(v, axis=1), -1, 1))))
        return np.stack(out, 1).min(1)
    for _ in range(20):
        e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1).astype(np.int64)
        code = e[:, 0] * (len(p) + 1) + e[:, 1]
        _, inv, cnt = np.unique(code, return_inverse=True, return_counts=True)
        on_boundary = (cnt[inv] == 1).reshape(3, -1).any(0)
        bad = on_boundary & (min_angle(t) < min_deg)
        if not bad.any():
            break
        t = t[~bad]
    return t
