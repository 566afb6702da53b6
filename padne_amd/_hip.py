"""ctypes binding of ``libpadne_hip.so`` (the C ABI declared in ``include/padne_hip.h``).

The product path has NO CPU fallback: if the shared library is missing, or no
GPU is visible, the first call raises ``HipUnavailableError``.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpadne_hip.so")

OK = 0
E_INVALID, E_HIP, E_NOMEM, E_NONMANIFOLD, E_NOTCONVERGED, E_COMM, E_BREAKDOWN, E_TOOLARGE, E_NOCOARSEN = \
    -1, -2, -3, -4, -5, -6, -7, -8, -9


class HipUnavailableError(RuntimeError):
    """libpadne_hip.so cannot be loaded or no MI355X is visible."""


class HipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libpadne_hip error {code}: {message}")
        self.code = code


class NotConvergedError(HipError):
    pass


class SolveOpts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("max_iter", C.c_int32),
                ("precond", C.c_int32), ("check_every", C.c_int32), ("flags", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("restarts", C.c_int32), ("rel_residual", C.c_double),
                ("abs_residual", C.c_double), ("solve_seconds", C.c_double), ("spmv_seconds", C.c_double),
                ("status", C.c_int32), ("n_rhs", C.c_int32), ("precond_setup_seconds", C.c_double),
                ("operator_complexity", C.c_double), ("levels", C.c_int32), ("precond_fallbacks", C.c_int32)]


_P = C.c_void_p
_I64 = C.c_int64
_PI32 = C.POINTER(C.c_int32)
_PI64 = C.POINTER(C.c_int64)
_PF64 = C.POINTER(C.c_double)

# name -> (restype, argtypes): every symbol include/padne_hip.h declares
SIGNATURES = {
    "padne_abi_version": (C.c_int, []),
    "padne_last_error": (C.c_char_p, []),
    "padne_device_count": (C.c_int, []),
    "padne_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "padne_ctx_destroy": (C.c_int, [_P]),
    "padne_ctx_synchronize": (C.c_int, [_P]),
    "padne_ctx_stream": (_P, [_P]),
    "padne_comm_unique_id": (C.c_int, [_P]),
    "padne_ctx_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "padne_ctx_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "padne_comm_call_counts": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "padne_launch_count": (C.c_int, [C.POINTER(C.c_longlong)]),
    "padne_ctx_comm_init_host": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "padne_ctx_p2p_export": (C.c_int, [_P, C.c_int32, _P]),
    "padne_ctx_p2p_import": (C.c_int, [_P, _P, C.c_int32]),
    "padne_ctx_p2p_selftest": (C.c_int, [_P, _PI32]),
    "padne_ctx_p2p_close": (C.c_int, [_P]),
    "padne_ctx_set_halo": (C.c_int, [_P, _I64, C.c_int32, C.c_int32, _PI32]),
    "padne_dev_alloc": (C.c_int, [_P, _I64, C.POINTER(_P)]),
    "padne_dev_free": (C.c_int, [_P, _P]),
    "padne_dev_upload": (C.c_int, [_P, _P, _P, _I64]),
    "padne_dev_download": (C.c_int, [_P, _P, _P, _I64]),
    "padne_dev_memset": (C.c_int, [_P, _P, C.c_int, _I64]),
    "padne_csr_from_host": (C.c_int, [_P, _I64, _I64, _PI32, _PI32, _PF64, C.POINTER(_P)]),
    "padne_csr_destroy": (C.c_int, [_P]),
    "padne_csr_shape": (C.c_int, [_P, _PI64, _PI64, _PI64]),
    "padne_csr_to_host": (C.c_int, [_P, _P, _PI32, _PI32, _PF64]),
    "padne_assemble_system": (C.c_int, [_P, _I64, _I64, _PF64, _I64, _PI32, _I64, _PI64, _PI64, _PF64,
                                        _I64, _PI64, _PI64, _PF64, C.POINTER(_P)]),
    "padne_assemble_system_ex": (C.c_int, [_P, _I64, _I64, _PF64, _I64, _PI32, _I64, _PI64, _PI64, _PF64,
                                           _I64, _PI64, _PI64, _PF64, C.c_int32, C.POINTER(_P)]),
    "padne_generate_grid_mesh": (C.c_int, [_P, _I64, _I64, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.POINTER(C.c_uint64), _P, _P]),
    "padne_csr_reduce": (C.c_int, [_P, _P, _PI32, _I64, C.c_double, C.POINTER(_P)]),
    "padne_csr_relabel": (C.c_int, [_P, _P, _PI32, _I64, _PI32, _I64, C.c_double, C.POINTER(_P)]),
    "padne_csr_vstack": (C.c_int, [_P, _P, _P, C.POINTER(_P)]),
    "padne_spmv": (C.c_int, [_P, _P, _PF64, _PF64]),
    "padne_spmv_dev": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "padne_spmm8_dev": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "padne_spmm8_algorithmic_bytes": (_I64, [_P]),
    "padne_spmm8_time": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _PF64]),
    "padne_residual_norm": (C.c_int, [_P, _P, _PF64, _PF64, _PF64]),
    "padne_solve_spd": (C.c_int, [_P, _P, _PF64, _PF64, C.c_int32, C.POINTER(SolveOpts), C.POINTER(SolveInfo)]),
    "padne_solve_spd_dev": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(SolveOpts), C.POINTER(SolveInfo)]),
    "padne_kkt_create": (C.c_int, [_P, _P, _I64, _I64, _PI64, _I64, _PI64, _PI64, _PI32, _I64, C.c_int32, C.POINTER(_P)]),
    "padne_kkt_destroy": (C.c_int, [_P]),
    "padne_kkt_matrix": (C.c_int, [_P, C.POINTER(_P)]),
    "padne_kkt_solve": (C.c_int, [_P, _P, _PF64, _I64, _PI64, _PF64, C.c_int32, _PI64, _PI64, _PF64, _I64, _PI64, _PF64,
                                  C.POINTER(SolveOpts), C.c_double, C.POINTER(SolveInfo)]),
    "padne_kkt_finish": (C.c_int, [_P, _P, C.c_int32, _PF64, _I64, _PI64, _PF64, _PF64, _PF64]),
    "padne_amg_apply": (C.c_int, [_P, _P, _PF64, _PF64]),
    "padne_csr_set_preconditioner_block": (C.c_int, [_P, _P]),
    "padne_amg_level": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "padne_nearest_vertex": (C.c_int, [_P, _I64, _PF64, _I64, _PF64, _PI64]),
    "padne_nearest_vertex_ties": (C.c_int, [_P, _I64, _PF64, _I64, _PF64, _PI64, _PI32]),
    "padne_power_density": (C.c_int, [_P, _I64, _PF64, _I64, _PI32, _I64, _PI64, _PI64, _PF64, _PF64, _PF64]),
    "padne_csr_power_density": (C.c_int, [_P, _P, _PF64, _PF64]),
    "padne_face_gradient": (C.c_int, [_P, _I64, _PF64, _I64, _PI32, _I64, _PI64, _PI64, _PF64, _PF64, _PF64]),
    "padne_spmv_algorithmic_bytes": (_I64, [_P]),
    "padne_spmv_time": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _PF64]),
}

# test-only entry points (include/padne_hip_test.h): the in-process team that rehearses several ranks on one GPU
TEST_SIGNATURES = {
    "padne_team_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "padne_team_destroy": (C.c_int, [_P]),
    "padne_team_abort": (C.c_int, [_P]),
    "padne_ctx_join_team": (C.c_int, [_P, _P, C.c_int]),
    "padne_csr_split_tiles": (C.c_int, [_P, C.c_int, _PI64, _PI64]),
    "padne_ctx_lockstep_groups": (C.c_int, [_P, _PI64]),
    "padne_asm_second_path_count": (C.c_int, [_PI64]),
    "padne_ctx_halo_exchange_time": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double)]),
    "padne_ctx_reload_options": (C.c_int, [_P]),
}

_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen the in-tree library and attach prototypes.  Does not touch the GPU."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipUnavailableError(
            f"{p} not found: build it with `python -m padne_amd.build` (hipcc --offload-arch=gfx950). "
            "padne_amd has no CPU fallback.")
    # When the process also uses torch (bench.py, torch.distributed ranks) let torch load its
    # HIP runtime first so that both share one libamdhip64.so.7 / librccl.so.1.
    if "torch" in sys.modules:
        pass
    try:
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise HipUnavailableError(f"cannot load {p}: {exc}") from exc
    for name, (res, args) in list(SIGNATURES.items()) + list(TEST_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.padne_abi_version() != 1:
        raise HipUnavailableError("libpadne_hip.so ABI version mismatch; rebuild it")
    if path is None:
        _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc == OK:
        return
    msg = (load_library().padne_last_error() or b"").decode("utf-8", "replace")
    if rc == E_NONMANIFOLD:
        raise ValueError(msg or "Non-manifold mesh")   # mesh.py:342-343
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    if rc == E_TOOLARGE:
        raise OverflowError(msg)
    if rc == E_NOTCONVERGED:
        raise NotConvergedError(rc, msg)
    raise HipError(rc, msg)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def device_count() -> int:
    n = load_library().padne_device_count()
    return max(n, 0)


def launch_count() -> int:
    """Kernels and asynchronous fills this process has queued through the library so far (all contexts)."""
    n = C.c_longlong(0)
    _check(load_library().padne_launch_count(C.byref(n)))
    return int(n.value)


def asm_second_path_count() -> int:
    """Assemblies of this process that took the two-pass second path of the row kernel (test introspection)."""
    n = C.c_int64(0)
    _check(load_library().padne_asm_second_path_count(C.byref(n)))
    return int(n.value)


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, _P, _P, _P, C.c_int64)      # padne_allgather_fn


_live_contexts = weakref.WeakSet()


def reload_options_everywhere() -> None:
    """Every live context reads the PADNE_* environment switches again (test scaffolding: the library reads them once,
    when a context is created)."""
    for c in list(_live_contexts):
        if getattr(c, "_h", None):
            c.reload_options()


class Context:
    """Device context: GPU, stream, workspaces, optional RCCL communicator."""

    def __init__(self, device: int = 0):
        lib = load_library()
        self._lib = lib
        h = _P()
        rc = lib.padne_ctx_create(int(device), C.byref(h))
        if rc != OK:
            msg = (lib.padne_last_error() or b"").decode()
            raise HipUnavailableError(f"cannot create a context on GPU {device}: {msg}")
        self._h = h
        self.device = int(device)
        self.halo_n_owned = None      # length of b / x when a halo plan is active
        _live_contexts.add(self)

    # -- lifetime ------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.padne_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        _check(self._lib.padne_ctx_synchronize(self._h))

    @property
    def stream(self) -> int:
        return int(self._lib.padne_ctx_stream(self._h) or 0)

    # -- communicator ---------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self._lib.padne_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(self._lib.padne_ctx_comm_init(self._h, buf, int(rank), int(world_size)))

    def comm_init_host(self, rank: int, world_size: int, allgather):
        """Collectives through a transport of the caller (gloo, MPI): ``allgather(send: np.ndarray[uint8]) -> bytes-like`` of
        ``world_size * len(send)`` bytes in rank order.  An exception in it fails the collective with PADNE_E_COMM."""
        world_size = int(world_size)

        def cb(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_ubyte)), shape=(int(nbytes),)).copy()
                out = np.frombuffer(allgather(src), dtype=np.uint8)
                if out.size != world_size * int(nbytes):
                    return 2
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except BaseException:      # noqa: BLE001 -- whatever the transport raises: the collective failed
                return 1
        self._host_allgather = ALLGATHER_FN(cb)             # (kept alive with the context)
        _check(self._lib.padne_ctx_comm_init_host(self._h, int(rank), world_size, C.cast(self._host_allgather, _P), None))

    def p2p_export(self, slots_per_rank: int) -> bytes:
        """This rank's mailbox of the peer-to-peer halo exchange between processes: allocate, return the 64-byte hipIpc handle."""
        buf = C.create_string_buffer(64)
        _check(self._lib.padne_ctx_p2p_export(self._h, int(slots_per_rank), buf))
        return buf.raw

    def p2p_import(self, handles: bytes, world_size: int) -> None:
        buf = C.create_string_buffer(bytes(handles), 64 * int(world_size))
        _check(self._lib.padne_ctx_p2p_import(self._h, buf, int(world_size)))

    def p2p_selftest(self) -> bool:
        """One real exchange of known values through the shared mailboxes (collective): did every rank's stores arrive here?"""
        ok = C.c_int32(0)
        _check(self._lib.padne_ctx_p2p_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def p2p_close(self) -> None:
        _check(self._lib.padne_ctx_p2p_close(self._h))

    def comm_call_counts(self):
        """(calls, bytes) of the communication issued so far: all-reduce, all-gather f64, all-gather f32 (the collectives)
        and peer-to-peer halo exchanges."""
        calls = (C.c_longlong * 4)()
        nbytes = (C.c_longlong * 4)()
        _check(self._lib.padne_comm_call_counts(calls, nbytes))
        return list(calls), list(nbytes)

    def reload_options(self) -> None:
        """Read the PADNE_* environment switches again (they are read once, at creation; test scaffolding)."""
        _check(self._lib.padne_ctx_reload_options(self._h))

    def halo_exchange_time(self, repeats: int = 200) -> float:
        """Average device seconds of one halo exchange of this context's plan (collective; test introspection)."""
        t = C.c_double(0.0)
        _check(self._lib.padne_ctx_halo_exchange_time(self._h, int(repeats), C.byref(t)))
        return float(t.value)

    def lockstep_groups(self) -> int:
        """Groups of right-hand sides this context has advanced in lockstep so far (test introspection)."""
        g = C.c_int64(0)
        _check(self._lib.padne_ctx_lockstep_groups(self._h, C.byref(g)))
        return int(g.value)

    def set_halo(self, n_owned: int, m: int, export_idx) -> None:
        e = _i32(export_idx)
        _check(self._lib.padne_ctx_set_halo(self._h, int(n_owned), int(m), int(e.shape[0]), _ptr(e, _PI32)))
        self.halo_n_owned = int(n_owned)

    def clear_halo(self) -> None:
        _check(self._lib.padne_ctx_set_halo(self._h, -1, 0, 0, None))
        self.halo_n_owned = None

    # -- raw device memory ------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = _P()
        _check(self._lib.padne_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return int(p.value)

    def free(self, dev: int):
        _check(self._lib.padne_dev_free(self._h, _P(dev)))

    def upload(self, dev: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        _check(self._lib.padne_dev_upload(self._h, _P(dev), host.ctypes.data_as(_P), host.nbytes))

    def download(self, dev: int, shape, dtype=np.float64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _check(self._lib.padne_dev_download(self._h, out.ctypes.data_as(_P), _P(dev), out.nbytes))
        return out

    def to_device(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host)
        d = DeviceArray(self, host.shape, host.dtype)
        self.upload(d.ptr, host)
        return d

    def empty(self, shape, dtype=np.float64) -> "DeviceArray":
        return DeviceArray(self, shape, dtype)

    # -- matrices -------------------------------------------------------------------
    def csr_from_scipy(self, A) -> "CsrMatrix":
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        A.sum_duplicates()
        A.sort_indices()
        if A.nnz >= 2**31 - 8192:
            raise ValueError("matrix too large for int32 indices")
        indptr, indices, data = _i32(A.indptr), _i32(A.indices), _f64(A.data)
        h = _P()
        _check(self._lib.padne_csr_from_host(self._h, A.shape[0], A.shape[1], _ptr(indptr, _PI32),
                                             _ptr(indices, _PI32), _ptr(data, _PF64), C.byref(h)))
        return CsrMatrix(self, h)

    def assemble_system(self, n_unknowns, xy, tri, mesh_vertex_offset, mesh_tri_offset, conductance,
                        coo_row, coo_col, coo_val, partial_mesh: bool = False) -> "CsrMatrix":
        """L in the reference layout: cotangent Laplacians of all meshes + lumped stamps.  ``partial_mesh``: the
        triangles are one rank's piece of a partitioned mesh (no manifold test, see padne_assemble_system_ex)."""
        # xy / tri: host arrays, or DeviceArrays (e.g. filled by generate_grid_mesh): then nothing crosses PCIe
        xy_dev = isinstance(xy, DeviceArray)
        tri_dev = isinstance(tri, DeviceArray)
        if xy_dev != tri_dev:
            raise ValueError("xy and tri must both be host arrays or both DeviceArrays")
        if not xy_dev:
            xy = _f64(xy).reshape(-1, 2)
            tri = _i32(tri).reshape(-1, 3)
        elif xy.dtype != np.float64 or tri.dtype != np.int32:
            raise ValueError("device xy must be float64 and device tri int32")
        n_xy = int(np.prod(xy.shape)) // 2
        n_tr = int(np.prod(tri.shape)) // 3
        p_xy = C.cast(_P(xy.ptr), _PF64) if xy_dev else _ptr(xy, _PF64)
        p_tri = C.cast(_P(tri.ptr), _PI32) if tri_dev else _ptr(tri, _PI32)
        mvo, mto, sig = _i64(mesh_vertex_offset), _i64(mesh_tri_offset), _f64(conductance)
        cr, cc, cv = _i64(coo_row), _i64(coo_col), _f64(coo_val)
        n_mesh = sig.shape[0]
        if mvo.shape[0] != n_mesh + 1 or mto.shape[0] != n_mesh + 1:
            raise ValueError("offset tables must have n_mesh+1 entries")
        if not (cr.shape == cc.shape == cv.shape):
            raise ValueError("coo arrays must have equal length")
        h = _P()
        _check(self._lib.padne_assemble_system_ex(
            self._h, int(n_unknowns), n_xy, p_xy, n_tr, p_tri, n_mesh,
            _ptr(mvo, _PI64), _ptr(mto, _PI64), _ptr(sig, _PF64), cr.shape[0], _ptr(cr, _PI64),
            _ptr(cc, _PI64), _ptr(cv, _PF64), 1 if partial_mesh else 0, C.byref(h)))
        return CsrMatrix(self, h)

    def generate_grid_mesh(self, nx: int, ny: int, h: float, seed: int = 0, jitter: float = 0.2, origin=(0.0, 0.0),
                           xy_out: "DeviceArray | None" = None, tri_out: "DeviceArray | None" = None,
                           vertex_offset: int = 0, tri_offset: int = 0):
        """``synthetic.jittered_grid(nx, ny, h, seed, jitter, origin)`` generated on the device, bit for bit (the jitter is
        numpy's ``default_rng(seed)`` stream, evaluated per vertex by jump-ahead).  Returns (xy, tri) DeviceArrays; with
        ``xy_out`` / ``tri_out`` the mesh is written at ``vertex_offset`` / ``tri_offset`` of existing arrays."""
        n_v, n_t = nx * ny, 2 * (nx - 1) * (ny - 1)
        xy = xy_out if xy_out is not None else DeviceArray(self, (n_v, 2), np.float64)
        tri = tri_out if tri_out is not None else DeviceArray(self, (n_t, 3), np.int32)
        st = np.random.default_rng(seed).bit_generator.state["state"]
        g = (C.c_uint64 * 4)(st["state"] >> 64, st["state"] & (2 ** 64 - 1), st["inc"] >> 64, st["inc"] & (2 ** 64 - 1))
        _check(self._lib.padne_generate_grid_mesh(self._h, int(nx), int(ny), float(h), float(jitter), float(origin[0]),
                                                  float(origin[1]), g, _P(xy.ptr + 16 * int(vertex_offset)),
                                                  _P(tri.ptr + 12 * int(tri_offset))))
        return xy, tri

    def nearest_vertex(self, points: np.ndarray, queries: np.ndarray, with_ties: bool = False):
        """Index of the nearest of ``points`` (n, 2) for every row of ``queries`` (m, 2); ties to the smallest index.
        ``with_ties``: also the number of points at exactly the minimum distance, per query."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 2)
        out = np.empty(len(q), dtype=np.int64)
        ties = np.ones(len(q), dtype=np.int32)
        if len(q):
            _check(self._lib.padne_nearest_vertex_ties(self._h, len(pts), _ptr(pts, _PF64), len(q), _ptr(q, _PF64),
                                                       _ptr(out, _PI64), _ptr(ties, _PI32)))
        return (out, ties) if with_ties else out

    def power_density(self, xy, tri, mesh_vertex_offset, mesh_tri_offset, conductance, potential) -> np.ndarray:
        # xy / tri: host arrays, or DeviceArrays (e.g. filled by generate_grid_mesh): then nothing crosses PCIe
        xy_dev = isinstance(xy, DeviceArray)
        tri_dev = isinstance(tri, DeviceArray)
        if xy_dev != tri_dev:
            raise ValueError("xy and tri must both be host arrays or both DeviceArrays")
        if not xy_dev:
            xy = _f64(xy).reshape(-1, 2)
            tri = _i32(tri).reshape(-1, 3)
        elif xy.dtype != np.float64 or tri.dtype != np.int32:
            raise ValueError("device xy must be float64 and device tri int32")
        n_xy = int(np.prod(xy.shape)) // 2
        n_tr = int(np.prod(tri.shape)) // 3
        p_xy = C.cast(_P(xy.ptr), _PF64) if xy_dev else _ptr(xy, _PF64)
        p_tri = C.cast(_P(tri.ptr), _PI32) if tri_dev else _ptr(tri, _PI32)
        mvo, mto, sig = _i64(mesh_vertex_offset), _i64(mesh_tri_offset), _f64(conductance)
        pot = _f64(potential)
        if pot.shape[0] < xy.shape[0]:
            raise ValueError("potential vector shorter than the vertex count")
        out = np.zeros(tri.shape[0], dtype=np.float64)
        _check(self._lib.padne_power_density(self._h, xy.shape[0], _ptr(xy, _PF64), tri.shape[0],
                                             _ptr(tri, _PI32), sig.shape[0], _ptr(mvo, _PI64), _ptr(mto, _PI64),
                                             _ptr(sig, _PF64), _ptr(pot, _PF64), _ptr(out, _PF64)))
        return out


    def face_gradient(self, xy, tri, mesh_vertex_offset, mesh_tri_offset, potential):
        xy = _f64(xy).reshape(-1, 2)
        tri = _i32(tri).reshape(-1, 3)
        mvo, mto, pot = _i64(mesh_vertex_offset), _i64(mesh_tri_offset), _f64(potential)
        gx = np.zeros(tri.shape[0], dtype=np.float64)
        gy = np.zeros(tri.shape[0], dtype=np.float64)
        _check(self._lib.padne_face_gradient(self._h, xy.shape[0], _ptr(xy, _PF64), tri.shape[0], _ptr(tri, _PI32),
                                             mvo.shape[0] - 1, _ptr(mvo, _PI64), _ptr(mto, _PI64), _ptr(pot, _PF64),
                                             _ptr(gx, _PF64), _ptr(gy, _PF64)))
        return gx, gy


class LocalTeam:
    """In-process team of contexts acting as ranks on one GPU (rehearsal of the multi-rank path)."""

    def __init__(self, world_size: int):
        lib = load_library()
        self._lib = lib
        h = _P()
        _check(lib.padne_team_create(int(world_size), C.byref(h)))
        self._h = h
        self.world = int(world_size)

    def join(self, ctx: "Context", rank: int) -> None:
        _check(self._lib.padne_ctx_join_team(ctx._h, self._h, int(rank)))
        ctx._team = self       # keep the team alive as long as its members

    def abort(self) -> None:
        """Wake every rank that waits in a collective; all team collectives return E_COMM from now on."""
        if getattr(self, "_h", None):
            self._lib.padne_team_abort(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.padne_team_destroy(self._h)
            self._h = None


class DeviceArray:
    """A flat device allocation with numpy-like shape/dtype metadata."""

    def __init__(self, ctx: Context, shape, dtype=np.float64):
        self.ctx = ctx
        self.shape = tuple(np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.alloc(max(self.nbytes, 8))

    def numpy(self) -> np.ndarray:
        return self.ctx.download(self.ptr, self.shape, self.dtype)

    def set(self, host: np.ndarray):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        self.ctx.upload(self.ptr, host)

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if self.ptr and self.ctx._h:
                self.free()
        except Exception:
            pass


@dataclass
class SolveResult:
    x: np.ndarray | None
    iterations: int
    restarts: int
    rel_residual: float
    abs_residual: float
    seconds: float
    status: int
    spmv_seconds: float = 0.0
    setup_seconds: float = 0.0
    operator_complexity: float = 0.0
    levels: int = 0
    precond_fallbacks: int = 0      # right-hand sides redone with the Jacobi preconditioner after a multigrid failure


class KktPlan:
    """``padne_kkt``: the reduction of one assembled KKT system to its SPD core, resident on the device (index map,
    reduced matrix with its multigrid hierarchy, the N-vectors).  ``solve`` + ``finish`` are the two device stages of
    ``solver.solve_system``; the multiplier recovery between them is O(#constraints) host work."""

    def __init__(self, L: "CsrMatrix", n_potential: int, elim, tied, n_free: int, index_map=None, strip_order: bool = False):
        self.ctx, self.L = L.ctx, L
        elim = _i64(elim)
        tm = _i64([m for m, _ in tied])
        tr = _i64([r for _, r in tied])
        imap = None if index_map is None else _i32(index_map)
        if imap is not None and imap.shape[0] != L.shape[0]:
            raise ValueError("index map length must equal the matrix dimension")
        h = _P()
        _check(self.ctx._lib.padne_kkt_create(self.ctx._h, L._h, int(n_potential), elim.shape[0], _ptr(elim, _PI64), tm.shape[0],
                                              _ptr(tm, _PI64), _ptr(tr, _PI64), None if imap is None else _ptr(imap, _PI32),
                                              int(n_free), 1 if strip_order else 0, C.byref(h)))
        self._h = h
        self.n_free = int(n_free)
        self.N = L.shape[0]

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx._lib.padne_kkt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reduced_matrix(self) -> "CsrMatrix":
        """Borrowed view of A = -P^T L P (valid while the plan lives)."""
        h = _P()
        _check(self.ctx._lib.padne_kkt_matrix(self._h, C.byref(h)))
        m = CsrMatrix(self.ctx, h)
        m._borrowed = True
        return m

    def solve(self, r, known: dict, extras: list, probes, *, rtol=1e-12, max_iter=200000, precond="amg",
              abs_residual_target=0.0, rebuild=False):
        """Stage 1.  ``known`` {unknown: c}; ``extras``: list of {row: value} columns; ``probes``: unknowns whose residual
        rows come back.  Returns (probe values [(1 + len(extras)), len(probes)], SolveResult)."""
        r = _f64(r)
        if r.shape[0] != self.N:
            raise ValueError("right-hand side has the wrong length")
        kidx = _i64(sorted(known))
        kval = _f64([known[int(i)] for i in kidx])
        ptr, rows, vals = [0], [], []
        for col in extras:
            for row, val in col.items():
                rows.append(int(row))
                vals.append(float(val))
            ptr.append(len(rows))
        ptr, rows, vals = _i64(ptr), _i64(rows), _f64(vals)
        pidx = _i64(list(probes))
        out = np.zeros((1 + len(extras), max(len(pidx), 1)), dtype=np.float64)
        opts = CsrMatrix._opts(rtol, 0.0, max_iter, 0, False, precond=precond, rebuild=rebuild)
        info = SolveInfo()
        # the vector stage 2 will hand back: its pages are touched while the device solves (a fresh 80 MB array at 10 M
        # unknowns is 20 000 page faults in the path of the copy that brings v home)
        self._v_next, self._v_toucher = None, None
        if self.N >= (1 << 18):
            import threading
            v_next = np.empty(self.N, dtype=np.float64)
            self._v_next = v_next
            # (the thread holds the array itself, not just its address: it outlives a plan that is dropped on an error path)
            self._v_toucher = threading.Thread(target=lambda a=v_next: C.memset(a.ctypes.data, 0, a.nbytes), daemon=True)
            self._v_toucher.start()
        rc = self.ctx._lib.padne_kkt_solve(self.ctx._h, self._h, _ptr(r, _PF64), kidx.shape[0], _ptr(kidx, _PI64),
                                           _ptr(kval, _PF64), len(extras), _ptr(ptr, _PI64), _ptr(rows, _PI64),
                                           _ptr(vals, _PF64), pidx.shape[0], _ptr(pidx, _PI64), _ptr(out, _PF64),
                                           C.byref(opts), float(abs_residual_target), C.byref(info))
        if rc != OK and rc != E_NOTCONVERGED:
            _check(rc)
        res = SolveResult(None, info.iterations, info.restarts, info.rel_residual, info.abs_residual, info.solve_seconds,
                          info.status, info.spmv_seconds, info.precond_setup_seconds, info.operator_complexity, info.levels,
                          info.precond_fallbacks)
        return out[:, :len(pidx)], res

    def finish(self, extra_coeff, multipliers: dict):
        """Stage 2: (v, ||L v - r||)."""
        coeff = _f64(extra_coeff)
        midx = _i64(sorted(multipliers))
        mval = _f64([multipliers[int(i)] for i in midx])
        toucher, v = getattr(self, "_v_toucher", None), getattr(self, "_v_next", None)
        self._v_next, self._v_toucher = None, None
        if toucher is not None:
            toucher.join()
        if v is None:
            v = np.empty(self.N, dtype=np.float64)
        norm = C.c_double()
        _check(self.ctx._lib.padne_kkt_finish(self.ctx._h, self._h, coeff.shape[0], _ptr(coeff, _PF64), midx.shape[0],
                                              _ptr(midx, _PI64), _ptr(mval, _PF64), _ptr(v, _PF64), C.byref(norm)))
        return v, norm.value


class CsrMatrix:
    """Device-resident CSR matrix (opaque handle of the C ABI)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle
        nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        _check(ctx._lib.padne_csr_shape(handle, C.byref(nr), C.byref(nc), C.byref(nnz)))
        self.shape = (nr.value, nc.value)
        self.nnz = nnz.value

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h and not getattr(self, "_borrowed", False):
            self.ctx._lib.padne_csr_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def spmv_bytes(self) -> int:
        return int(self.ctx._lib.padne_spmv_algorithmic_bytes(self._h))

    def to_scipy(self):
        import scipy.sparse as sp
        indptr = np.empty(self.shape[0] + 1, dtype=np.int32)
        indices = np.empty(self.nnz, dtype=np.int32)
        data = np.empty(self.nnz, dtype=np.float64)
        _check(self.ctx._lib.padne_csr_to_host(self.ctx._h, self._h, _ptr(indptr, _PI32), _ptr(indices, _PI32),
                                               _ptr(data, _PF64)))
        return sp.csr_matrix((data, indices, indptr), shape=self.shape)

    def reduce(self, index_map, n_out: int, scale: float = 1.0) -> "CsrMatrix":
        """``scale * P^T M P``.  ``index_map``: a host array, or a ``DeviceArray`` of int32 already on this GPU (the
        library reads a device-resident map where it lies: no upload per call)."""
        if isinstance(index_map, DeviceArray):
            if index_map.dtype != np.int32 or int(np.prod(index_map.shape)) != self.shape[0]:
                raise ValueError("index map must be int32 of the matrix dimension")
            mp = C.cast(_P(index_map.ptr), _PI32)
        else:
            m = _i32(index_map)
            if m.shape[0] != self.shape[0]:
                raise ValueError("index map length must equal the matrix dimension")
            mp = _ptr(m, _PI32)
        h = _P()
        _check(self.ctx._lib.padne_csr_reduce(self.ctx._h, self._h, mp, int(n_out), float(scale), C.byref(h)))
        return CsrMatrix(self.ctx, h)

    def relabel(self, row_map: np.ndarray, n_rows_out: int, col_map: np.ndarray, n_cols_out: int,
                scale: float = 1.0) -> "CsrMatrix":
        """``scale * R^T M C`` with separate row / column index maps (-1 drops the row / column)."""
        rm = np.ascontiguousarray(row_map, dtype=np.int32)
        cm = np.ascontiguousarray(col_map, dtype=np.int32)
        if rm.shape[0] != self.shape[0] or cm.shape[0] != self.shape[1]:
            raise ValueError("index map lengths must equal the matrix dimensions")
        h = _P()
        _check(self.ctx._lib.padne_csr_relabel(self.ctx._h, self._h, _ptr(rm, _PI32), int(n_rows_out), _ptr(cm, _PI32),
                                               int(n_cols_out), float(scale), C.byref(h)))
        return CsrMatrix(self.ctx, h)

    def power_density(self, potential: np.ndarray, n_tri: int) -> np.ndarray:
        """Per-triangle power density on the mesh this system was assembled from (kept on the device with it)."""
        pot = _f64(potential)
        out = np.empty(int(n_tri), dtype=np.float64)
        _check(self.ctx._lib.padne_csr_power_density(self.ctx._h, self._h, _ptr(pot, _PF64), _ptr(out, _PF64)))
        return out

    def vstack(self, bottom: "CsrMatrix") -> "CsrMatrix":
        h = _P()
        _check(self.ctx._lib.padne_csr_vstack(self.ctx._h, self._h, bottom._h, C.byref(h)))
        return CsrMatrix(self.ctx, h)

    def matvec(self, x) -> np.ndarray:
        x = _f64(x)
        if x.shape[0] != self.shape[1]:
            raise ValueError("dimension mismatch")
        y = np.empty(self.shape[0], dtype=np.float64)
        _check(self.ctx._lib.padne_spmv(self.ctx._h, self._h, _ptr(x, _PF64), _ptr(y, _PF64)))
        return y

    def matvec_dev(self, x: DeviceArray, y: DeviceArray, repeat: int = 1):
        _check(self.ctx._lib.padne_spmv_dev(self.ctx._h, self._h, _P(x.ptr), _P(y.ptr), int(repeat)))

    def spmv_time(self, x: DeviceArray, y: DeviceArray, warmup: int = 5, repeat: int = 50) -> float:
        t = C.c_double()
        _check(self.ctx._lib.padne_spmv_time(self.ctx._h, self._h, _P(x.ptr), _P(y.ptr), warmup, repeat,
                                             C.byref(t)))
        return t.value

    def matmat8_dev(self, x: DeviceArray, y: DeviceArray, repeat: int = 1):
        """Y = M X for 8 interleaved vectors: x holds shape[1]*8 doubles laid out [i][j], y shape[0]*8."""
        if x.nbytes != self.shape[1] * 64 or y.nbytes != self.shape[0] * 64:
            raise ValueError("dimension mismatch")
        _check(self.ctx._lib.padne_spmm8_dev(self.ctx._h, self._h, _P(x.ptr), _P(y.ptr), int(repeat)))

    def matmat8(self, X) -> np.ndarray:
        """Host convenience: X (n_cols, 8) -> M @ X (n_rows, 8)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.shape != (self.shape[1], 8):
            raise ValueError("X must have shape (n_cols, 8)")
        xd = self.ctx.to_device(X.reshape(-1))
        yd = self.ctx.empty(self.shape[0] * 8)
        self.matmat8_dev(xd, yd)
        return yd.numpy().reshape(self.shape[0], 8)

    def spmm8_time(self, x: DeviceArray, y: DeviceArray, warmup: int = 5, repeat: int = 50) -> float:
        t = C.c_double()
        _check(self.ctx._lib.padne_spmm8_time(self.ctx._h, self._h, _P(x.ptr), _P(y.ptr), warmup, repeat,
                                              C.byref(t)))
        return t.value

    @property
    def spmm8_bytes(self) -> int:
        return int(self.ctx._lib.padne_spmm8_algorithmic_bytes(self._h))

    def residual_norm(self, x, b) -> float:
        x, b = _f64(x), _f64(b)
        out = C.c_double()
        _check(self.ctx._lib.padne_residual_norm(self.ctx._h, self._h, _ptr(x, _PF64), _ptr(b, _PF64),
                                                 C.byref(out)))
        return out.value

    def set_preconditioner_block(self, block: "CsrMatrix | None") -> None:
        """Row-partitioned runs: multigrid is built on this owned x owned diagonal block."""
        _check(self.ctx._lib.padne_csr_set_preconditioner_block(self._h, block._h if block is not None else None))
        self._prec_block = block        # keep it alive

    def amg_level(self, level: int, which: str = "A"):
        """scipy copy of a hierarchy operator: which in 'A', 'P', 'R'."""
        import scipy.sparse as sp
        h = _P()
        _check(self.ctx._lib.padne_amg_level(self.ctx._h, self._h, int(level), {"A": 0, "P": 1, "R": 2}[which], C.byref(h)))
        nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self.ctx._lib.padne_csr_shape(h, C.byref(nr), C.byref(nc), C.byref(nnz)))
        indptr = np.empty(nr.value + 1, dtype=np.int32)
        indices = np.empty(nnz.value, dtype=np.int32)
        data = np.empty(nnz.value, dtype=np.float64)
        _check(self.ctx._lib.padne_csr_to_host(self.ctx._h, h, _ptr(indptr, _PI32), _ptr(indices, _PI32), _ptr(data, _PF64)))
        return sp.csr_matrix((data, indices, indptr), shape=(nr.value, nc.value))

    def amg_shapes(self):
        """(rows, cols, nnz) of every operator of the cached hierarchy, level by level: [{"A": .., "P": .., "R": ..}, ...]
        (the last level has only "A").  Shapes only, nothing is downloaded."""
        out = []
        for level in range(32):
            entry = {}
            for which, code in (("A", 0), ("P", 1), ("R", 2)):
                h = _P()
                if self.ctx._lib.padne_amg_level(self.ctx._h, self._h, level, code, C.byref(h)) != OK:
                    continue
                nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
                _check(self.ctx._lib.padne_csr_shape(h, C.byref(nr), C.byref(nc), C.byref(nnz)))
                entry[which] = (nr.value, nc.value, nnz.value)
            if "A" not in entry:
                break
            out.append(entry)
        return out

    def split_tiles(self, level: int = -1):
        """(interior, boundary) 64-row tiles of the split plan of this row-partitioned operator (level >= 0: of the
        level operator of its hierarchy); (0, 0) without one.  Test-only introspection."""
        a, b = C.c_int64(), C.c_int64()
        _check(self.ctx._lib.padne_csr_split_tiles(self._h, int(level), C.byref(a), C.byref(b)))
        return a.value, b.value

    def amg_apply(self, r) -> np.ndarray:
        """z = M^-1 r: one multigrid V-cycle (the preconditioner of solve_spd)."""
        r = _f64(r)
        if r.shape[0] != self.shape[0]:
            raise ValueError("dimension mismatch")
        z = np.empty_like(r)
        _check(self.ctx._lib.padne_amg_apply(self.ctx._h, self._h, _ptr(r, _PF64), _ptr(z, _PF64)))
        return z

    @staticmethod
    def _opts(rtol, atol, max_iter, check_every, guess, time_spmv=False, precond="jacobi", rebuild=False) -> SolveOpts:
        pc = {"jacobi": 0, "amg": 1, 0: 0, 1: 1}[precond]
        return SolveOpts(float(rtol), float(atol), int(max_iter), pc, int(check_every),
                         (1 if guess else 0) | (2 if time_spmv else 0) | (4 if rebuild else 0))

    def solve_spd(self, b, *, rtol=1e-12, atol=0.0, max_iter=200000, check_every=0, x0=None,
                  raise_on_fail=True, precond="amg", rebuild=False) -> SolveResult:
        """Jacobi-PCG on the device; b is host f64[n] or f64[k, n]."""
        b = _f64(b)
        n = self.shape[0] if self.ctx.halo_n_owned is None else self.ctx.halo_n_owned
        k = 1 if b.ndim == 1 else b.shape[0]
        if b.shape[-1] != n:
            raise ValueError("right-hand side has the wrong length")
        x = np.empty_like(b) if x0 is None else _f64(x0).copy()      # (no guess: the device starts from zero and writes all of x)
        opts = self._opts(rtol, atol, max_iter, check_every, x0 is not None, precond=precond, rebuild=rebuild)
        info = SolveInfo()
        rc = self.ctx._lib.padne_solve_spd(self.ctx._h, self._h, _ptr(b, _PF64), _ptr(x, _PF64), k,
                                           C.byref(opts), C.byref(info))
        if rc != OK and (raise_on_fail or rc != E_NOTCONVERGED):
            _check(rc)
        return SolveResult(x, info.iterations, info.restarts, info.rel_residual, info.abs_residual,
                           info.solve_seconds, info.status, info.spmv_seconds, info.precond_setup_seconds,
                           info.operator_complexity, info.levels, info.precond_fallbacks)

    def solve_spd_dev(self, b: DeviceArray, x: DeviceArray, *, n_rhs=1, rtol=1e-12, atol=0.0, max_iter=200000,
                      check_every=0, guess=False, raise_on_fail=True, time_spmv=False, precond="amg",
                      rebuild=False) -> SolveResult:
        opts = self._opts(rtol, atol, max_iter, check_every, guess, time_spmv, precond, rebuild)
        info = SolveInfo()
        rc = self.ctx._lib.padne_solve_spd_dev(self.ctx._h, self._h, _P(b.ptr), _P(x.ptr), int(n_rhs),
                                               C.byref(opts), C.byref(info))
        if rc != OK and (raise_on_fail or rc != E_NOTCONVERGED):
            _check(rc)
        return SolveResult(None, info.iterations, info.restarts, info.rel_residual, info.abs_residual,
                           info.solve_seconds, info.status, info.spmv_seconds, info.precond_setup_seconds,
                           info.operator_complexity, info.levels, info.precond_fallbacks)
