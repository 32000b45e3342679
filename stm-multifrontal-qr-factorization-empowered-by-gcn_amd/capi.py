"""ctypes binding of libstmmqr_hip.so (include/stmmqr_hip.h).

Mirrors the reference's operator interface for the hot path (STMMQR/include/SparseQR.h:127-268):
``qr_factorize``, ``qr_front``, ``qr_larftb``, ``qr_cpack``, ``qr_rhpack``, ``qr_assemble`` keep their names and
argument meaning; array arguments are numpy arrays (int64 = the reference's Long, float64).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
lib_path = _HERE / "libstmmqr_hip.so"

c_long_p = C.POINTER(C.c_long)
c_double_p = C.POINTER(C.c_double)
I64 = np.int64


class StmmqrError(RuntimeError):
    code = 0


ERR_RESCHEDULE = -6                        # STMMQR_ERR_RESCHEDULE (include/stmmqr_hip.h)


if not lib_path.exists():
    raise ImportError(
        f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (torch/lib, rpath $ORIGIN).  A
    process that loads /opt/rocm's runtime through this library FIRST and imports torch afterwards ends up with two runtimes:
    torch then reports "No HIP GPUs are available", and device pointers could not be handed from one to the other
    (sharded.py sends contribution blocks and panels from torch tensors).  So the runtime torch will use is loaded first, by
    path and without importing torch; the library's NEEDED libamdhip64.so.7 then binds to it (same soname).  Processes that
    never see torch (the C drivers, the relinked reference) use /opt/rocm's.  STMMQR_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import os
    if os.environ.get("STMMQR_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if rt.exists():
        C.CDLL(str(rt), mode=C.RTLD_GLOBAL)


_preload_torch_hip_runtime()
lib = C.CDLL(str(lib_path))


class SymbolicView(C.Structure):
    _fields_ = [(k, C.c_long) for k in ["m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "do_rank_detection"]] + \
               [(k, c_long_p) for k in ["Sp", "Sj", "Qfill", "PLinv", "Sleft", "Child", "Childp", "Super", "Rp", "Rj",
                                        "Post", "Hip", "Fm"]] + [("maxstack", C.c_long)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ["flops", "ms_total", "ms_assemble", "ms_front", "ms_pack", "ms_h2d", "ms_d2h",
                                          "ms_host", "bytes_assemble", "bytes_pack", "flops_update", "ms_update"]] + \
               [("nlaunch", C.c_long), ("nlevels", C.c_long), ("ms_panel", C.c_double), ("ms_small", C.c_double),
                ("npanel_launch", C.c_long), ("nupdate_launch", C.c_long), ("nsteps", C.c_long),
                ("flops_update_pair", C.c_double), ("retries", C.c_long), ("device_bytes", C.c_double), ("reschedules", C.c_long)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Options(C.Structure):
    _fields_ = [("panel_width", C.c_int), ("big_front_cols", C.c_int), ("verbose", C.c_int), ("use_graph", C.c_int),
                ("panel_algo", C.c_int), ("split_update", C.c_int), ("tall_min_rows", C.c_int),
                ("lookahead", C.c_int), ("fused_update", C.c_int), ("pair_update", C.c_int)]


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_long_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


lib.stmmqr_last_error.restype = C.c_char_p
lib.stmmqr_version.restype = C.c_char_p
lib.stmmqr_device_name.restype = C.c_char_p
lib.stmmqr_device_name.argtypes = [C.c_int]
lib.stmmqr_plan_create.restype = C.c_void_p
lib.stmmqr_plan_create.argtypes = [C.POINTER(SymbolicView), C.c_int, C.POINTER(C.c_int)]
lib.stmmqr_plan_destroy.argtypes = [C.c_void_p]
lib.stmmqr_plan_destroy.restype = None
lib.stmmqr_plan_set_pattern.argtypes = [C.c_void_p, c_long_p, c_long_p]
lib.stmmqr_factorize_device.argtypes = [C.c_void_p, c_long_p, c_long_p, C.c_void_p, C.c_int, C.c_double, C.c_long,
                                        C.POINTER(Stats)]
lib.stmmqr_plan_result_sizes.argtypes = [C.c_void_p, c_long_p, c_long_p]
lib.stmmqr_plan_download.argtypes = [C.c_void_p, c_double_p, c_long_p, C.c_char_p, c_long_p, c_double_p, c_long_p,
                                     c_long_p, c_long_p, c_long_p, c_long_p, C.POINTER(Stats)]
lib.stmmqr_front.restype = C.c_long
lib.stmmqr_front.argtypes = [C.c_long, C.c_long, C.c_long, C.c_double, C.c_long, c_double_p, c_long_p, C.c_char_p,
                             c_double_p, c_double_p]
lib.stmmqr_larftb_qtx.argtypes = [C.c_long] * 5 + [c_double_p, c_double_p, c_double_p]
lib.stmmqr_larftb.argtypes = [C.c_int] + [C.c_long] * 5 + [c_double_p, c_double_p, c_double_p]
lib.qr_larftb.restype = None
lib.qr_larftb.argtypes = [C.c_int] + [C.c_long] * 5 + [c_double_p, c_double_p, c_double_p, c_double_p, C.c_void_p]
lib.stmmqr_last_seam_ms.restype = C.c_double
lib.stmmqr_last_seam_ms.argtypes = []
lib.qr_cpack.restype = C.c_long
lib.qr_cpack.argtypes = [C.c_long] * 4 + [c_double_p, c_double_p]
lib.qr_rhpack.restype = C.c_long
lib.qr_rhpack.argtypes = [C.c_int, C.c_long, C.c_long, C.c_long, c_long_p, c_double_p, c_double_p, c_long_p]
lib.qr_fcsize.restype = C.c_long
lib.qr_fcsize.argtypes = [C.c_long] * 4
lib.qr_assemble.restype = None
lib.qr_assemble.argtypes = [C.c_long, C.c_long, C.c_int] + [c_long_p] * 8 + [c_double_p, c_long_p, c_long_p,
                                                                             C.POINTER(c_double_p), c_long_p, c_long_p,
                                                                             c_long_p, c_long_p, c_double_p, c_long_p]
lib.stmmqr_plan_rsolve.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_long, c_double_p, C.c_long, C.c_long]
lib.stmmqr_read_matrix_market.argtypes = [C.c_char_p, c_long_p, c_long_p, c_long_p, C.POINTER(c_long_p), C.POINTER(c_long_p), C.POINTER(c_double_p)]
lib.qr_fsize.argtypes = [C.c_long] + [c_long_p] * 9
lib.stmmqr_factorize_begin.argtypes = [C.c_void_p, c_long_p, c_long_p, C.c_void_p, C.c_int, C.c_double, C.c_long]
lib.stmmqr_factorize_group.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.stmmqr_factorize_finish.argtypes = [C.c_void_p, C.POINTER(Stats)]
lib.stmmqr_plan_set_groups.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
lib.stmmqr_plan_set_early_end.argtypes = [C.c_void_p, C.c_int]
lib.stmmqr_plan_front_info.argtypes = [C.c_void_p, C.c_long, c_long_p]
lib.stmmqr_plan_export_front.argtypes = [C.c_void_p, C.c_long, C.c_void_p, c_long_p, C.c_int]
lib.stmmqr_plan_import_front.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p, c_long_p, C.c_int]
lib.stmmqr_plan_qmult.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_long, C.c_long]
lib.stmmqr_plan_group_steps.argtypes = [C.c_void_p, C.c_int]
lib.stmmqr_factorize_step.argtypes = [C.c_void_p] + [C.c_int] * 6
lib.stmmqr_plan_panel_doubles.argtypes = [C.c_void_p, C.c_long, c_long_p]
lib.stmmqr_plan_export_panel.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_int]
lib.stmmqr_plan_import_panel.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_int]
lib.stmmqr_plan_export_front_cols.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, c_long_p]
lib.stmmqr_plan_import_front_cols.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.stmmqr_plan_front_rhoff.argtypes = [C.c_void_p, C.c_long, c_long_p]
lib.stmmqr_plan_front_flops.argtypes = [C.c_void_p, C.c_long, c_double_p]
lib.stmmqr_plan_device_bytes.argtypes = [C.c_void_p]
lib.stmmqr_plan_device_bytes.restype = C.c_double
lib.stmmqr_plan_solve.argtypes = [C.c_void_p, c_double_p, C.c_long, c_double_p, C.c_long, C.c_long]
lib.stmmqr_get_options.argtypes = [C.POINTER(Options)]
lib.stmmqr_set_options.argtypes = [C.POINTER(Options)]


class TransportC(C.Structure):
    """stmmqr_transport (include/stmmqr_hip.h): callbacks on device buffers, enqueued on the HIP stream they are given"""
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    GRP = C.CFUNCTYPE(C.c_int, C.c_void_p)
    _fields_ = [("ctx", C.c_void_p), ("send", SEND), ("recv", RECV), ("group_begin", GRP), ("group_end", GRP), ("rank", C.c_int),
                ("size", C.c_int)]


lib.stmmqr_factorize_shared_front.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.POINTER(TransportC)]


class ShardPhasesC(C.Structure):
    """stmmqr_shard_phases (include/stmmqr_hip.h): what one rank does in each phase of a sharded factorization"""
    _LP, _IP = C.POINTER(C.c_long), C.POINTER(C.c_int)
    _fields_ = [("nphase", C.c_int), ("out_ptr", _LP), ("out_front", _LP), ("out_peer", _IP), ("in_ptr", _LP), ("in_front", _LP),
                ("in_peer", _IP), ("shared_front", _LP), ("shared_first", _IP), ("shared_span", _IP), ("has_group", _IP)]


lib.stmmqr_factorize_exchange.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_long, C.POINTER(C.c_long),
                                          C.POINTER(C.c_int), C.POINTER(TransportC)]
lib.stmmqr_shared_front_gather.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.POINTER(TransportC)]
lib.stmmqr_factorize_phases.argtypes = [C.c_void_p, C.POINTER(ShardPhasesC), C.POINTER(TransportC)]
lib.stmmqr_rccl_unique_id.argtypes = [C.c_char_p]
lib.stmmqr_rccl_transport_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.POINTER(TransportC))]
lib.stmmqr_rccl_transport_destroy.restype = None
lib.stmmqr_rccl_transport_destroy.argtypes = [C.POINTER(TransportC)]
lib.stmmqr_transport_sendrecv.argtypes = [C.POINTER(TransportC), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
lib.stmmqr_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


class RcclTransport:
    """RCCL point-to-point transport of the native shared-front loop (stmmqr_rccl_transport_create).  `dist`: an initialised
    torch.distributed (any backend) that carries the 128-byte id from rank 0 to the others; the communicator itself is this
    library's own (ncclCommInitRank on the copy of librccl the process already holds)."""

    def __init__(self, dist, device=None):
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = C.create_string_buffer(128)
        if rank == 0:
            _check(lib.stmmqr_rccl_unique_id(ident), "stmmqr_rccl_unique_id")
        t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
        if device is not None and dist.get_backend() == "nccl":
            t = t.to(device)
        dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        self._p = C.POINTER(TransportC)()
        _check(lib.stmmqr_rccl_transport_create(world, rank, raw, C.byref(self._p)), "stmmqr_rccl_transport_create")
        self.rank, self.size = rank, world

    @property
    def ptr(self):
        return self._p

    def sendrecv(self, send_ptr, send_bytes, dst, recv_ptr, recv_bytes, src, stream=None):
        _check(lib.stmmqr_transport_sendrecv(self._p, send_ptr, send_bytes, dst, recv_ptr, recv_bytes, src, stream), "stmmqr_transport_sendrecv")

    def close(self):
        if self._p:
            lib.stmmqr_rccl_transport_destroy(self._p)
            self._p = C.POINTER(TransportC)()


class CallbackTransport:
    """A stmmqr_transport whose send / receive are Python callables (tests: the ranks of a group as threads on one GPU)"""

    def __init__(self, rank, size, send, recv):
        self._send = TransportC.SEND(lambda ctx, buf, nbytes, peer, stream: int(send(buf, nbytes, peer, stream) or 0))
        self._recv = TransportC.RECV(lambda ctx, buf, nbytes, peer, stream: int(recv(buf, nbytes, peer, stream) or 0))
        self._t = TransportC(None, self._send, self._recv, TransportC.GRP(), TransportC.GRP(), rank, size)
        self.rank, self.size = rank, size

    @property
    def ptr(self):
        return C.pointer(self._t)


def device_copy(dst: int, src: int, nbytes: int, stream=None):
    _check(lib.stmmqr_device_copy(C.c_void_p(dst), C.c_void_p(src), nbytes, stream), "stmmqr_device_copy")


def last_error() -> str:
    return lib.stmmqr_last_error().decode()


def device_count() -> int:
    return int(lib.stmmqr_device_count())


def device_name(i: int = 0) -> str:
    return lib.stmmqr_device_name(i).decode()


def get_options() -> dict:
    o = Options()
    lib.stmmqr_get_options(C.byref(o))
    return {k: getattr(o, k) for k, _ in o._fields_}


lib.stmmqr_device_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
lib.stmmqr_device_free.argtypes = [C.c_void_p]


def device_alloc(nbytes: int) -> int:
    """Device buffer from the HIP runtime the library is bound to (current device); release with device_free."""
    p = C.c_void_p()
    _check(lib.stmmqr_device_alloc(nbytes, C.byref(p)), "stmmqr_device_alloc")
    return p.value


def device_free(ptr: int):
    _check(lib.stmmqr_device_free(ptr), "stmmqr_device_free")


def set_options(**kw):
    o = Options()
    lib.stmmqr_get_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    lib.stmmqr_set_options(C.byref(o))


def _check(rc: int, what: str):
    if rc != 0:
        e = StmmqrError(f"{what} failed ({rc}): {last_error()}")
        e.code = int(rc)
        raise e


class QRNumeric:
    """Host copy of the reference's qr_numeric (STMMQR/include/SparseQR_struct.h:145-209), one stack."""

    def __init__(self, nf, n, m, rjsize, hisize, rh_total):
        self.Stack = np.zeros(max(rh_total, 1))
        self.Rblock_off = np.zeros(max(nf, 1), I64)
        self.Rdead = np.zeros(max(n, 1), np.int8)
        self.HStair = np.zeros(max(rjsize, 1), I64)
        self.HTau = np.zeros(max(rjsize, 1))
        self.Hii = np.zeros(max(hisize, 1), I64)
        self.HPinv = np.zeros(max(m, 1), I64)
        self.Hm = np.zeros(max(nf, 1), I64)
        self.Hr = np.zeros(max(nf, 1), I64)
        self.rh_total = rh_total
        self.rank = self.rank1 = self.maxfrank = self.maxfm = 0
        self.stats = {}


class HipQR:
    """A device plan for one symbolic analysis (reusable across numeric factorizations).

    ``sym`` is a dict with the qr_symbolic arrays/scalars of the same names (int64 numpy arrays).
    """

    def __init__(self, sym: dict, device: int = -1):
        self._keep = {}
        v = SymbolicView()
        for k in ["m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "do_rank_detection"]:
            setattr(v, k, int(sym[k]))
        for k in ["Sp", "Sj", "Qfill", "PLinv", "Sleft", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip", "Fm"]:
            a = sym.get(k)
            if a is not None and np.size(a) == 0 and k == "Qfill":
                a = None
            if a is not None:
                a = np.ascontiguousarray(a, dtype=I64)
                self._keep[k] = a
            setattr(v, k, _ip(a))
        v.maxstack = int(sym.get("maxstack", 0) or 0)
        self.sym = {k: int(sym[k]) for k in ["m", "n", "anz", "nf", "maxfn", "rjsize", "hisize"]}
        st = C.c_int(0)
        self._h = lib.stmmqr_plan_create(C.byref(v), device, C.byref(st))
        if not self._h:
            raise StmmqrError(f"stmmqr_plan_create failed ({st.value}): {last_error()}")

    def close(self):
        if getattr(self, "_h", None):
            lib.stmmqr_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def set_pattern(self, Ap, Ai):
        Ap = np.ascontiguousarray(Ap, I64); Ai = np.ascontiguousarray(Ai, I64)
        _check(lib.stmmqr_plan_set_pattern(self._h, _ip(Ap), _ip(Ai)), "stmmqr_plan_set_pattern")

    def factorize(self, Ax, tol, ntol, Ap=None, Ai=None, device_ptr: int | None = None, detail=False) -> dict:
        """Numeric factorization; results stay in HBM.  Ax: numpy array, or device_ptr (int) of a device buffer."""
        st = Stats()
        if detail:
            st.nlaunch = -1
        if Ap is not None:
            Ap = np.ascontiguousarray(Ap, I64); Ai = np.ascontiguousarray(Ai, I64)
        if device_ptr is not None:
            rc = lib.stmmqr_factorize_device(self._h, _ip(Ap), _ip(Ai), C.c_void_p(device_ptr), 1, float(tol), int(ntol),
                                             C.byref(st))
        else:
            Ax = np.ascontiguousarray(Ax, np.float64)
            rc = lib.stmmqr_factorize_device(self._h, _ip(Ap), _ip(Ai), Ax.ctypes.data_as(C.c_void_p), 0, float(tol),
                                             int(ntol), C.byref(st))
        _check(rc, "stmmqr_factorize_device")
        return st.as_dict()

    # ---- phased interface / subtree sharding (include/stmmqr_hip.h, "Subtree sharding") ----
    def set_groups(self, group):
        g = np.ascontiguousarray(group, np.int32)
        assert g.size == self.sym["nf"]
        _check(lib.stmmqr_plan_set_groups(self._h, g.ctypes.data_as(C.POINTER(C.c_int))), "stmmqr_plan_set_groups")

    def set_early_end(self, mode: int):
        """1: the cut schedule under the phased interface (the caller handles ERR_RESCHEDULE at finish); 0: every panel a step"""
        _check(lib.stmmqr_plan_set_early_end(self._h, int(mode)), "stmmqr_plan_set_early_end")

    def begin(self, Ax, tol, ntol, Ap=None, Ai=None, device_ptr=None):
        if Ap is not None:
            Ap = np.ascontiguousarray(Ap, I64); Ai = np.ascontiguousarray(Ai, I64)
        if device_ptr is not None:
            rc = lib.stmmqr_factorize_begin(self._h, _ip(Ap), _ip(Ai), C.c_void_p(device_ptr), 1, float(tol), int(ntol))
        else:
            Ax = np.ascontiguousarray(Ax, np.float64)
            rc = lib.stmmqr_factorize_begin(self._h, _ip(Ap), _ip(Ai), Ax.ctypes.data_as(C.c_void_p), 0, float(tol), int(ntol))
        _check(rc, "stmmqr_factorize_begin")

    def run_group(self, g, detail=False):
        _check(lib.stmmqr_factorize_group(self._h, int(g), int(detail)), "stmmqr_factorize_group")

    def finish(self) -> dict:
        st = Stats()
        _check(lib.stmmqr_factorize_finish(self._h, C.byref(st)), "stmmqr_factorize_finish")
        return st.as_dict()

    def front_info(self, f) -> dict:
        info = np.zeros(6, I64)
        _check(lib.stmmqr_plan_front_info(self._h, int(f), _ip(info)), "stmmqr_plan_front_info")
        return dict(zip(["fm", "rank", "cm", "csize", "fn", "fp"], (int(x) for x in info)))

    def export_front(self, f):
        """-> (info, packed C [csize], row ids [cm]) of a factorized front (host arrays)."""
        info = self.front_info(f)
        Cb = np.zeros(max(info["csize"], 1)); rows = np.zeros(max(info["cm"], 1), I64)
        _check(lib.stmmqr_plan_export_front(self._h, int(f), Cb.ctypes.data_as(C.c_void_p), _ip(rows), 0),
               "stmmqr_plan_export_front")
        return info, Cb[:info["csize"]], rows[:info["cm"]]

    def export_front_dev(self, f, dev_ptr: int, info=None):
        """packed C of front f copied device-to-device into the buffer at dev_ptr (>= csize doubles); returns the row ids
        (host int64).  The contribution block never crosses PCIe (multi-GPU: it is then sent by RCCL from that buffer)."""
        info = info or self.front_info(f)
        rows = np.zeros(max(info["cm"], 1), I64)
        _check(lib.stmmqr_plan_export_front(self._h, int(f), C.c_void_p(dev_ptr), _ip(rows), 1), "stmmqr_plan_export_front")
        return rows[:info["cm"]]

    def import_front_dev(self, f, fm, rank, cm, dev_ptr: int, rows):
        rows = np.ascontiguousarray(rows, I64)
        _check(lib.stmmqr_plan_import_front(self._h, int(f), int(fm), int(rank), int(cm), C.c_void_p(dev_ptr), _ip(rows), 1),
               "stmmqr_plan_import_front")

    def import_front(self, f, fm, rank, cm, Cb, rows):
        Cb = np.ascontiguousarray(Cb, np.float64); rows = np.ascontiguousarray(rows, I64)
        _check(lib.stmmqr_plan_import_front(self._h, int(f), int(fm), int(rank), int(cm), Cb.ctypes.data_as(C.c_void_p),
                                            _ip(rows), 0), "stmmqr_plan_import_front")

    # ---- a front shared between plans (include/stmmqr_hip.h, "A front SHARED between plans") ----
    SHARED = 1 << 30
    PREP, PANEL, UPDATE, GRAM, POST = 1, 2, 4, 8, 16

    def group_steps(self, g) -> int:
        n = lib.stmmqr_plan_group_steps(self._h, int(g))
        if n < 0:
            raise StmmqrError(f"stmmqr_plan_group_steps failed: {last_error()}")
        return int(n)

    def run_step(self, g, step, what, cb_first=0, cb_stride=1, cb_count=-1):
        _check(lib.stmmqr_factorize_step(self._h, int(g), int(step), int(what), int(cb_first), int(cb_stride), int(cb_count)),
               "stmmqr_factorize_step")

    def shared_front_native(self, group, f, first_rank, nranks, transport):
        """the whole panel loop of a shared front in ONE native call (stmmqr_factorize_shared_front); transport: RcclTransport or
        CallbackTransport"""
        _check(lib.stmmqr_factorize_shared_front(self._h, int(group), int(f), int(first_rank), int(nranks), transport.ptr),
               "stmmqr_factorize_shared_front")

    def exchange_native(self, out, inn, transport):
        """the contribution blocks entering a phase, [(front, peer)] each way, in ONE native call (stmmqr_factorize_exchange)"""
        of, op = np.array([c for c, _ in out], I64), np.array([p for _, p in out], np.int32)
        nf, npr = np.array([c for c, _ in inn], I64), np.array([p for _, p in inn], np.int32)
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))       # noqa: E731
        _check(lib.stmmqr_factorize_exchange(self._h, len(of), _ip(of), i32(op), len(nf), _ip(nf), i32(npr), transport.ptr),
               "stmmqr_factorize_exchange")

    def gather_native(self, f, first_rank, nranks, transport):
        _check(lib.stmmqr_shared_front_gather(self._h, int(f), int(first_rank), int(nranks), transport.ptr), "stmmqr_shared_front_gather")

    def phases_native(self, phases, transport):
        """every phase of a sharded factorization in ONE native call (stmmqr_factorize_phases); phases: ShardPlan.c_phases()"""
        _check(lib.stmmqr_factorize_phases(self._h, C.byref(phases[0]), transport.ptr), "stmmqr_factorize_phases")

    def panel_doubles(self, f) -> int:
        n = np.zeros(1, I64)
        _check(lib.stmmqr_plan_panel_doubles(self._h, int(f), _ip(n)), "stmmqr_plan_panel_doubles")
        return int(n[0])

    def export_panel(self, f, p):
        buf = np.zeros(self.panel_doubles(f))
        _check(lib.stmmqr_plan_export_panel(self._h, int(f), int(p), buf.ctypes.data_as(C.c_void_p), 0), "stmmqr_plan_export_panel")
        return buf

    def export_panel_dev(self, f, p, dev_ptr: int):
        _check(lib.stmmqr_plan_export_panel(self._h, int(f), int(p), C.c_void_p(dev_ptr), 1), "stmmqr_plan_export_panel")

    def import_panel(self, f, p, buf):
        buf = np.ascontiguousarray(buf, np.float64)
        assert buf.size >= self.panel_doubles(f)
        _check(lib.stmmqr_plan_import_panel(self._h, int(f), int(p), buf.ctypes.data_as(C.c_void_p), 0), "stmmqr_plan_import_panel")

    def import_panel_dev(self, f, p, dev_ptr: int):
        _check(lib.stmmqr_plan_import_panel(self._h, int(f), int(p), C.c_void_p(dev_ptr), 1), "stmmqr_plan_import_panel")

    def front_cols_doubles(self, f, part, nparts) -> int:
        n = np.zeros(1, I64)
        _check(lib.stmmqr_plan_export_front_cols(self._h, int(f), int(part), int(nparts), None, 0, _ip(n)), "stmmqr_plan_export_front_cols")
        return int(n[0])

    def export_front_cols(self, f, part, nparts):
        buf = np.zeros(max(self.front_cols_doubles(f, part, nparts), 1))
        n = np.zeros(1, I64)
        _check(lib.stmmqr_plan_export_front_cols(self._h, int(f), int(part), int(nparts), buf.ctypes.data_as(C.c_void_p), 0, _ip(n)),
               "stmmqr_plan_export_front_cols")
        return buf[:int(n[0])]

    def export_front_cols_dev(self, f, part, nparts, dev_ptr: int) -> int:
        n = np.zeros(1, I64)
        _check(lib.stmmqr_plan_export_front_cols(self._h, int(f), int(part), int(nparts), C.c_void_p(dev_ptr), 1, _ip(n)),
               "stmmqr_plan_export_front_cols")
        return int(n[0])

    def import_front_cols(self, f, part, nparts, buf):
        buf = np.ascontiguousarray(buf, np.float64)
        if buf.size:
            _check(lib.stmmqr_plan_import_front_cols(self._h, int(f), int(part), int(nparts), buf.ctypes.data_as(C.c_void_p), 0),
                   "stmmqr_plan_import_front_cols")

    def import_front_cols_dev(self, f, part, nparts, dev_ptr: int):
        _check(lib.stmmqr_plan_import_front_cols(self._h, int(f), int(part), int(nparts), C.c_void_p(dev_ptr), 1),
               "stmmqr_plan_import_front_cols")

    def front_rhoff(self, f, fn):
        off = np.zeros(int(fn) + 1, I64)
        _check(lib.stmmqr_plan_front_rhoff(self._h, int(f), _ip(off)), "stmmqr_plan_front_rhoff")
        return off

    def device_bytes(self) -> float:
        return float(lib.stmmqr_plan_device_bytes(self._h))

    def result_sizes(self):
        """-> (entries of the packed R+H, rank) of the factorization held"""
        rh, rk = C.c_long(0), C.c_long(0)
        _check(lib.stmmqr_plan_result_sizes(self._h, C.byref(rh), C.byref(rk)), "stmmqr_plan_result_sizes")
        return int(rh.value), int(rk.value)

    def front_flops(self, f):
        """-> (reference flop count of front f, the part done by trailing updates) of the factorization in progress / held"""
        v = np.zeros(2)
        _check(lib.stmmqr_plan_front_flops(self._h, int(f), _dp(v)), "stmmqr_plan_front_flops")
        return float(v[0]), float(v[1])

    def qmult(self, method: int, X: np.ndarray) -> np.ndarray:
        """QR_qmult (SparseQR.h:403-409) on the resident factors: method 0 = QR_QTX (Q'X), 1 = QR_QX (Q X) with X m or
        m x nrhs; 2 = QR_XQT (X Q'), 3 = QR_XQ (X Q) with X k x m.  Returns a new array (orders as in the reference: Q'X
        and X Q in the permuted order of the factorization)."""
        m = self.sym["m"]
        if method <= 1:
            Xf = np.array(X, dtype=np.float64, order="F", copy=True).reshape(m, -1, order="F")
            _check(lib.stmmqr_plan_qmult(self._h, int(method), _dp(Xf), m, Xf.shape[1]), "stmmqr_plan_qmult")
            return Xf.reshape(np.shape(X), order="F")
        Xf = np.array(X, dtype=np.float64, order="F", copy=True).reshape(-1, m, order="F")
        _check(lib.stmmqr_plan_qmult(self._h, int(method), _dp(Xf), Xf.shape[0], Xf.shape[0]), "stmmqr_plan_qmult")
        return Xf.reshape(np.shape(X), order="F")

    def rsolve(self, system: int, B: np.ndarray) -> np.ndarray:
        """QR_solve (SparseQR.h:411-417): system 0 RX=B, 1 RE'X=B (B m x nrhs in R's row order, X n x nrhs), 2 R'X=B,
        3 R'X=E'B (B n x nrhs, X m x nrhs)."""
        m, n = self.sym["m"], self.sym["n"]
        br, xr = (m, n) if system <= 1 else (n, m)
        Bf = np.array(B, dtype=np.float64, order="F", copy=True).reshape(br, -1, order="F")
        X = np.zeros((xr, Bf.shape[1]), order="F")
        _check(lib.stmmqr_plan_rsolve(self._h, int(system), _dp(Bf), br, _dp(X), xr, Bf.shape[1]), "stmmqr_plan_rsolve")
        return X[:, 0] if np.ndim(B) == 1 else X

    def solve(self, B: np.ndarray) -> np.ndarray:
        """QR_solve(QR_RETX_EQUALS_B) (SparseQR.h:411-417): X = E R^-1 (Q'B)(1:n), least-squares solution; rank == n only."""
        m, n = self.sym["m"], self.sym["n"]
        Bf = np.array(B, dtype=np.float64, order="F", copy=True).reshape(m, -1, order="F")
        X = np.zeros((n, Bf.shape[1]), order="F")
        _check(lib.stmmqr_plan_solve(self._h, _dp(Bf), m, _dp(X), n, Bf.shape[1]), "stmmqr_plan_solve")
        return X[:, 0] if np.ndim(B) == 1 else X

    def download(self) -> QRNumeric:
        rh = C.c_long(0); rk = C.c_long(0)
        _check(lib.stmmqr_plan_result_sizes(self._h, C.byref(rh), C.byref(rk)), "stmmqr_plan_result_sizes")
        s = self.sym
        N = QRNumeric(s["nf"], s["n"], s["m"], s["rjsize"], s["hisize"], rh.value)
        sc = np.zeros(4, I64)
        st = Stats()
        _check(lib.stmmqr_plan_download(self._h, _dp(N.Stack), _ip(N.Rblock_off),
                                        C.cast(N.Rdead.ctypes.data, C.c_char_p), _ip(N.HStair), _dp(N.HTau), _ip(N.Hii),
                                        _ip(N.HPinv), _ip(N.Hm), _ip(N.Hr), _ip(sc), C.byref(st)),
               "stmmqr_plan_download")
        N.rank, N.rank1, N.maxfrank, N.maxfm = (int(x) for x in sc)
        N.stats = st.as_dict()
        return N


def qr_factorize(sym: dict, Ap, Ai, Ax, tol, ntol) -> QRNumeric:
    """qr_factorize(&A, freeA, tol, ntol, QRsym, cc) on arrays (SparseQR.h:127-135)."""
    plan = HipQR(sym)
    try:
        stats = plan.factorize(Ax, tol, ntol, Ap, Ai)
        N = plan.download()
        N.stats = {**stats, **{k: v for k, v in N.stats.items() if k in ("ms_d2h", "ms_host")}}
        return N
    finally:
        plan.close()


class SeamNumeric:
    """What the exported qr_factorize (the drop-in seam with the reference's structs) returned: a view of the qr_numeric; the
    arrays are copied out on demand, close() releases it through stmmqr_free_numeric."""

    def __init__(self, ptr, nf, n, m, rjsize, hisize):
        self._p, self.nf, self.n, self.m, self.rjsize, self.hisize = ptr, nf, n, m, rjsize, hisize
        c = ptr.contents
        self.rank, self.rank1, self.maxfrank, self.maxfm = c.rank, c.rank1, c.maxfrank, c.maxfm
        self.rh_total = int(c.Stack_size[0])

    def arrays(self) -> dict:
        c = self._p.contents
        take = lambda ptr, cnt, ty: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(max(cnt, 1),))[:cnt].copy()
        stacks = C.cast(c.Stacks, C.POINTER(c_double_p))
        base = C.cast(stacks[0], C.c_void_p).value
        rb = C.cast(c.Rblock, C.POINTER(C.c_void_p))
        return {"Stack": take(stacks[0], self.rh_total, C.c_double), "Rdead": take(c.Rdead, self.n, C.c_int8),
                "HStair": take(c.HStair, self.rjsize, C.c_long), "HTau": take(c.HTau, self.rjsize, C.c_double),
                "Hii": take(c.Hii, self.hisize, C.c_long), "HPinv": take(c.HPinv, self.m, C.c_long),
                "Hm": take(c.Hm, self.nf, C.c_long), "Hr": take(c.Hr, self.nf, C.c_long),
                "Rblock_off": np.array([((rb[f] or base) - base) // 8 for f in range(self.nf)], I64)}

    def close(self):
        if self._p:
            pp = C.POINTER(QrNumericC)(self._p.contents)
            lib.stmmqr_free_numeric(C.byref(pp), None)
            self._p = None

    __del__ = close


def qr_factorize_seam(sym: dict, Ap, Ai, Ax, tol, ntol) -> SeamNumeric:
    """ONE call of the exported qr_factorize -- the drop-in seam itself, with the reference's structs (sparse_csc **, freeA = 0,
    qr_symbolic *, cc = NULL), exactly what SparseQR.c:349 calls: plan (built or from the seam's cache), H2D, factorization, host
    allocation of the qr_numeric, D2H of the packed factors."""
    keep = {}
    S = QrSymbolicC()
    for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "do_rank_detection", "maxstack", "hisize", "keepH", "ntasks", "ns"):
        setattr(S, k, int(sym.get(k, 1 if k in ("keepH", "ntasks", "ns") else 0)))
    for k in ("Sp", "Sj", "Qfill", "PLinv", "Sleft", "Parent", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip", "Fm", "Cm"):
        a = sym.get(k)
        if a is not None and len(a):
            keep[k] = np.ascontiguousarray(a, I64)
            setattr(S, k, _ip(keep[k]))
    keep["Ap"], keep["Ai"], keep["Ax"] = np.ascontiguousarray(Ap, I64), np.ascontiguousarray(Ai, I64), np.ascontiguousarray(Ax, np.float64)
    A = SparseCsc()
    A.nrow, A.ncol, A.nzmax = S.m, S.n, len(keep["Ax"])
    A.p, A.i, A.x = keep["Ap"].ctypes.data, keep["Ai"].ctypes.data, keep["Ax"].ctypes.data
    A.stype, A.itype, A.xtype, A.dtype, A.sorted, A.packed = 0, 2, 1, 0, 1, 1
    Ah = C.pointer(A)
    lib.qr_factorize.restype = C.POINTER(QrNumericC)
    lib.qr_factorize.argtypes = [C.POINTER(C.POINTER(SparseCsc)), C.c_long, C.c_double, C.c_long, C.POINTER(QrSymbolicC), C.c_void_p]
    lib.stmmqr_free_numeric.restype = None
    lib.stmmqr_free_numeric.argtypes = [C.POINTER(C.POINTER(QrNumericC)), C.c_void_p]
    N = lib.qr_factorize(C.byref(Ah), 0, float(tol), int(ntol), C.byref(S), None)
    if not N:
        raise StmmqrError("qr_factorize (seam) returned NULL: " + last_error())
    return SeamNumeric(N, S.nf, S.n, S.m, S.rjsize, S.hisize)


def plan_cache_clear():
    lib.stmmqr_plan_cache_clear()


def qr_front(m, n, npiv, tol, ntol, F, Stair):
    """qr_front (SparseQR.h:209-226).  F: (m,n) Fortran-ordered, modified in place; Stair int64, in/out.
    Returns (rank, Tau, Rdead, flops)."""
    assert F.flags.f_contiguous and F.dtype == np.float64 and F.shape == (m, n)
    assert Stair.dtype == I64
    Tau = np.zeros(max(n, 1)); Rdead = np.zeros(max(n, 1), np.int8); fl = C.c_double(0)
    r = lib.stmmqr_front(m, n, npiv, float(tol), int(ntol), _dp(F), _ip(Stair), C.cast(Rdead.ctypes.data, C.c_char_p),
                         _dp(Tau), C.byref(fl))
    if r < 0:
        raise StmmqrError(f"stmmqr_front failed: {last_error()}")
    return int(r), Tau[:n], Rdead[:min(n, max(npiv, 0))], fl.value


def last_seam_ms() -> float:
    """Device time (ms) of the kernels of the last qr_front / qr_assemble seam call (-1: none)."""
    return float(lib.stmmqr_last_seam_ms())


def qr_larftb(method, m, n, k, ldc, ldv, V, Tau, Cmat):
    """qr_larftb (SparseQR.h:255-268; SparseQR_factorize.c:1851-1904), C in place: method 0 QR_QTX C <- Q'C, 1 QR_QX
    C <- QC (C m x n, V m x k), 2 QR_XQT C <- CQ', 3 QR_XQ C <- CQ (C m x n, V n x k), Q = I - V T V'."""
    _check(lib.stmmqr_larftb(method, m, n, k, ldc, ldv, _dp(V), _dp(Tau), _dp(Cmat)), "stmmqr_larftb")


def qr_fcsize(m, n, npiv, g):
    return int(lib.qr_fcsize(m, n, npiv, g))


def qr_cpack(m, n, npiv, g, F):
    """qr_cpack (SparseQR.h:235-242): returns (cm, packed C)."""
    Cp = np.zeros(max(qr_fcsize(m, n, npiv, g), 1))
    cm = lib.qr_cpack(m, n, npiv, g, _dp(F), _dp(Cp))
    if cm < 0:
        raise StmmqrError("qr_cpack failed")
    return int(cm), Cp[:qr_fcsize(m, n, npiv, g)]


def qr_rhpack(m, n, npiv, Stair, F):
    """qr_rhpack with keepH (SparseQR.h:244-253): returns (rsize, rm, packed R+H)."""
    R = np.zeros(max(m * n, 1)); rm = C.c_long(0)
    rs = lib.qr_rhpack(1, m, n, npiv, _ip(Stair), _dp(F), _dp(R), C.byref(rm))
    if rs < 0:
        raise StmmqrError("qr_rhpack failed")
    return int(rs), int(rm.value), R[:rs]


class SparseCsc(C.Structure):
    """sparse_csc (STMMQR/include/SparseCore.h:514-556), the fields the seams read"""
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t), ("p", C.c_void_p), ("i", C.c_void_p),
                ("nz", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p), ("stype", C.c_int), ("itype", C.c_int),
                ("xtype", C.c_int), ("dtype", C.c_int), ("sorted", C.c_int), ("packed", C.c_int)]


class QrSymbolicC(C.Structure):
    """qr_symbolic (SparseQR_struct.h:26-137), field order of include/stmmqr_hip.h"""
    _fields_ = [("m", C.c_long), ("n", C.c_long), ("anz", C.c_long)] + \
               [(k, c_long_p) for k in ("Sp", "Sj", "Qfill", "PLinv", "Sleft")] + [("nf", C.c_long), ("maxfn", C.c_long)] + \
               [(k, c_long_p) for k in ("Parent", "Child", "Childp", "Super", "Rp", "Rj", "Post")] + \
               [(k, C.c_long) for k in ("rjsize", "do_rank_detection", "maxstack", "hisize", "keepH")] + [("Hip", c_long_p)] + \
               [("ntasks", C.c_long), ("ns", C.c_long)] + \
               [(k, c_long_p) for k in ("TaskChildp", "TaskChild", "TaskStack", "TaskFront", "TaskFrontp", "On_stack",
                                        "Stack_maxstack", "Fm", "Cm")]


class QrNumericC(C.Structure):
    """qr_numeric (SparseQR_struct.h:145-209)"""
    _fields_ = [("Rblock", C.c_void_p), ("Stacks", C.c_void_p), ("Stack_size", c_long_p)] + \
               [(k, C.c_long) for k in ("hisize", "n", "m", "nf", "ntasks", "ns", "maxstack")] + [("Rdead", C.c_char_p)] + \
               [(k, C.c_long) for k in ("rank", "rank1", "maxfrank")] + [("norm_E_fro", C.c_double)] + \
               [("keepH", C.c_long), ("rjsize", C.c_long), ("HStair", c_long_p), ("HTau", c_double_p)] + \
               [(k, c_long_p) for k in ("Hii", "HPinv", "Hm", "Hr")] + [("maxfm", C.c_long)]


class Relax(C.Structure):
    _fields_ = [("nrelax", C.c_long * 3), ("zrelax", C.c_double * 3)]


lib.stmmqr_relax_for_qr.restype = None
lib.stmmqr_relax_for_qr.argtypes = [C.c_long, C.c_long, C.POINTER(Relax)]
lib.stmmqr_analyze.argtypes = [C.c_long, C.c_long, c_long_p, c_long_p, c_long_p, C.c_int, C.POINTER(Relax), C.POINTER(C.c_void_p)]
lib.stmmqr_analysis_symbolic.restype = C.POINTER(QrSymbolicC)
lib.stmmqr_analysis_symbolic.argtypes = [C.c_void_p]
lib.stmmqr_analysis_info.argtypes = [C.c_void_p, c_double_p]
lib.stmmqr_analysis_free.restype = None
lib.stmmqr_analysis_free.argtypes = [C.c_void_p]


def relax_for_qr(n: int, nnz: int) -> Relax:
    """Relaxfactor_setting(n, nnz, RELAX_FOR_QR) (SparseCore_common.c:1172-1203, called by the driver at qrtest.c:153)."""
    r = Relax()
    lib.stmmqr_relax_for_qr(n, nnz, C.byref(r))
    return r


def analyze(m: int, n: int, Ap, Ai, Qfill=None, do_rank_detection: bool = True, relax: Relax | None = None) -> dict:
    """qr_analyze (SparseQR_analyze.c:20-700) on plain arrays: the whole qr_symbolic as a dict of numpy arrays / ints,
    plus 'info' = [flop bound, fl, lnz, QR_CHUNK_FLAG, nnz(R) bound, nnz(H) bound, maxstack, nf].  Host-only."""
    Ap = np.ascontiguousarray(Ap, I64)
    Ai = np.ascontiguousarray(Ai, I64)
    Q = None if Qfill is None else np.ascontiguousarray(Qfill, I64)
    h = C.c_void_p()
    _check(lib.stmmqr_analyze(m, n, _ip(Ap), _ip(Ai), _ip(Q), 1 if do_rank_detection else 0,
                              None if relax is None else C.byref(relax), C.byref(h)), "stmmqr_analyze")
    try:
        S = lib.stmmqr_analysis_symbolic(h).contents
        nf, anz = S.nf, S.anz
        out = {k: int(getattr(S, k)) for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "do_rank_detection", "maxstack", "hisize",
                                               "keepH", "ntasks", "ns")}
        sizes = {"Sp": m + 1, "Sj": anz, "Qfill": n, "PLinv": m, "Sleft": n + 2, "Parent": nf + 1, "Child": nf + 1,
                 "Childp": nf + 2, "Super": nf + 1, "Rp": nf + 1, "Rj": S.Rp[nf] if nf > 0 else 0, "Post": nf + 1, "Hip": nf + 1,
                 "Fm": nf + 1, "Cm": nf + 1}
        for k, cnt in sizes.items():
            ptr = getattr(S, k)
            out[k] = np.ctypeslib.as_array(ptr, shape=(cnt,)).copy() if cnt > 0 else np.zeros(0, I64)
        info = np.zeros(8)
        _check(lib.stmmqr_analysis_info(h, _dp(info)), "stmmqr_analysis_info")
        out["info"] = info
        return out
    finally:
        lib.stmmqr_analysis_free(h)


lib.stmmqr_sparseqr.argtypes = [C.c_int, C.c_double, C.c_long, C.c_long, c_long_p, c_long_p, c_double_p, c_long_p, C.POINTER(Relax), C.c_int,
                                C.POINTER(C.c_void_p)]
lib.stmmqr_sparseqr_symbolic.argtypes = [C.c_int, C.c_double, C.c_long, C.c_long, c_long_p, c_long_p, c_double_p, c_long_p, C.POINTER(Relax),
                                         C.POINTER(C.c_void_p)]
lib.stmmqr_sparseqr_numeric.argtypes = [C.c_void_p, C.c_int]
lib.stmmqr_sparseqr_free.restype = None
lib.stmmqr_sparseqr_free.argtypes = [C.c_void_p]
lib.stmmqr_sparseqr_info.argtypes = [C.c_void_p, c_double_p]
lib.stmmqr_sparseqr_q1fill.restype = c_long_p
lib.stmmqr_sparseqr_q1fill.argtypes = [C.c_void_p]
lib.stmmqr_sparseqr_symbolic_view.restype = C.POINTER(QrSymbolicC)
lib.stmmqr_sparseqr_symbolic_view.argtypes = [C.c_void_p]
lib.stmmqr_sparseqr_y.argtypes = [C.c_void_p, C.POINTER(c_long_p), C.POINTER(c_long_p), C.POINTER(c_double_p)]
lib.stmmqr_sparseqr_qmult.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_long, C.c_long, C.c_long, c_double_p, C.c_long]
lib.stmmqr_sparseqr_solve.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_long, C.c_long, c_double_p, C.c_long]


lib.stmmqr_sparseqr_plan.restype = C.c_void_p
lib.stmmqr_sparseqr_plan.argtypes = [C.c_void_p]
lib.stmmqr_plan_export_r.argtypes = [C.c_void_p, C.POINTER(QrSymbolicC), C.c_long, C.POINTER(c_long_p), C.POINTER(c_long_p),
                                     C.POINTER(c_double_p), c_long_p, C.POINTER(c_long_p), C.POINTER(c_long_p), C.POINTER(c_double_p),
                                     C.POINTER(c_double_p)]
lib.stmmqr_sparselq.argtypes = [C.c_int, C.c_double, C.c_long, C.c_long, c_long_p, c_long_p, c_double_p, C.POINTER(Relax), C.c_int,
                                C.POINTER(C.c_void_p)]
lib.stmmqr_free.restype = None
lib.stmmqr_free.argtypes = [C.c_void_p]


def _take(ptr, count, dtype):
    out = np.ctypeslib.as_array(ptr, shape=(max(count, 1),))[:count].astype(dtype, copy=True)
    lib.stmmqr_free(C.cast(ptr, C.c_void_p))
    return out


def _symbolic_dict(S, m, n) -> dict:
    nf, anz = S.nf, S.anz
    out = {k: int(getattr(S, k)) for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "do_rank_detection", "maxstack", "hisize", "keepH",
                                           "ntasks", "ns")}
    sizes = {"Sp": m + 1, "Sj": anz, "Qfill": n, "PLinv": m, "Sleft": n + 2, "Parent": nf + 1, "Child": nf + 1, "Childp": nf + 2,
             "Super": nf + 1, "Rp": nf + 1, "Rj": S.Rp[nf] if nf > 0 else 0, "Post": nf + 1, "Hip": nf + 1, "Fm": nf + 1, "Cm": nf + 1}
    for k, cnt in sizes.items():
        out[k] = np.ctypeslib.as_array(getattr(S, k), shape=(cnt,)).copy() if cnt > 0 else np.zeros(0, I64)
    return out


class SparseQR:
    """SparseQR() / QR_qmult / QR_solve / SparseQR_free of the reference (STMMQR/include/SparseQR.h:25-36,403-417) on this library
    alone: singletons + COLAMD + symbolic analysis on the host, numeric factorization and the Q / R operations on the device.
    symbolic_only=True stops after the host half (no GPU needed)."""

    def __init__(self, m, n, Ap, Ai, Ax, ordering=7, tol=-2.0, relax: Relax | None = None, Quser=None, device=-1, symbolic_only=False):
        self.m, self.n = int(m), int(n)
        self._A = (np.ascontiguousarray(Ap, I64), np.ascontiguousarray(Ai, I64), np.ascontiguousarray(Ax, np.float64))
        Q = None if Quser is None else np.ascontiguousarray(Quser, I64)
        self._h = C.c_void_p()
        rp = None if relax is None else C.byref(relax)
        if symbolic_only:
            _check(lib.stmmqr_sparseqr_symbolic(ordering, tol, m, n, _ip(self._A[0]), _ip(self._A[1]), _dp(self._A[2]), _ip(Q), rp,
                                                C.byref(self._h)), "stmmqr_sparseqr_symbolic")
        else:
            _check(lib.stmmqr_sparseqr(ordering, tol, m, n, _ip(self._A[0]), _ip(self._A[1]), _dp(self._A[2]), _ip(Q), rp, device,
                                       C.byref(self._h)), "stmmqr_sparseqr")

    def numeric(self, device=-1):
        _check(lib.stmmqr_sparseqr_numeric(self._h, device), "stmmqr_sparseqr_numeric")

    def close(self):
        if self._h:
            lib.stmmqr_sparseqr_free(self._h)
            self._h = None

    __del__ = close

    @property
    def info(self) -> dict:
        v = np.zeros(12)
        _check(lib.stmmqr_sparseqr_info(self._h, _dp(v)), "stmmqr_sparseqr_info")
        keys = ["rank", "n1rows", "n1cols", "nf", "ana_seconds", "fac_seconds", "flops", "flop_bound", "ms_device", "ordering", "chunk_flag",
                "retries"]
        return dict(zip(keys, v.tolist()))

    @property
    def Q1fill(self):
        return np.ctypeslib.as_array(lib.stmmqr_sparseqr_q1fill(self._h), shape=(max(self.n, 1),))[:self.n].copy()

    def symbolic(self) -> dict:
        S = lib.stmmqr_sparseqr_symbolic_view(self._h).contents
        return _symbolic_dict(S, S.m, S.n)

    def Y(self):
        """(Yp, Yi, Yx) of the matrix handed to the numeric phase when singletons were removed, else None"""
        p, i, x = c_long_p(), c_long_p(), c_double_p()
        _check(lib.stmmqr_sparseqr_y(self._h, C.byref(p), C.byref(i), C.byref(x)), "stmmqr_sparseqr_y")
        if not p:
            return None
        n2 = self.n - int(self.info["n1cols"])
        Yp = np.ctypeslib.as_array(p, shape=(n2 + 1,)).copy()
        nz = int(Yp[-1])
        return Yp, np.ctypeslib.as_array(i, shape=(max(nz, 1),))[:nz].copy(), np.ctypeslib.as_array(x, shape=(max(nz, 1),))[:nz].copy()

    def export_r(self, econ=None, with_h=True) -> dict:
        """qr_rcount + qr_rconvert (SparseLQ.c:102-517) on the factors resident in HBM: R of the multifrontal part as CSC over
        the columns of the factorized matrix (Rp, Ri, Rx) and, with_h, H as CSC (Hp, Hi, Hx) + HTau."""
        Sptr = lib.stmmqr_sparseqr_symbolic_view(self._h)
        S = Sptr.contents
        econ = S.m if econ is None else econ
        Rp, Ri, Hp, Hi = c_long_p(), c_long_p(), c_long_p(), c_long_p()
        Rx, Hx, Ht = c_double_p(), c_double_p(), c_double_p()
        nh = C.c_long(0)
        _check(lib.stmmqr_plan_export_r(lib.stmmqr_sparseqr_plan(self._h), Sptr, econ, C.byref(Rp), C.byref(Ri), C.byref(Rx),
                                        C.byref(nh) if with_h else None, C.byref(Hp) if with_h else None, C.byref(Hi) if with_h else None,
                                        C.byref(Hx) if with_h else None, C.byref(Ht) if with_h else None), "stmmqr_plan_export_r")
        n = S.n
        out = {"Rp": _take(Rp, n + 1, I64)}
        out["Ri"], out["Rx"] = _take(Ri, int(out["Rp"][-1]), I64), _take(Rx, int(out["Rp"][-1]), np.float64)
        if with_h:
            out["Hp"] = _take(Hp, nh.value + 1, I64)
            hnz = int(out["Hp"][-1])
            out["Hi"], out["Hx"], out["HTau"] = _take(Hi, hnz, I64), _take(Hx, hnz, np.float64), _take(Ht, nh.value, np.float64)
        return out

    @classmethod
    def lq(cls, m, n, Ap, Ai, Ax, tol=-2.0, relax: Relax | None = None, device=-1):
        """SparseLQ (SparseLQ.c:691-734): the QR object of A' (L = R')."""
        self = cls.__new__(cls)
        self.m, self.n = int(n), int(m)
        self._A = (np.ascontiguousarray(Ap, I64), np.ascontiguousarray(Ai, I64), np.ascontiguousarray(Ax, np.float64))
        self._h = C.c_void_p()
        _check(lib.stmmqr_sparselq(7, tol, m, n, _ip(self._A[0]), _ip(self._A[1]), _dp(self._A[2]), None if relax is None else C.byref(relax),
                                   device, C.byref(self._h)), "stmmqr_sparselq")
        return self

    def qmult(self, method, X):
        X = np.asfortranarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1, order="F")
        Y = np.zeros_like(X, order="F")
        _check(lib.stmmqr_sparseqr_qmult(self._h, method, _dp(X), X.shape[0], X.shape[0], X.shape[1], _dp(Y), Y.shape[0]), "stmmqr_sparseqr_qmult")
        return Y

    def solve(self, system, B):
        B = np.asfortranarray(B, dtype=np.float64)
        if B.ndim == 1:
            B = B.reshape(-1, 1, order="F")
        rows = self.n if system <= 1 else self.m
        X = np.zeros((rows, B.shape[1]), order="F")
        _check(lib.stmmqr_sparseqr_solve(self._h, system, _dp(B), B.shape[0], B.shape[1], _dp(X), rows), "stmmqr_sparseqr_solve")
        return X


def read_matrix_market(path):
    """The driver's Matrix Market reader (SparseCore_read_matrix with prefer = 1, qrtest.c:112): (m, n, Ap, Ai, Ax) of the
    unsymmetric CSC with both triangles."""
    m, n, nz = C.c_long(0), C.c_long(0), C.c_long(0)
    pp, pi, px = c_long_p(), c_long_p(), c_double_p()
    lib.stmmqr_mm_last_error.restype = C.c_char_p
    rc = lib.stmmqr_read_matrix_market(str(path).encode(), C.byref(m), C.byref(n), C.byref(nz), C.byref(pp), C.byref(pi), C.byref(px))
    if rc != 0:
        raise StmmqrError(f"stmmqr_read_matrix_market({path}): {lib.stmmqr_mm_last_error().decode()}")
    try:
        Ap = np.ctypeslib.as_array(pp, shape=(n.value + 1,)).copy()
        Ai = np.ctypeslib.as_array(pi, shape=(max(nz.value, 1),))[:nz.value].copy()
        Ax = np.ctypeslib.as_array(px, shape=(max(nz.value, 1),))[:nz.value].copy()
    finally:
        lib.stmmqr_free.argtypes = [C.c_void_p]
        for q in (pp, pi, px):
            lib.stmmqr_free(C.cast(q, C.c_void_p))
    return m.value, n.value, Ap, Ai, Ax


def qr_fsize(f, Super, Rp, Rj, Sleft, Child, Childp, Cm, Fmap, Stair):
    """qr_fsize (SparseQR.h:159-176 / SparseQR_factorize.c:1066-1145): Fmap and Stair are written; returns fm"""
    a = [np.ascontiguousarray(x, I64) for x in (Super, Rp, Rj, Sleft, Child, Childp, Cm)]
    assert Fmap.dtype == I64 and Stair.dtype == I64
    lib.qr_fsize.restype = C.c_long
    return int(lib.qr_fsize(C.c_long(int(f)), *[_ip(x) for x in a], _ip(Fmap), _ip(Stair)))


def qr_stranspose2(m, n, Ap, Ai, Ax, Qfill, Sp, PLinv):
    """qr_stranspose2 (SparseQR_factorize.c:755-785): returns Sx, the values of S = A(P,Q) in row form"""
    Ap = np.ascontiguousarray(Ap, I64); Ai = np.ascontiguousarray(Ai, I64); Ax = np.ascontiguousarray(Ax, np.float64)
    Sp = np.ascontiguousarray(Sp, I64); PLinv = np.ascontiguousarray(PLinv, I64)
    Qf = None if Qfill is None else np.ascontiguousarray(Qfill, I64)
    A = SparseCsc(m, n, Ax.size, Ap.ctypes.data, Ai.ctypes.data, None, Ax.ctypes.data, None, 0, 2, 1, 0, 1, 1)
    Sx = np.zeros(max(Ax.size, 1)); W = np.zeros(max(m, 1), I64)
    lib.qr_stranspose2.restype = None
    lib.qr_stranspose2(C.byref(A), _ip(Qf), _ip(Sp), _ip(PLinv), _dp(Sx), _ip(W))
    return Sx[:Ax.size]


def qr_hpinv(sym: dict, Hm, Hr, Hii):
    """qr_hpinv (SparseQR_factorize.c:991-1060): Hii is rewritten in place; returns (HPinv, maxfm)"""
    keep = {k: np.ascontiguousarray(sym[k], I64) for k in ("Sleft", "Super", "Rp", "Hip", "PLinv")}
    S = QrSymbolicC()
    S.m, S.n, S.nf = int(sym["m"]), int(sym["n"]), int(sym["nf"])
    for k, a in keep.items():
        setattr(S, k, _ip(a))
    Hm = np.ascontiguousarray(Hm, I64); Hr = np.ascontiguousarray(Hr, I64)
    assert Hii.dtype == I64 and Hii.flags.c_contiguous
    HPinv = np.zeros(max(S.m, 1), I64); W = np.zeros(max(S.m, 1), I64)
    N = QrNumericC()
    N.m, N.n, N.nf = S.m, S.n, S.nf
    N.Hm, N.Hr, N.Hii, N.HPinv = _ip(Hm), _ip(Hr), _ip(Hii), _ip(HPinv)
    lib.qr_hpinv.restype = None
    lib.qr_hpinv(C.byref(S), C.byref(N), _ip(W))
    return HPinv[:S.m], int(N.maxfm)


def qr_assemble(f, fm, Super, Rp, Rj, Sp, Sj, Sleft, Child, Childp, Sx, Fmap, Cm, Cblocks: dict, Hr, Stair, Hii, Hip):
    """qr_assemble (SparseQR.h:178-200).  Cblocks: {child: packed C array}.  Returns (F, Cmap); Stair/Hii in place."""
    nf = len(Rp) - 1
    fn = int(Rp[f + 1] - Rp[f])
    F = np.zeros((fm, fn), order="F")
    Cmap = np.zeros(max(fn, 1), I64)
    ptrs = (c_double_p * (nf + 1))()
    keep = {}
    for c, a in Cblocks.items():
        keep[c] = np.ascontiguousarray(a, np.float64)
        ptrs[c] = _dp(keep[c])
    arrs = [np.ascontiguousarray(a, I64) for a in (Super, Rp, Rj, Sp, Sj, Sleft, Child, Childp)]
    Sx = np.ascontiguousarray(Sx, np.float64)
    Fmap = np.ascontiguousarray(Fmap, I64); Cm = np.ascontiguousarray(Cm, I64); Hr = np.ascontiguousarray(Hr, I64)
    Hip = np.ascontiguousarray(Hip, I64)
    lib.qr_assemble(f, fm, 1, *[_ip(a) for a in arrs], _dp(Sx), _ip(Fmap), _ip(Cm), ptrs, _ip(Hr), _ip(Stair), _ip(Hii),
                    _ip(Hip), _dp(F), _ip(Cmap))
    return F, Cmap
