"""ctypes binding of the C-ABI in include/proxqp_hip.h (libproxqp_hip.so).

`load()` only ever opens proxsuite_amd/csrc/libproxqp_hip.so -- the HIP build for gfx950 --
and raises if it is missing or if no HIP device is present: there is no CPU fallback in the
product path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from ._ctypes_defs import pqp_info, pqp_settings

PQP_STATS_COUNT = 33
STAT_NAMES = ("cyc_total", "cyc_scale", "cyc_factor_h", "cyc_zg", "cyc_schur", "cyc_kkt_solve",
              "cyc_residual", "cyc_linesearch", "cyc_global_res", "cyc_newton_misc", "n_newton",
              "n_schur_fact", "n_new_rows", "n_kkt_solves", "n_ls_breakpoints", "n_active_final",
              "cyc_f_load", "cyc_f_update", "cyc_f_panel", "cyc_f_writeback", "cyc_f_tinv", "cyc_s_gather",
              "cyc_solve_ldlt", "n_schur_blocked", "n_append", "n_delete", "bytes_engine", "n_refactorize", "cyc_ls_eval", "cyc_cert", "cyc_update", "wall_ticks", "flops_fact")

_DP = C.POINTER(C.c_double)
NAN = float("nan")


class NativeError(RuntimeError):
    pass


class _Tolerant:
    """prototype sink for NativeLib(legacy=True): attributes of symbols the library lacks are swallowed"""

    class _Sink:
        pass

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        try:
            return getattr(object.__getattribute__(self, "_lib"), name)
        except AttributeError:
            return _Tolerant._Sink()


class NativeLib:
    """Prototypes of every symbol include/proxqp_hip.h declares."""

    SYMBOLS = ("pqp_last_error", "pqp_device_count", "pqp_batch_create", "pqp_batch_destroy",
               "pqp_batch_size", "pqp_batch_dense_backend", "pqp_batch_settings", "pqp_batch_init",
               "pqp_batch_update", "pqp_batch_warm_start", "pqp_batch_cleanup", "pqp_batch_reset_qp", "pqp_batch_flush",
               "pqp_batch_solve", "pqp_batch_solve_range", "pqp_batch_solve_subset", "pqp_batch_copy_qp", "pqp_batch_set_stream", "pqp_batch_set_schedule", "pqp_batch_backward", "pqp_batch_backward_range",
               "pqp_batch_get_backward", "pqp_batch_get_results", "pqp_batch_result_device_ptrs", "pqp_batch_pack_results",
               "pqp_batch_get_scaled", "pqp_batch_get_schur_factor", "pqp_batch_get_stats", "pqp_batch_get_trace", "pqp_batch_last_solve_ms", "pqp_batch_last_prologue_ms",
               "pqp_batch_launch_config", "pqp_batch_solve_async", "pqp_batch_solve_range_async",
               "pqp_batch_solve_subset_async", "pqp_batch_wait", "pqp_batch_enable_host_results",
               "pqp_batch_host_results", "pqp_batch_host_results_fresh", "pqp_batch_own_stream", "pqp_batch_backward_subset",
               "pqp_multi_create", "pqp_multi_destroy", "pqp_multi_size", "pqp_multi_shard_count", "pqp_multi_shard",
               "pqp_multi_locate", "pqp_multi_settings", "pqp_multi_init", "pqp_multi_update", "pqp_multi_warm_start",
               "pqp_multi_cleanup", "pqp_multi_flush", "pqp_multi_solve", "pqp_multi_solve_range",
               "pqp_multi_solve_async", "pqp_multi_solve_range_async", "pqp_multi_wait", "pqp_multi_get_results",
               "pqp_multi_gather_device", "pqp_multi_get_trace", "pqp_multi_last_solve_ms", "pqp_box_calibrate",
               "pqp_batch_host_results_fresh_range")

    def __init__(self, path, legacy=False):
        """legacy=True (A/B scripts only): an older build of the library that lacks the newer entries can still be
        driven through the entries it has -- the product path (`load()`) binds every symbol or fails"""
        self.path = str(path)
        real = C.CDLL(self.path)
        L = _Tolerant(real) if legacy else real
        vp = C.c_void_p
        L.pqp_last_error.restype = C.c_char_p
        L.pqp_device_count.restype = C.c_int
        L.pqp_batch_create.argtypes = [C.c_int64] * 4 + [C.c_int] * 4 + [C.POINTER(vp)]
        L.pqp_batch_destroy.argtypes = [vp]
        L.pqp_batch_destroy.restype = None
        L.pqp_batch_size.argtypes = [vp]
        L.pqp_batch_size.restype = C.c_int64
        L.pqp_batch_dense_backend.argtypes = [vp]
        L.pqp_batch_settings.argtypes = [vp, C.c_int64]
        L.pqp_batch_settings.restype = C.POINTER(pqp_settings)
        for name in ("pqp_batch_init", "pqp_batch_update"):
            getattr(L, name).argtypes = [vp, C.c_int64] + [_DP] * 9 + [C.c_int] + [C.c_double] * 4
        L.pqp_batch_warm_start.argtypes = [vp, C.c_int64] + [_DP] * 3
        L.pqp_batch_cleanup.argtypes = [vp, C.c_int64]
        L.pqp_batch_reset_qp.argtypes = [vp, C.c_int64]
        L.pqp_batch_flush.argtypes = [vp]
        L.pqp_batch_solve.argtypes = [vp]
        L.pqp_batch_solve_range.argtypes = [vp, C.c_int64, C.c_int64]
        L.pqp_batch_solve_subset.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64]
        L.pqp_batch_copy_qp.argtypes = [vp, C.c_int64, vp, C.c_int64]
        L.pqp_batch_set_stream.argtypes = [vp, vp]
        L.pqp_batch_set_schedule.argtypes = [vp, C.c_int]
        L.pqp_batch_backward.argtypes = [vp, _DP] + [C.c_double] * 3
        L.pqp_batch_backward_range.argtypes = [vp, C.c_int64, C.c_int64, _DP] + [C.c_double] * 3
        L.pqp_batch_get_backward.argtypes = [vp, C.c_int64] + [_DP] * 7
        L.pqp_batch_get_results.argtypes = [vp, C.c_int64] + [_DP] * 5 + [C.POINTER(pqp_info)]
        L.pqp_batch_result_device_ptrs.argtypes = [vp] + [C.POINTER(_DP)] * 3
        L.pqp_batch_pack_results.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
        L.pqp_batch_get_scaled.argtypes = [vp, C.c_int64] + [_DP] * 9
        L.pqp_batch_get_stats.argtypes = [vp, C.POINTER(C.c_int64)]
        L.pqp_batch_get_trace.argtypes = [vp, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pqp_batch_get_schur_factor.argtypes = [vp, C.c_int64] + [_DP] * 3 + [C.POINTER(C.c_int32), C.POINTER(C.c_int64), _DP]
        L.pqp_batch_last_solve_ms.argtypes = [vp]
        L.pqp_batch_last_solve_ms.restype = C.c_double
        L.pqp_batch_last_prologue_ms.argtypes = [vp]
        L.pqp_batch_last_prologue_ms.restype = C.c_double
        L.pqp_batch_launch_config.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.pqp_batch_solve_async.argtypes = [vp]
        L.pqp_batch_solve_range_async.argtypes = [vp, C.c_int64, C.c_int64]
        L.pqp_batch_solve_subset_async.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64]
        L.pqp_batch_wait.argtypes = [vp]
        L.pqp_batch_enable_host_results.argtypes = [vp, C.c_int]
        L.pqp_batch_host_results.argtypes = [vp] + [C.POINTER(_DP)] * 5 + [C.POINTER(C.POINTER(pqp_info))]
        L.pqp_batch_host_results_fresh.argtypes = [vp, C.c_int64]
        L.pqp_batch_own_stream.argtypes = [vp]
        L.pqp_batch_backward_subset.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64, _DP] + [C.c_double] * 3
        L.pqp_multi_create.argtypes = [C.c_int64] * 4 + [C.c_int] * 3 + [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.pqp_multi_destroy.argtypes = [vp]
        L.pqp_multi_destroy.restype = None
        L.pqp_multi_size.argtypes = [vp]
        L.pqp_multi_size.restype = C.c_int64
        L.pqp_multi_shard_count.argtypes = [vp]
        L.pqp_multi_shard.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pqp_multi_locate.argtypes = [vp, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.pqp_multi_settings.argtypes = [vp, C.c_int64]
        L.pqp_multi_settings.restype = C.POINTER(pqp_settings)
        for name in ("pqp_multi_init", "pqp_multi_update"):
            getattr(L, name).argtypes = [vp, C.c_int64] + [_DP] * 9 + [C.c_int] + [C.c_double] * 4
        L.pqp_multi_warm_start.argtypes = [vp, C.c_int64] + [_DP] * 3
        L.pqp_multi_cleanup.argtypes = [vp, C.c_int64]
        for name in ("pqp_multi_flush", "pqp_multi_solve", "pqp_multi_solve_async", "pqp_multi_wait"):
            getattr(L, name).argtypes = [vp]
        for name in ("pqp_multi_solve_range", "pqp_multi_solve_range_async"):
            getattr(L, name).argtypes = [vp, C.c_int64, C.c_int64]
        L.pqp_multi_get_results.argtypes = [vp, C.c_int64] + [_DP] * 5 + [C.POINTER(pqp_info)]
        L.pqp_multi_gather_device.argtypes = [vp, C.c_int, vp]
        L.pqp_multi_last_solve_ms.argtypes = [vp]
        L.pqp_multi_last_solve_ms.restype = C.c_double
        L.pqp_multi_get_trace.argtypes = [vp, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pqp_box_calibrate.argtypes = [C.c_int, _DP, C.c_int]
        L.pqp_batch_host_results_fresh_range.argtypes = [vp, C.c_int64, C.c_int64]
        self.L = real

    def check(self, rc):
        if rc != 0:
            msg = self.L.pqp_last_error().decode(errors="replace")
            if rc == -1:
                raise ValueError(msg)  # the reference throws std::invalid_argument
            raise NativeError("libproxqp_hip error %d: %s" % (rc, msg))


_lib = None


def hip_library_path() -> Path:
    """In-tree build of the HIP library; PQP_HIP_LIBRARY selects another build of the SAME library
    (kernel experiments) -- it is never a fallback: the file must exist and a GPU must be there."""
    override = os.environ.get("PQP_HIP_LIBRARY")
    if override:
        return Path(override).resolve()
    return Path(__file__).resolve().parent / "csrc" / "libproxqp_hip.so"


def load() -> NativeLib:
    """The product path: the gfx950 HIP library, or a loud failure."""
    global _lib
    if _lib is None:
        p = hip_library_path()
        if not p.exists():
            raise NativeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7, and whichever
        # copy is loaded first serves every later library with that SONAME.  Loading torch's first
        # keeps torch.cuda usable next to this library (the reverse order leaves torch without GPUs).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = NativeLib(p)
        if lib.L.pqp_device_count() <= 0:
            raise NativeError("no HIP device visible: proxsuite_amd runs on MI355X only (no CPU fallback)")
        _lib = lib
    return _lib


def box_calibration(device: int = 0, lib: NativeLib = None) -> dict:
    """pqp_box_calibrate (include/proxqp_hip.h): what the box delivers right now -- HBM read rate, the time of a fixed
    latency-chain kernel at the C2 kernel's residency, the shader clock a fixed FMA chain implies.  ~60 ms of GPU time."""
    lib = lib or load()
    out = (C.c_double * 8)()
    rc = lib.L.pqp_box_calibrate(int(device), out, 8)
    if rc != 0:
        raise NativeError("pqp_box_calibrate: %s" % (lib.L.pqp_last_error() or b"").decode())
    return {"hbm_read_gbs": out[0], "chain_ms": out[1], "valu_ms": out[2], "sclk_mhz_est": out[3], "n_cu": int(out[4])}


def _as_array(a, shape, name):
    """numpy / torch (host or ROCm) -> (keepalive, pointer).  None or size 0 == nullopt."""
    if a is None:
        return None, None
    if hasattr(a, "data_ptr"):  # torch tensor: zero-copy when fp64 contiguous, host or device
        import torch
        if a.numel() == 0:
            return None, None
        t = a.detach()
        if t.dtype != torch.float64 or not t.is_contiguous():
            t = t.to(torch.float64).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("wrong argument size: %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
        return t, C.cast(t.data_ptr(), _DP)
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if arr.size == 0:
        return None, None
    if tuple(arr.shape) != tuple(shape):
        raise ValueError("wrong argument size: %s has shape %s, expected %s" % (name, arr.shape, tuple(shape)))
    return arr, arr.ctypes.data_as(_DP)


def _opt(v):
    return NAN if v is None else float(v)


class Batch:
    """Thin object wrapper over a pqp_batch handle (one device, B QPs of identical sizes)."""

    def __init__(self, batch_size, n, n_eq, n_in, box_constraints=False, hessian_type=1, dense_backend=0,
                 device=0, lib: NativeLib | None = None):
        self.lib = lib if lib is not None else load()
        self.B, self.n, self.n_eq, self.n_in = int(batch_size), int(n), int(n_eq), int(n_in)
        self.box = bool(box_constraints)
        self.n_c = self.n_in + (self.n if self.box else 0)
        h = C.c_void_p()
        self.lib.check(self.lib.L.pqp_batch_create(self.B, self.n, self.n_eq, self.n_in, int(self.box),
                                                   int(hessian_type), int(dense_backend), int(device),
                                                   C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.pqp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dense_backend(self):
        return self.lib.L.pqp_batch_dense_backend(self._h)

    def settings(self, idx) -> pqp_settings:
        p = self.lib.L.pqp_batch_settings(self._h, int(idx))
        if not p:
            raise IndexError(idx)
        return p.contents

    def settings_table(self):
        """All B settings records as ONE numpy structured array aliasing the handle's host memory (the records
        are contiguous: `pqp_batch_settings(h, i)` is &settings[i]).  `tab["eps_abs"][:] = 1e-9` writes the field
        of every QP at once -- what a per-QP Python loop costs milliseconds for at B = 2048."""
        p = self.lib.L.pqp_batch_settings(self._h, 0)
        arr = (pqp_settings * self.B).from_address(C.addressof(p.contents))
        return np.frombuffer(arr, dtype=np.dtype(pqp_settings))

    def set_all_settings(self, **fields):
        tab = self.settings_table()
        for k, v in fields.items():
            tab[k][:] = v

    def _shapes(self, idx):
        pre = (self.B,) if idx < 0 else ()
        n, ne, ni = self.n, self.n_eq, self.n_in
        return dict(H=pre + (n, n), g=pre + (n,), A=pre + (ne, n), b=pre + (ne,), C=pre + (ni, n),
                    l=pre + (ni,), u=pre + (ni,), l_box=pre + (n,), u_box=pre + (n,))

    def _setup(self, fn, idx, H, g, A, b, Cm, l, u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig):
        sh = self._shapes(idx)
        keep, ptrs = [], []
        for name, arr in (("H", H), ("g", g), ("A", A), ("b", b), ("C", Cm), ("l", l), ("u", u),
                          ("l_box", l_box), ("u_box", u_box)):
            k, p = _as_array(arr, sh[name], name)
            keep.append(k)
            ptrs.append(p)
        self.lib.check(fn(self._h, int(idx), *ptrs, int(bool(flag)), _opt(rho), _opt(mu_eq), _opt(mu_in),
                          _opt(min_eig)))

    def init(self, idx=-1, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
             compute_preconditioner=True, rho=None, mu_eq=None, mu_in=None, manual_minimal_H_eigenvalue=None):
        self._setup(self.lib.L.pqp_batch_init, idx, H, g, A, b, C, l, u, l_box, u_box,
                    compute_preconditioner, rho, mu_eq, mu_in, manual_minimal_H_eigenvalue)

    def update(self, idx=-1, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
               update_preconditioner=False, rho=None, mu_eq=None, mu_in=None,
               manual_minimal_H_eigenvalue=None):
        self._setup(self.lib.L.pqp_batch_update, idx, H, g, A, b, C, l, u, l_box, u_box,
                    update_preconditioner, rho, mu_eq, mu_in, manual_minimal_H_eigenvalue)

    def warm_start(self, idx=-1, x=None, y=None, z=None):
        pre = (self.B,) if idx < 0 else ()
        kx, px = _as_array(x, pre + (self.n,), "x")
        ky, py = _as_array(y, pre + (self.n_eq,), "y")
        kz, pz = _as_array(z, pre + (self.n_c,), "z")
        self.lib.check(self.lib.L.pqp_batch_warm_start(self._h, int(idx), px, py, pz))

    def cleanup(self, idx=-1):
        self.lib.check(self.lib.L.pqp_batch_cleanup(self._h, int(idx)))

    def flush(self):
        self.lib.check(self.lib.L.pqp_batch_flush(self._h))

    def solve(self, first=None, count=None):
        if first is None:
            self.lib.check(self.lib.L.pqp_batch_solve(self._h))
        else:
            self.lib.check(self.lib.L.pqp_batch_solve_range(self._h, int(first), int(1 if count is None else count)))

    def solve_async(self, first=None, count=None):
        """the launch is enqueued on the handle's stream; `wait()` (or any other call on the handle) completes it"""
        if first is None:
            self.lib.check(self.lib.L.pqp_batch_solve_async(self._h))
        else:
            self.lib.check(self.lib.L.pqp_batch_solve_range_async(self._h, int(first), int(1 if count is None else count)))

    def wait(self):
        self.lib.check(self.lib.L.pqp_batch_wait(self._h))

    def solve_subset(self, idx, asynchronous=False):
        """QP::solve() on the listed QPs in ONE launch (dense::solve_in_parallel(std::vector<QP>&))"""
        ii = np.ascontiguousarray(idx, dtype=np.int64)
        fn = self.lib.L.pqp_batch_solve_subset_async if asynchronous else self.lib.L.pqp_batch_solve_subset
        self.lib.check(fn(self._h, ii.ctypes.data_as(C.POINTER(C.c_int64)), len(ii)))

    def enable_host_results(self, on=True):
        """pinned host mirrors of (x, y, z, se, si, Info) written by the epilogue of the solve kernel: the results are
        on the host when solve() / wait() returns, with no device-to-host copy (include/proxqp_hip.h)"""
        self.lib.check(self.lib.L.pqp_batch_enable_host_results(self._h, int(bool(on))))

    def host_results(self):
        """numpy views (no copy) of the host mirrors: x [B, n], y [B, n_eq], z [B, n_c], se, si and the Info records
        as a structured array; contents are those of the last finished solve (see `host_results_fresh`)"""
        px = [_DP() for _ in range(5)]
        pi = C.POINTER(pqp_info)()
        self.lib.check(self.lib.L.pqp_batch_host_results(self._h, *[C.byref(p) for p in px], C.byref(pi)))
        shapes = ((self.B, self.n), (self.B, self.n_eq), (self.B, self.n_c), (self.B, self.n_eq), (self.B, self.n_c))
        out = []
        for p, sh in zip(px, shapes):
            cnt = sh[0] * sh[1]
            out.append(np.ctypeslib.as_array(p, shape=(cnt,)).reshape(sh) if cnt else np.zeros(sh))
        if self.B:
            arr = (pqp_info * self.B).from_address(C.addressof(pi.contents))
            info = np.frombuffer(arr, dtype=np.dtype(pqp_info))
        else:
            info = np.zeros(0, dtype=np.dtype(pqp_info))
        return (*out, info)

    def host_results_fresh(self, idx=-1, count=None):
        """the pinned mirrors hold the device's results of QP idx (-1: every QP); `count`: of the QPs idx .. idx+count-1"""
        if count is not None:
            return bool(self.lib.L.pqp_batch_host_results_fresh_range(self._h, int(idx), int(count)))
        return bool(self.lib.L.pqp_batch_host_results_fresh(self._h, int(idx)))

    def backward(self, loss_derivatives, eps=1e-4, rho_backward=1e-6, mu_backward=1e-6, first=None, count=None):
        """dense::compute_backward for the whole batch (or the QPs first .. first+count-1).
        `loss_derivatives`: [B or count, n + n_eq + n_in] (numpy or torch, host or device)."""
        ntot = self.n + self.n_eq + self.n_in
        rows = self.B if first is None else int(1 if count is None else count)
        k, p = _as_array(loss_derivatives, (rows, ntot), "loss_derivatives")
        if first is None:
            self.lib.check(self.lib.L.pqp_batch_backward(self._h, p, float(eps), float(rho_backward),
                                                         float(mu_backward)))
        else:
            self.lib.check(self.lib.L.pqp_batch_backward_range(self._h, int(first), rows, p, float(eps),
                                                               float(rho_backward), float(mu_backward)))

    def backward_results(self, idx=-1, into=None):
        """Model::backward_data as a dict of numpy arrays (or filled into the given tensors/arrays)."""
        pre = (self.B,) if idx < 0 else ()
        n, ne, ni = self.n, self.n_eq, self.n_in
        shapes = dict(dL_dH=pre + (n, n), dL_dg=pre + (n,), dL_dA=pre + (ne, n), dL_db=pre + (ne,),
                      dL_dC=pre + (ni, n), dL_du=pre + (ni,), dL_dl=pre + (ni,))
        out = into if into is not None else {k: np.zeros(v) for k, v in shapes.items()}
        ptrs = []
        for name in ("dL_dH", "dL_dg", "dL_dA", "dL_db", "dL_dC", "dL_du", "dL_dl"):
            buf = out.get(name)
            if buf is None or (hasattr(buf, "numel") and buf.numel() == 0) or (
                    not hasattr(buf, "numel") and buf.size == 0):
                ptrs.append(None)
            elif hasattr(buf, "data_ptr"):
                import torch
                if buf.dtype != torch.float64 or not buf.is_contiguous() or tuple(buf.shape) != shapes[name]:
                    raise ValueError("backward_results: %s must be contiguous float64 of shape %s" % (name, shapes[name]))
                ptrs.append(C.cast(buf.data_ptr(), _DP))
            else:
                ptrs.append(buf.ctypes.data_as(_DP))
        self.lib.check(self.lib.L.pqp_batch_get_backward(self._h, int(idx), *ptrs))
        return out

    def pack_results(self, out, first=0, count=None, stream=None):
        """(x, y, z, status, iter) of QPs [first, first+count) -> rows of the fp64 buffer `out`
        ([count][n + n_eq + n_c + 2], DEVICE memory: a torch ROCm tensor or a raw pointer), by a
        device kernel on `stream` (int hipStream_t or None).  Not synchronised."""
        count = self.B - first if count is None else count
        ptr = out.data_ptr() if hasattr(out, "data_ptr") else (out.ctypes.data if hasattr(out, "ctypes") else int(out))
        self.lib.check(self.lib.L.pqp_batch_pack_results(self._h, int(first), int(count), C.c_void_p(ptr),
                                                         C.c_void_p(int(stream) if stream else None)))
        return out

    def set_schedule(self, longest_first=True):
        """Dispatch order of whole-batch solves: index order (default) or longest-processing-time first
        (uses the device cycle counts of the previous whole-batch solve of this handle)."""
        self.lib.check(self.lib.L.pqp_batch_set_schedule(self._h, int(bool(longest_first))))

    def own_stream(self):
        """a non-blocking stream of the handle's own: handles with their own streams overlap on one device"""
        self.lib.check(self.lib.L.pqp_batch_own_stream(self._h))

    def backward_subset(self, idx, loss_derivatives, eps=1e-4, rho_backward=1e-6, mu_backward=1e-6):
        """compute_backward on the QPs idx[0..] in one launch; row i of loss_derivatives belongs to QP idx[i]"""
        ntot = self.n + self.n_eq + self.n_in
        ii = np.ascontiguousarray(idx, dtype=np.int64)
        k, p = _as_array(loss_derivatives, (len(ii), ntot), "loss_derivatives")
        self.lib.check(self.lib.L.pqp_batch_backward_subset(self._h, ii.ctypes.data_as(C.POINTER(C.c_int64)), len(ii), p,
                                                            float(eps), float(rho_backward), float(mu_backward)))

    def set_stream(self, stream):
        """`stream`: a hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        self.lib.check(self.lib.L.pqp_batch_set_stream(self._h, C.c_void_p(int(stream) if stream else None)))

    @property
    def last_solve_ms(self):
        return self.lib.L.pqp_batch_last_solve_ms(self._h)

    @property
    def last_prologue_ms(self):
        """the part of last_solve_ms the 256-thread factorisation prologue took when the launch went to the one-wavefront
        dense kernel (two kernels per launch); 0 otherwise"""
        return self.lib.L.pqp_batch_last_prologue_ms(self._h)

    def results(self, idx=-1):
        pre = (self.B,) if idx < 0 else ()
        x = np.zeros(pre + (self.n,))
        y = np.zeros(pre + (self.n_eq,))
        z = np.zeros(pre + (self.n_c,))
        se = np.zeros(pre + (self.n_eq,))
        si = np.zeros(pre + (self.n_c,))
        info = (pqp_info * self.B)() if idx < 0 else pqp_info()
        p = lambda a: a.ctypes.data_as(_DP)
        ip = C.cast(info, C.POINTER(pqp_info)) if idx < 0 else C.byref(info)
        self.lib.check(self.lib.L.pqp_batch_get_results(self._h, int(idx), p(x), p(y), p(z), p(se), p(si), ip))
        return x, y, z, se, si, info

    def results_into(self, x=None, y=None, z=None, se=None, si=None, idx=-1):
        """Copy results straight into caller buffers (numpy arrays or torch tensors, host or
        ROCm device; fp64 contiguous).  A device tensor makes this a device-to-device copy."""
        pre = (self.B,) if idx < 0 else ()
        want = dict(x=pre + (self.n,), y=pre + (self.n_eq,), z=pre + (self.n_c,), se=pre + (self.n_eq,),
                    si=pre + (self.n_c,))
        ptrs = []
        for name, buf in (("x", x), ("y", y), ("z", z), ("se", se), ("si", si)):
            if buf is None or (hasattr(buf, "numel") and buf.numel() == 0) or (hasattr(buf, "size") and not callable(buf.size) and buf.size == 0):
                ptrs.append(None)
                continue
            if hasattr(buf, "data_ptr"):
                import torch
                if buf.dtype != torch.float64 or not buf.is_contiguous() or tuple(buf.shape) != want[name]:
                    raise ValueError("results_into: %s must be a contiguous float64 tensor of shape %s" % (name, want[name]))
                ptrs.append(C.cast(buf.data_ptr(), _DP))
            else:
                if buf.dtype != np.float64 or not buf.flags.c_contiguous or tuple(buf.shape) != want[name]:
                    raise ValueError("results_into: %s must be a C-contiguous float64 array of shape %s" % (name, want[name]))
                ptrs.append(buf.ctypes.data_as(_DP))
        self.lib.check(self.lib.L.pqp_batch_get_results(self._h, int(idx), *ptrs, None))

    def infos(self):
        info = (pqp_info * self.B)()
        self.lib.check(self.lib.L.pqp_batch_get_results(self._h, -1, None, None, None, None, None,
                                                        C.cast(info, C.POINTER(pqp_info))))
        return info

    def scaled(self, idx):
        n, ne, ni = self.n, self.n_eq, self.n_in
        out = dict(H=np.zeros((n, n)), g=np.zeros(n), A=np.zeros((ne, n)), b=np.zeros(ne), C=np.zeros((ni, n)),
                   l=np.zeros(ni), u=np.zeros(ni), delta=np.zeros(n + ne + self.n_c))
        c = C.c_double(0)
        p = lambda a: a.ctypes.data_as(_DP)
        self.lib.check(self.lib.L.pqp_batch_get_scaled(
            self._h, int(idx), p(out["H"]), p(out["g"]), p(out["A"]), p(out["b"]), p(out["C"]), p(out["l"]),
            p(out["u"]), p(out["delta"]), C.cast(C.byref(c), _DP)))
        out["c"] = c.value
        return out

    def schur_factor(self, idx):
        """diagnostic: (W_S, D_S, G, slots, meta, mus) of QP `idx` after its last solve (see proxqp_hip.h)"""
        nd, nc = self.n_eq + self.n_c, self.n_c
        WS, dS, G = np.zeros((nd, nd)), np.zeros(nd), np.zeros((nd, nd))
        slots, meta, mus = np.zeros(max(nc, 1), dtype=np.int32), np.zeros(4, dtype=np.int64), np.zeros(2)
        self.lib.check(self.lib.L.pqp_batch_get_schur_factor(
            self._h, idx, WS.ctypes.data_as(_DP), dS.ctypes.data_as(_DP), G.ctypes.data_as(_DP),
            slots.ctypes.data_as(C.POINTER(C.c_int32)), meta.ctypes.data_as(C.POINTER(C.c_int64)), mus.ctypes.data_as(_DP)))
        return WS, dS, G, slots[:nc], dict(zip(("n_slots", "n_c", "ls_valid", "ls_edited"), meta.tolist())), mus

    def stats(self):
        a = np.zeros((self.B, PQP_STATS_COUNT), dtype=np.int64)
        self.lib.check(self.lib.L.pqp_batch_get_stats(self._h, a.ctypes.data_as(C.POINTER(C.c_int64))))
        return a

    def trace(self, idx):
        """settings.verbose: the per-iteration lines of QP idx from the last launch as an (N, 8) array -- rows
        [1, k, pri_res, dua_res, duality_gap, mu_in, rho, 0] (outer iteration k) and [2, k, inner residual, alpha,
        0, 0, 0, 0] (inner iteration k), in the order the reference prints them (pqp_batch_get_trace)."""
        n = C.c_int64(0)
        self.lib.check(self.lib.L.pqp_batch_get_trace(self._h, int(idx), None, 0, C.byref(n)))
        out = np.zeros((n.value, 8))
        if n.value:
            self.lib.check(self.lib.L.pqp_batch_get_trace(self._h, int(idx), out.ctypes.data, n.value, C.byref(n)))
        return out

    def launch_config(self):
        t = C.c_int(0)
        b = C.c_int64(0)
        self.lib.check(self.lib.L.pqp_batch_launch_config(self._h, C.byref(t), C.byref(b)))
        return t.value, b.value

    def result_device_ptrs(self):
        x, y, z = _DP(), _DP(), _DP()
        self.lib.check(self.lib.L.pqp_batch_result_device_ptrs(self._h, C.byref(x), C.byref(y), C.byref(z)))
        return tuple(C.cast(p, C.c_void_p).value for p in (x, y, z))



class MultiBatch:
    """One batch over several GPUs of the node from ONE process (pqp_multi_*, include/proxqp_hip.h): shard g is an
    ordinary batch on devices[g] holding a contiguous range of the QPs; a solve launches every shard and then waits
    for all of them.  `devices` may repeat an ordinal (logical shards on one GPU)."""

    def __init__(self, batch_size, n, n_eq, n_in, devices, box_constraints=False, hessian_type=1, dense_backend=0,
                 lib: NativeLib | None = None):
        self.lib = lib if lib is not None else load()
        self.B, self.n, self.n_eq, self.n_in = int(batch_size), int(n), int(n_eq), int(n_in)
        self.box = bool(box_constraints)
        self.n_c = self.n_in + (self.n if self.box else 0)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        self.lib.check(self.lib.L.pqp_multi_create(self.B, self.n, self.n_eq, self.n_in, int(self.box), int(hessian_type),
                                                   int(dense_backend), devs, len(devices), C.byref(h)))
        self._h = h
        self.devices = [int(d) for d in devices]

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.pqp_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shard_count(self):
        return self.lib.L.pqp_multi_shard_count(self._h)

    def shard(self, g):
        """(first, count) of shard g"""
        f, c = C.c_int64(0), C.c_int64(0)
        self.lib.check(self.lib.L.pqp_multi_shard(self._h, int(g), None, C.byref(f), C.byref(c)))
        return f.value, c.value

    def settings(self, idx) -> pqp_settings:
        p = self.lib.L.pqp_multi_settings(self._h, int(idx))
        if not p:
            raise IndexError(idx)
        return p.contents

    _shapes = Batch._shapes

    def _setup(self, fn, idx, H, g, A, b, Cm, l, u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig):
        Batch._setup(self, fn, idx, H, g, A, b, Cm, l, u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig)

    def init(self, idx=-1, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
             compute_preconditioner=True, rho=None, mu_eq=None, mu_in=None, manual_minimal_H_eigenvalue=None):
        self._setup(self.lib.L.pqp_multi_init, idx, H, g, A, b, C, l, u, l_box, u_box, compute_preconditioner, rho, mu_eq,
                    mu_in, manual_minimal_H_eigenvalue)

    def update(self, idx=-1, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
               update_preconditioner=False, rho=None, mu_eq=None, mu_in=None, manual_minimal_H_eigenvalue=None):
        self._setup(self.lib.L.pqp_multi_update, idx, H, g, A, b, C, l, u, l_box, u_box, update_preconditioner, rho,
                    mu_eq, mu_in, manual_minimal_H_eigenvalue)

    def warm_start(self, idx=-1, x=None, y=None, z=None):
        pre = (self.B,) if idx < 0 else ()
        kx, px = _as_array(x, pre + (self.n,), "x")
        ky, py = _as_array(y, pre + (self.n_eq,), "y")
        kz, pz = _as_array(z, pre + (self.n_c,), "z")
        self.lib.check(self.lib.L.pqp_multi_warm_start(self._h, int(idx), px, py, pz))

    def cleanup(self, idx=-1):
        self.lib.check(self.lib.L.pqp_multi_cleanup(self._h, int(idx)))

    def flush(self):
        self.lib.check(self.lib.L.pqp_multi_flush(self._h))

    def solve(self, first=None, count=None):
        if first is None:
            self.lib.check(self.lib.L.pqp_multi_solve(self._h))
        else:
            self.lib.check(self.lib.L.pqp_multi_solve_range(self._h, int(first), int(1 if count is None else count)))

    def solve_async(self):
        self.lib.check(self.lib.L.pqp_multi_solve_async(self._h))

    def wait(self):
        self.lib.check(self.lib.L.pqp_multi_wait(self._h))

    @property
    def last_solve_ms(self):
        return self.lib.L.pqp_multi_last_solve_ms(self._h)

    def trace(self, idx):
        """settings.verbose: the per-iteration lines of QP idx from the last launch (Batch.trace, pqp_multi_get_trace)"""
        n = C.c_int64(0)
        self.lib.check(self.lib.L.pqp_multi_get_trace(self._h, int(idx), None, 0, C.byref(n)))
        out = np.zeros((n.value, 8))
        if n.value:
            self.lib.check(self.lib.L.pqp_multi_get_trace(self._h, int(idx), out.ctypes.data, n.value, C.byref(n)))
        return out

    def results(self, idx=-1):
        pre = (self.B,) if idx < 0 else ()
        x = np.zeros(pre + (self.n,))
        y = np.zeros(pre + (self.n_eq,))
        z = np.zeros(pre + (self.n_c,))
        se = np.zeros(pre + (self.n_eq,))
        si = np.zeros(pre + (self.n_c,))
        info = (pqp_info * self.B)() if idx < 0 else pqp_info()
        p = lambda a: a.ctypes.data_as(_DP)
        ip = C.cast(info, C.POINTER(pqp_info)) if idx < 0 else C.byref(info)
        self.lib.check(self.lib.L.pqp_multi_get_results(self._h, int(idx), p(x), p(y), p(z), p(se), p(si), ip))
        return x, y, z, se, si, info

    def gather_device(self, out, root_shard=0):
        """(x, y, z, status, iter) of every QP -> `out` ([B][n + n_eq + n_c + 2] fp64, memory of the root shard's device:
        a torch ROCm tensor or a raw pointer): pack kernel per shard + peer copies"""
        ptr = out.data_ptr() if hasattr(out, "data_ptr") else (out.ctypes.data if hasattr(out, "ctypes") else int(out))
        self.lib.check(self.lib.L.pqp_multi_gather_device(self._h, int(root_shard), C.c_void_p(ptr)))
        return out
