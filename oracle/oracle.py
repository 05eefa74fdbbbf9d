"""TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so, the CPU restatement of the reference's dense
ProxQP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (proxsuite_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import math
import subprocess
import sys
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(_HERE.parent))
from proxsuite_amd._ctypes_defs import (  # noqa: E402  (shared POD layout only)
    DenseBackend, HessianType, InitialGuess, QPSolverOutput, pqp_info, pqp_settings)

_lib = None
NAN = float("nan")


def build(force: bool = False) -> Path:
    lib = _HERE / "liboracle.so"
    srcs = [_HERE / f for f in ("proxqp_oracle.cpp", "oracle_capi.cpp", "proxqp_oracle.hpp",
                                "ldlt_oracle.hpp", "Makefile")] + [_HERE.parent / "include" / "pqp_types.h"]
    stale = (not lib.exists()) or any(s.stat().st_mtime > lib.stat().st_mtime for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return lib


def lib():
    global _lib
    if _lib is None:
        path = _HERE / "liboracle.so"
        if not path.exists() or Path("/usr/bin/make").exists():
            try:
                build()
            except Exception:
                if not path.exists():
                    raise
        L = C.CDLL(str(path))
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.pqo_create.restype = vp
        L.pqo_create.argtypes = [C.c_int64] * 3 + [C.c_int] * 3
        L.pqo_destroy.argtypes = [vp]
        L.pqo_settings.restype = C.POINTER(pqp_settings)
        L.pqo_settings.argtypes = [vp]
        L.pqo_info.restype = C.POINTER(pqp_info)
        L.pqo_info.argtypes = [vp]
        L.pqo_trace.restype = C.c_int64
        L.pqo_trace.argtypes = [vp, C.c_void_p, C.c_int64]
        L.pqo_dense_backend.argtypes = [vp]
        L.pqo_init.argtypes = [vp] + [dp] * 9 + [C.c_int] + [C.c_double] * 4
        L.pqo_update.argtypes = [vp] + [dp] * 9 + [C.c_int] + [C.c_double] * 4
        L.pqo_solve.argtypes = [vp] + [dp] * 3
        L.pqo_cleanup.argtypes = [vp]
        L.pqo_compute_backward.argtypes = [vp, dp] + [C.c_double] * 3 + [dp] * 7
        L.pqo_get_results.argtypes = [vp] + [dp] * 5 + [C.POINTER(pqp_info)]
        L.pqo_get_scaled.argtypes = [vp] + [dp] * 9
        L.pqo_get_counters.argtypes = [vp, dp]
        L.pqo_solve_in_parallel.argtypes = [C.POINTER(vp), C.c_int64, C.c_int]
        L.pqo_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _arr(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.size == 0:
        return None
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("wrong argument size: got %s expected %s" % (a.shape, shape))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _opt(v):
    return NAN if v is None else float(v)


class _Results:
    pass


class QP:
    """Mirror of proxsuite::proxqp::dense::QP<double> (reference dense/wrapper.hpp:114-963)."""

    def __init__(self, n, n_eq, n_in, box_constraints=False, hessian_type=HessianType.Dense,
                 dense_backend=DenseBackend.Automatic):
        self._L = lib()
        self.n, self.n_eq, self.n_in, self.box = int(n), int(n_eq), int(n_in), bool(box_constraints)
        self._h = self._L.pqo_create(n, n_eq, n_in, int(self.box), int(hessian_type), int(dense_backend))
        if not self._h:
            raise ValueError(self._L.pqo_last_error().decode())
        self.settings = self._L.pqo_settings(self._h).contents
        self.results = _Results()
        self.results.info = self._L.pqo_info(self._h).contents
        self._sync()

    def __del__(self):
        try:
            if self._h:
                self._L.pqo_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def n_c(self):
        return self.n_in + (self.n if self.box else 0)

    def _mats(self, H, g, A, b, Cm, l, u, l_box, u_box):
        n, ne, ni = self.n, self.n_eq, self.n_in
        keep = [_arr(H, (n, n)), _arr(g, (n,)), _arr(A, (ne, n)), _arr(b, (ne,)), _arr(Cm, (ni, n)),
                _arr(l, (ni,)), _arr(u, (ni,)), _arr(l_box, (n,)), _arr(u_box, (n,))]
        return keep

    def init(self, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
             compute_preconditioner=True, rho=None, mu_eq=None, mu_in=None,
             manual_minimal_H_eigenvalue=None):
        keep = self._mats(H, g, A, b, C, l, u, l_box, u_box)
        self._L.pqo_init(self._h, *map(_ptr, keep), int(compute_preconditioner), _opt(rho), _opt(mu_eq),
                         _opt(mu_in), _opt(manual_minimal_H_eigenvalue))
        self._sync()

    def update(self, H=None, g=None, A=None, b=None, C=None, l=None, u=None, l_box=None, u_box=None,
               update_preconditioner=False, rho=None, mu_eq=None, mu_in=None,
               manual_minimal_H_eigenvalue=None):
        keep = self._mats(H, g, A, b, C, l, u, l_box, u_box)
        self._L.pqo_update(self._h, *map(_ptr, keep), int(update_preconditioner), _opt(rho), _opt(mu_eq),
                           _opt(mu_in), _opt(manual_minimal_H_eigenvalue))
        self._sync()

    def solve(self, x=None, y=None, z=None):
        keep = [_arr(x, (self.n,)), _arr(y, (self.n_eq,)), _arr(z, (self.n_c,))]
        self._L.pqo_solve(self._h, *map(_ptr, keep))
        self._sync()

    def trace(self):
        """settings.verbose: the per-iteration lines of the last solve as an (N, 8) array -- outer iterations
        [1, k, pri_res, dua_res, duality_gap, mu_in, rho, 0] (reference solver.hpp:1478-1485), inner ones
        [2, k, inner residual, alpha, 0, 0, 0, 0] (solver.hpp:1021-1027), in the order the reference prints them."""
        n = int(self._L.pqo_trace(self._h, None, 0))
        out = np.zeros((n, 8))
        if n:
            self._L.pqo_trace(self._h, out.ctypes.data, n)
        return out

    def compute_backward(self, loss_derivative, eps=1e-4, rho_backward=1e-6, mu_backward=1e-6):
        """dense::compute_backward (reference dense/compute_ECJ.hpp:29-189).  Returns a dict with
        dL_dH, dL_dg, dL_dA, dL_db, dL_dC, dL_du, dL_dl (the reference's model.backward_data)."""
        n, ne, ni = self.n, self.n_eq, self.n_in
        ld = _arr(loss_derivative, (n + ne + ni,))
        out = dict(dL_dH=np.zeros((n, n)), dL_dg=np.zeros(n), dL_dA=np.zeros((ne, n)), dL_db=np.zeros(ne),
                   dL_dC=np.zeros((ni, n)), dL_du=np.zeros(ni), dL_dl=np.zeros(ni))
        rc = self._L.pqo_compute_backward(self._h, _ptr(ld), float(eps), float(rho_backward), float(mu_backward),
                                          *[_ptr(out[k]) for k in ("dL_dH", "dL_dg", "dL_dA", "dL_db", "dL_dC",
                                                                   "dL_du", "dL_dl")])
        if rc != 0:
            raise ValueError(self._L.pqo_last_error().decode())
        self._sync()
        return out

    def cleanup(self):
        self._L.pqo_cleanup(self._h)
        self._sync()

    def _sync(self):
        r = self.results
        r.x = np.zeros(self.n)
        r.y = np.zeros(self.n_eq)
        r.z = np.zeros(self.n_c)
        r.se = np.zeros(self.n_eq)
        r.si = np.zeros(self.n_c)
        self._L.pqo_get_results(self._h, _ptr(r.x), _ptr(r.y), _ptr(r.z), _ptr(r.se), _ptr(r.si), None)

    def scaled(self):
        n, ne, ni = self.n, self.n_eq, self.n_in
        out = dict(H=np.zeros((n, n)), g=np.zeros(n), A=np.zeros((ne, n)), b=np.zeros(ne),
                   C=np.zeros((ni, n)), l=np.zeros(ni), u=np.zeros(ni), delta=np.zeros(n + ne + self.n_c))
        c = C.c_double(0)
        self._L.pqo_get_scaled(self._h, _ptr(out["H"]), _ptr(out["g"]), _ptr(out["A"]), _ptr(out["b"]),
                               _ptr(out["C"]), _ptr(out["l"]), _ptr(out["u"]), _ptr(out["delta"]),
                               C.cast(C.byref(c), C.POINTER(C.c_double)))
        out["c"] = c.value
        return out

    def counters(self):
        a = np.zeros(9)
        self._L.pqo_get_counters(self._h, _ptr(a))
        keys = ("fact_flops", "fact_bytes", "level2_flops", "n_solves", "n_residuals", "n_ls_evals",
                "n_inserted", "n_deleted", "n_refactorize")
        return dict(zip(keys, a.tolist()))


def solve_in_parallel(qps, num_threads=None):
    """reference parallel/qp_solve.hpp:17-59 restated (OpenMP dynamic schedule)."""
    L = lib()
    arr = (C.c_void_p * len(qps))(*[q._h for q in qps])
    nt = L.pqo_solve_in_parallel(arr, len(qps), 0 if num_threads is None else int(num_threads))
    for q in qps:
        q._sync()
    return nt


def kkt_residuals(H, g, A, b, Cm, l, u, x, y, z, l_box=None, u_box=None):
    """The reference's universal acceptance test (test/src/dense_qp_with_eq_and_in.cpp:46-56),
    recomputed on the unscaled model by independent numpy code."""
    n_in = Cm.shape[0] if Cm is not None and Cm.size else 0
    pri = 0.0
    if A is not None and A.size:
        pri = max(pri, float(np.max(np.abs(A @ x - b))))
    zc = z[:n_in]
    dua = H @ x + g
    if A is not None and A.size:
        dua = dua + A.T @ y
    if n_in:
        Cx = Cm @ x
        viol = np.maximum(Cx - u, 0) + np.minimum(Cx - l, 0)
        pri = max(pri, float(np.max(np.abs(viol))))
        dua = dua + Cm.T @ zc
    if l_box is not None:
        viol = np.maximum(x - u_box, 0) + np.minimum(x - l_box, 0)
        pri = max(pri, float(np.max(np.abs(viol))))
        dua = dua + z[n_in:]
    return pri, float(np.max(np.abs(dua)))
