"""`proxsuite.proxqp.dense` rebuilt on the MI355X batch solver (libproxqp_hip.so).

Same names, argument order, defaults and error behaviour as the reference's nanobind module
(reference bindings/python/src/expose-qpobject.hpp:60-230, expose-qpvector.hpp:22-39,
expose-parallel.hpp:27-46, expose-solve.hpp) for the dense ProxQP path:

    QP(n, n_eq, n_in, box_constraints=False, hessian_type=Dense, dense_backend=Automatic)
        .init(...) .update(...) .solve([x, y, z]) .cleanup() .settings .results .model
    BatchQP(batch_size) .init_qp_in_place(n, n_eq, n_in) .insert(qp) .get(i) .size()
    VectorQP()          .append(qp)
    solve_in_parallel(qps, num_threads=None)
    solve(H, g, A, b, C, l, u, ...)
    compute_backward(qp, loss_derivative, eps, rho_backward, mu_backward)   -> qp.model.backward_data
    solve_backward_in_parallel(num_threads, qps, loss_derivatives, ...)
    estimate_minimal_eigen_value_of_symmetric_matrix(H, ...)               (host-side helper)

What differs is where the work happens.  A `BatchQP` owns device-resident *pools*: one
C-ABI batch handle per problem signature (n, n_eq, n_in, box, Hessian type, backend), sized
`batch_size`; `init_qp_in_place` hands out a `QP` that is a (pool, slot) view, and
`solve_in_parallel` is ONE kernel launch per pool with one workgroup per QP.  A standalone
`QP` is a pool of one.  Every numerical step runs in the HIP kernels; this file only moves
arguments.  There is no CPU fallback: constructing a QP without the HIP library or without
a GPU raises.
"""
from __future__ import annotations

import numpy as np

from .. import _native
from .._ctypes_defs import (DenseBackend, EigenValueEstimateMethodOption, HessianType, InitialGuess,
                            MeritFunctionType, QPSolverOutput, pqp_info, pqp_settings)

__all__ = ["QP", "BatchQP", "VectorQP", "VectorLossDerivatives", "solve_in_parallel", "solve", "solve_no_gil",
           "compute_backward", "solve_backward_in_parallel", "estimate_minimal_eigen_value_of_symmetric_matrix",
           "EigenValueEstimateMethodOption", "DenseBackend", "HessianType", "InitialGuess",
           "QPSolverOutput", "MeritFunctionType", "Settings", "Results", "Info", "Model", "BackwardData"]

# (nanobind `.export_values()`, bindings/python/src/expose-qpobject.hpp:28-38: proxsuite.proxqp.dense.PrimalDualLDLT,
# proxsuite.proxqp.dense.Diagonal, ... exist beside the enum classes)
for _e in (DenseBackend, HessianType):
    globals().update(_e.__members__)
del _e

_BOOL_SETTINGS = ("verbose", "update_preconditioner", "compute_preconditioner", "compute_timings",
                  "check_duality_gap", "bcl_update", "primal_infeasibility_solving")
_ENUM_SETTINGS = {"initial_guess": InitialGuess, "merit_function_type": MeritFunctionType}
_SETTING_NAMES = tuple(n for n, _ in pqp_settings._fields_ if n != "_pad")
_INFO_NAMES = tuple(n for n, _ in pqp_info._fields_ if n != "_pad")


class Settings:
    """`qp.settings` (reference settings.hpp:95-315): a live view of the host-side
    pqp_settings record of one QP; it is re-read at every solve, as in the reference."""

    __slots__ = ("_s",)

    def __init__(self, raw):
        object.__setattr__(self, "_s", raw)

    def __getattr__(self, name):
        if name not in _SETTING_NAMES:
            raise AttributeError(name)
        v = getattr(self._s, name)
        if name in _BOOL_SETTINGS:
            return bool(v)
        if name in _ENUM_SETTINGS:
            return _ENUM_SETTINGS[name](v)
        return v

    def __setattr__(self, name, value):
        if name not in _SETTING_NAMES:
            raise AttributeError("Settings has no field %r" % name)
        setattr(self._s, name, int(value) if (name in _BOOL_SETTINGS or name in _ENUM_SETTINGS) else value)

    def __repr__(self):
        return "Settings(" + ", ".join("%s=%r" % (n, getattr(self, n)) for n in _SETTING_NAMES) + ")"


class Info:
    """`qp.results.info` (reference results.hpp:27-74)."""

    __slots__ = ("_i",)

    def __init__(self, raw):
        object.__setattr__(self, "_i", raw)

    def __getattr__(self, name):
        if name in ("sparse_backend",):
            raise AttributeError("sparse_backend belongs to the sparse solver")
        if name not in _INFO_NAMES:
            raise AttributeError(name)
        v = getattr(self._i, name)
        return QPSolverOutput(v) if name == "status" else v

    def __repr__(self):
        return "Info(" + ", ".join("%s=%r" % (n, getattr(self, n)) for n in _INFO_NAMES) + ")"


class Results:
    """`qp.results` (reference results.hpp:76-204): x, y, z, se, si, info."""

    __slots__ = ("x", "y", "z", "se", "si", "info")

    def __init__(self, x, y, z, se, si, info):
        self.x, self.y, self.z, self.se, self.si, self.info = x, y, z, se, si, Info(info)


class BackwardData:
    """`qp.model.backward_data` (reference dense/backward_data.hpp:27-133): the jacobians of the loss
    wrt the model, filled by compute_backward / solve_backward_in_parallel."""

    __slots__ = ("dL_dH", "dL_dg", "dL_dA", "dL_db", "dL_dC", "dL_du", "dL_dl")

    def __init__(self, dim, n_eq, n_in):
        self.dL_dH = np.zeros((dim, dim))
        self.dL_dg = np.zeros(dim)
        self.dL_dA = np.zeros((n_eq, dim))
        self.dL_db = np.zeros(n_eq)
        self.dL_dC = np.zeros((n_in, dim))
        self.dL_du = np.zeros(n_in)
        self.dL_dl = np.zeros(n_in)


class Model:
    """`qp.model` (reference dense/model.hpp:24-63): the last data handed to init/update."""

    def __init__(self, dim, n_eq, n_in, box):
        self.dim, self.n_eq, self.n_in = dim, n_eq, n_in
        self.backward_data = BackwardData(dim, n_eq, n_in)
        self.n_total = dim + n_eq + n_in
        self.H = np.zeros((dim, dim))
        self.g = np.zeros(dim)
        self.A = np.zeros((n_eq, dim))
        self.b = np.zeros(n_eq)
        self.C = np.zeros((n_in, dim))
        self.l = np.zeros(n_in)
        self.u = np.zeros(n_in)
        if box:
            self.l_box = np.zeros(dim)
            self.u_box = np.zeros(dim)

    def is_valid(self, box_constraints=False):
        """reference dense/model.hpp:104-148: sizes, symmetry of H to machine precision (Eigen's isApprox:
        ||H - H^T||_F <= eps ||H||_F), C not identically zero when there are inequality rows; raises ValueError (the
        binding's translation of std::invalid_argument) with the reference's messages, returns True otherwise"""
        def size_is(got, want, what):
            if got != want:
                raise ValueError("wrong argument size: expected %d, got %d\n%s" % (want, got, what))
        size_is(np.size(self.g), self.dim, "g has not the expected size.")
        size_is(np.size(self.b), self.n_eq, "b has not the expected size.")
        size_is(np.size(self.l), self.n_in, "l has not the expected size.")
        size_is(np.size(self.u), self.n_in, "u has not the expected size.")
        if box_constraints:
            size_is(np.size(getattr(self, "u_box", ())), self.dim, "u_box has not the expected size")
            size_is(np.size(getattr(self, "l_box", ())), self.dim, "l_box has not the expected size")
        H, A, C = (np.atleast_2d(np.asarray(m, dtype=np.float64)) for m in (self.H, self.A, self.C))
        if H.size:
            size_is(H.shape[0], self.dim, "H has not the expected number of rows.")
            size_is(H.shape[1], self.dim, "H has not the expected number of cols.")
            if np.sum((H - H.T) ** 2) > np.finfo(np.float64).eps ** 2 * np.sum(H ** 2):
                raise ValueError("H is not symmetric.")
        if A.size:
            size_is(A.shape[0], self.n_eq, "A has not the expected number of rows.")
            size_is(A.shape[1], self.dim, "A has not the expected number of cols.")
        if C.size:
            size_is(C.shape[0], self.n_in, "C has not the expected number of rows.")
            size_is(C.shape[1], self.dim, "C has not the expected number of cols.")
            if np.all(np.abs(C) <= 1e-12):
                raise ValueError("C is zero, while n_in != 0.")
        return True


class _Pool:
    """One device batch handle: `capacity` slots of one problem signature."""

    def __init__(self, capacity, n, n_eq, n_in, box, hessian_type, dense_backend, device):
        self.batch = _native.Batch(capacity, n, n_eq, n_in, box_constraints=box, hessian_type=int(hessian_type),
                                   dense_backend=int(dense_backend), device=device)
        # results are host members when a solve returns (reference parallel/qp_solve.hpp:33-37): the solve kernel writes
        # them into pinned host mirrors; a stream of its own lets this pool overlap with the others.  A pool of ONE slot
        # (a standalone QP, a one-shot dense.solve()) gets neither before its first asynchronous solve: a Python list of
        # N standalone QPs would otherwise mean N streams and 6 N pinned allocations before anything is solved.
        self._overlap = False
        if capacity > 1:
            self._enable_overlap()
        self.device = device
        self.capacity = capacity
        self.used = 0
        self._epoch = 0
        self._cache_epoch = -1
        self._cache = None

    def _enable_overlap(self):
        if not self._overlap:
            self.batch.enable_host_results(True)
            self.batch.own_stream()
            self._overlap = True

    def touch(self):
        self._epoch += 1

    def _mirrors_fresh(self):
        """the pinned mirrors hold the results of every USED slot (a partially filled pool never solves its free slots)"""
        if not self._overlap:
            return False
        return self.used > 0 and self.batch.host_results_fresh(0, self.used)

    def fetch(self):
        if self._cache_epoch != self._epoch:
            if self._mirrors_fresh():
                # one host copy out of the pinned mirrors (a snapshot: a Results object handed out earlier must not change
                # under a later solve); the Info records as a ctypes array like results() returns
                hx, hy, hz, hse, hsi, hinfo = self.batch.host_results()
                info = (_native.pqp_info * self.capacity)()
                if self.capacity:
                    _native.C.memmove(info, hinfo.ctypes.data, self.capacity * _native.C.sizeof(_native.pqp_info))
                self._cache = (hx.copy(), hy.copy(), hz.copy(), hse.copy(), hsi.copy(), info)
            else:
                self._cache = self.batch.results(-1)
            self._cache_epoch = self._epoch
        return self._cache

    def solve(self, first=None, count=None):
        if first is None:
            first, count = 0, self.used
        if count:
            self.batch.solve(first, count)
        self.touch()

    def solve_async(self, slots=None):
        """every used slot (or the listed ones, one launch) enqueued; `wait()` completes it"""
        self._enable_overlap()
        if slots is None:
            if self.used:
                self.batch.solve_async(0, self.used)
        elif len(slots):
            self.batch.solve_subset(slots, asynchronous=True)
        self.touch()

    def wait(self):
        self.batch.wait()


_INF_BOUND = 1.0e20


def _host(a):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.asarray(a, dtype=np.float64)
    return None if a.size == 0 else a


class QP:
    """reference dense/wrapper.hpp:114-963 (python: expose-qpobject.hpp:60-230)."""

    def __init__(self, n=0, n_eq=0, n_in=0, box_constraints=False, hessian_type=HessianType.Dense,
                 dense_backend=DenseBackend.Automatic, *, device=0, _pool=None, _slot=0):
        # the reference has 8 constructor overloads; python exposes this single keyword form
        if isinstance(box_constraints, (HessianType, DenseBackend)):
            raise TypeError("box_constraints must be a bool (use keywords for hessian_type / dense_backend)")
        n, n_eq, n_in = int(n), int(n_eq), int(n_in)
        if n <= 0:
            # reference dense/model.hpp:65-68 (PROXSUITE_THROW_PRETTY -> std::invalid_argument)
            raise ValueError("wrong argument size: the dimension wrt the primal variable x should be strictly positive.")
        self._box = bool(box_constraints)
        self._hessian = HessianType(hessian_type)
        if _pool is None:
            _pool = _Pool(1, n, n_eq, n_in, self._box, self._hessian, DenseBackend(dense_backend), device)
            _pool.used = 1
            _slot = 0
        self._pool, self._slot = _pool, _slot
        self._backend = DenseBackend(_pool.batch.dense_backend)
        self.model = Model(n, n_eq, n_in, self._box)
        self.settings = Settings(_pool.batch.settings(_slot))

    # -- introspection (wrapper.hpp:334-336)
    def is_box_constrained(self):
        return self._box

    def which_dense_backend(self):
        return self._backend

    def which_hessian_type(self):
        return self._hessian

    # -- helpers
    def _split_tail(self, args, kw, flag_name, flag_default):
        """The reference overloads: (..., u, flag, rho, mu_eq, mu_in, min_eig) without boxes and
        (..., u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig) with them."""
        names = ["l_box", "u_box", flag_name, "rho", "mu_eq", "mu_in", "manual_minimal_H_eigenvalue"]
        if args and isinstance(args[0], (bool, np.bool_)):
            names = names[2:]  # the overload without boxes: the first extra positional is the flag
        if len(args) > len(names):
            raise TypeError("too many positional arguments")
        vals = dict(zip(names, args))
        for k, v in kw.items():
            if k not in ("l_box", "u_box", flag_name, "rho", "mu_eq", "mu_in", "manual_minimal_H_eigenvalue"):
                raise TypeError("unexpected keyword argument %r" % k)
            if k in vals:
                raise TypeError("got multiple values for argument %r" % k)
            vals[k] = v
        vals.setdefault(flag_name, flag_default)
        if self._box and "l_box" not in vals and "u_box" not in vals:
            # the overload without boxes on a box-constrained QP (reference wrapper.hpp:367-372 / :736-741)
            raise ValueError("wrong model setup: the QP object is designed with box constraints, but is used "
                             "without lower or upper box inequalities.")
        l_box, u_box = _host(vals.get("l_box")), _host(vals.get("u_box"))
        if not self._box and (l_box is not None or u_box is not None):
            # reference dense/wrapper.hpp:542-546 / :846-850
            raise ValueError("wrong model setup: the QP object was initialized without box constraints, "
                             "but is used with box constraints inputs.")
        return (l_box, u_box, bool(vals[flag_name]), vals.get("rho"), vals.get("mu_eq"), vals.get("mu_in"),
                vals.get("manual_minimal_H_eigenvalue"))

    def _remember(self, **arrays):
        m = self.model
        for k, a in arrays.items():
            if a is not None:
                tgt = getattr(m, k)
                if a.shape == tgt.shape:
                    if k in ("l", "l_box"):
                        a = np.maximum(a, -_INF_BOUND)
                    elif k in ("u", "u_box"):
                        a = np.minimum(a, _INF_BOUND)
                    setattr(m, k, np.array(a, dtype=np.float64))

    def init(self, H=None, g=None, A=None, b=None, C=None, l=None, u=None, *args, **kw):
        """QP::init (reference dense/wrapper.hpp:354-498 and, with boxes, :520-703)."""
        l_box, u_box, flag, rho, mu_eq, mu_in, min_eig = self._split_tail(args, kw, "compute_preconditioner", True)
        H, g, A, b, C, l, u = map(_host, (H, g, A, b, C, l, u))
        self._pool.batch.init(self._slot, H, g, A, b, C, l, u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig)
        self._remember(H=H, g=g, A=A, b=b, C=C, l=l, u=u, l_box=l_box, u_box=u_box)
        self._pool.touch()

    def update(self, H=None, g=None, A=None, b=None, C=None, l=None, u=None, *args, **kw):
        """QP::update (reference dense/wrapper.hpp:723-807 / :831-918)."""
        l_box, u_box, flag, rho, mu_eq, mu_in, min_eig = self._split_tail(args, kw, "update_preconditioner", False)
        H, g, A, b, C, l, u = map(_host, (H, g, A, b, C, l, u))
        self._pool.batch.update(self._slot, H, g, A, b, C, l, u, l_box, u_box, flag, rho, mu_eq, mu_in, min_eig)
        self._remember(H=H, g=g, A=A, b=b, C=C, l=l, u=u, l_box=l_box, u_box=u_box)
        self._pool.touch()

    def solve(self, x=None, y=None, z=None):
        """QP::solve() / QP::solve(x, y, z) (reference dense/wrapper.hpp:922-957)."""
        x, y, z = _host(x), _host(y), _host(z)
        if x is not None or y is not None or z is not None:
            self._pool.batch.warm_start(self._slot, x, y, z)
        self._pool.solve(self._slot, 1)

    def cleanup(self):
        """QP::cleanup (reference dense/wrapper.hpp:958-962)."""
        self._pool.batch.cleanup(self._slot)
        self._pool.touch()

    @property
    def results(self) -> Results:
        x, y, z, se, si, info = self._pool.fetch()
        s = self._slot
        return Results(x[s], y[s], z[s], se[s], si[s], info[s])


class BatchQP:
    """reference dense/wrapper.hpp:1253-1311 (python: expose-qpvector.hpp:22-39).

    `batch_size` is the capacity of each device pool (the reference reserves a vector of that
    size).  QPs of different sizes may share a BatchQP: each signature gets its own pool."""

    def __init__(self, batch_size=0, *, device=None, devices=None):
        """`device`: a HIP device ordinal; `devices`: a list of them.  Default: every visible device -- with more than one
        the batch is spread over all of them in ONE process: every signature gets one pool per device holding a
        contiguous range of `batch_size` QPs, and solve_in_parallel launches every pool before it waits for any (the
        reference's solve_in_parallel uses every core of the host, parallel/qp_solve.hpp:41-59)."""
        self._capacity = max(int(batch_size), 1)
        if devices is None:
            if device is None:
                # PQP_BATCH_DEVICES=0 | 0,2,3 | all overrides the default (all visible devices) for a process that
                # must not allocate on GPUs it does not use
                import os
                env = os.environ.get("PQP_BATCH_DEVICES", "all").strip().lower()
                nd = _native.load().L.pqp_device_count()
                if env in ("", "all"):
                    devices = list(range(nd)) if nd > 1 else [0]
                else:
                    devices = [int(t) for t in env.split(",") if t.strip() != ""]
                    bad = [d_ for d_ in devices if d_ < 0 or d_ >= max(nd, 1)]
                    if bad or not devices:
                        raise ValueError("PQP_BATCH_DEVICES=%r: device ordinals must be in [0, %d)" % (env, max(nd, 1)))
            else:
                devices = [int(device)]
        self._devices = [int(d) for d in devices]
        self._device = self._devices[0]
        self._pools = {}   # signature -> [pools]
        self._qps = []

    def init_qp_in_place(self, dim, n_eq, n_in, box_constraints=False, hessian_type=HessianType.Dense,
                         dense_backend=DenseBackend.Automatic) -> QP:
        key = (int(dim), int(n_eq), int(n_in), bool(box_constraints), int(hessian_type), int(dense_backend))
        if key[0] <= 0:
            raise ValueError("wrong argument size: the dimension wrt the primal variable x should be strictly positive.")
        chain = self._pools.setdefault(key, [])
        if not chain or all(p.used == p.capacity for p in chain):
            # one pool per device, each with a contiguous share of `batch_size` slots (the first B % G one more)
            G = len(self._devices)
            for g, dev in enumerate(self._devices):
                cap = self._capacity // G + (1 if g < self._capacity % G else 0)
                if cap > 0:
                    chain.append(_Pool(cap, key[0], key[1], key[2], key[3], key[4], key[5], dev))
        pool = next(p for p in chain if p.used < p.capacity)
        qp = QP(key[0], key[1], key[2], key[3], HessianType(key[4]), DenseBackend(key[5]), _pool=pool,
                _slot=pool.used)
        pool.used += 1
        self._qps.append(qp)
        return qp

    def insert(self, qp: QP):
        """Copies `qp` (model, settings) into a new slot.  The reference's insert forgets to bump
        m_size (dense/wrapper.hpp:1288), which hides the inserted QP from size() and
        solve_in_parallel; here it counts."""
        m = qp.model
        new = self.init_qp_in_place(m.dim, m.n_eq, m.n_in, qp._box, qp._hessian, qp._backend)
        for name in _SETTING_NAMES:
            setattr(new.settings._s, name, getattr(qp.settings._s, name))
        extra = dict(l_box=m.l_box, u_box=m.u_box) if qp._box else {}
        new.init(m.H, m.g, m.A if m.n_eq else None, m.b if m.n_eq else None, m.C if m.n_in else None,
                 m.l if m.n_in else None, m.u if m.n_in else None,
                 compute_preconditioner=qp.settings.compute_preconditioner, **extra)
        return new

    def get(self, i) -> QP:
        return self._qps[i]

    __getitem__ = get

    def size(self):
        return len(self._qps)

    __len__ = size

    def __iter__(self):
        return iter(self._qps)

    def _all_pools(self):
        return [p for chain in self._pools.values() for p in chain]


class VectorQP(list):
    """`std::vector<dense::QP<T>>` of the reference (expose-parallel.hpp:27-31).  Prefer BatchQP:
    QPs collected here are standalone pools of one and are launched one kernel each."""

    def append(self, qp):
        if not isinstance(qp, QP):
            raise TypeError("VectorQP holds dense.QP objects")
        super().append(qp)


def solve_in_parallel(qps, num_threads=None):
    """dense::solve_in_parallel (reference parallel/qp_solve.hpp:17-59; python
    expose-parallel.hpp:33-46).  `num_threads` is accepted for compatibility; the degree of
    parallelism is one workgroup per QP over the whole device."""
    # every pool is launched before any is waited for (own streams; the pools of a multi-device BatchQP sit on different
    # devices); the results are then in the pools' pinned host mirrors
    if isinstance(qps, BatchQP):
        pools = qps._all_pools()
        for pool in pools:
            pool.solve_async()
        for pool in pools:
            pool.wait()
        return
    pools = {}
    for qp in qps:
        pools.setdefault(id(qp._pool), (qp._pool, []))[1].append(qp._slot)
    launched, first_error = [], None
    try:
        for pool, slots in pools.values():
            launched.append(pool)
            pool.solve_async(sorted(set(slots)))  # the listed slots of a pool in ONE launch (a QP listed twice is solved once)
    except Exception as e:  # a pool refused its launch: the ones already in flight are still drained below
        first_error = e
    for pool in launched:
        # every pool is waited for, whatever the others did: nothing stays in flight behind an exception, and the FIRST
        # error is the one the caller sees (a wait that raises must not hide it, nor leave later pools unwaited)
        try:
            pool.wait()
        except Exception as e:
            if first_error is None:
                first_error = e
    if first_error is not None:
        raise first_error


def estimate_minimal_eigen_value_of_symmetric_matrix(H, estimate_method_option=EigenValueEstimateMethodOption.ExactMethod,
                                                     power_iteration_accuracy=1.0e-3, nb_power_iteration=1000):
    """Host-side helper for non-convex QPs, as in the reference (dense/helpers.hpp:24-165, python
    expose-helpers.hpp:22-46): the value goes to `init(..., manual_minimal_H_eigenvalue=...)`.
    ExactMethod = smallest eigenvalue (Householder + Sturm bisection in libpqp_randqp.so);
    PowerIteration = the reference's two power iterations, restated."""
    H = np.ascontiguousarray(np.asarray(H, dtype=np.float64))
    if H.ndim != 2 or H.shape[0] != H.shape[1]:
        raise ValueError("wrong argument size: H has a number of rows different of the number of columns.")
    if not np.allclose(H, H.T, rtol=np.finfo(float).eps, atol=0.0):
        raise ValueError("H is not symmetric.")
    n = H.shape[0]
    if EigenValueEstimateMethodOption(estimate_method_option) == EigenValueEstimateMethodOption.ExactMethod:
        from ..utils import random_qp as _rq
        return float(_rq.min_eigenvalue_symmetric(H))

    def power(op):
        rhs = np.full(n, 1.0 / np.sqrt(n))
        dw = op(rhs)
        eig = 0.0
        for _ in range(int(nb_power_iteration)):
            rhs = dw / np.linalg.norm(dw)
            dw = op(rhs)
            eig = float(rhs @ dw)
            if np.max(np.abs(dw - eig * rhs)) <= power_iteration_accuracy:
                break
        return eig

    dominant = power(lambda v: H @ v)
    min_eig = dominant - power(lambda v: dominant * v - H @ v)
    return float(min(min_eig, dominant))


class VectorLossDerivatives(list):
    """`std::vector<Vec<T>>` of loss derivatives (reference expose-parallel.hpp:29-31)."""

    def append(self, v):
        super().append(np.asarray(v, dtype=np.float64))


def _store_backward(qp, rows, i):
    bd = qp.model.backward_data
    for name in BackwardData.__slots__:
        setattr(bd, name, np.array(rows[name][i]))


def compute_backward(qp, loss_derivative, eps=1e-4, rho_backward=1e-6, mu_backward=1e-6):
    """dense::compute_backward (reference dense/compute_ECJ.hpp:29-132; python expose-backward.hpp):
    fills qp.model.backward_data for one solved QP."""
    ld = np.ascontiguousarray(np.asarray(loss_derivative, dtype=np.float64)).reshape(1, -1)
    qp._pool.batch.backward(ld, eps, rho_backward, mu_backward, first=qp._slot, count=1)
    out = qp._pool.batch.backward_results(qp._slot)
    _store_backward(qp, {k: v[None] for k, v in out.items()}, 0)
    qp._pool.touch()


def solve_backward_in_parallel(num_threads=None, qps=None, loss_derivatives=None, eps=1e-4, rho_backward=1e-6,
                               mu_backward=1e-6):
    """dense::solve_backward_in_parallel (reference parallel/qp_solve.hpp:83-137; python
    expose-parallel.hpp:48-82): compute_backward for every QP, one launch per device pool."""
    if qps is None or loss_derivatives is None:
        raise TypeError("qps and loss_derivatives are required")
    qlist = list(qps) if not isinstance(qps, BatchQP) else [qps.get(i) for i in range(qps.size())]
    if len(loss_derivatives) != len(qlist):
        raise ValueError("wrong argument size: one loss derivative per QP is expected")
    pools = {}
    for qp, ld in zip(qlist, loss_derivatives):
        pools.setdefault(id(qp._pool), (qp._pool, []))[1].append((qp._slot, qp, np.asarray(ld, dtype=np.float64)))
    for pool, items in pools.values():
        items.sort(key=lambda t: t[0])
        slots = [t[0] for t in items]
        if slots == list(range(slots[0], slots[0] + len(slots))):
            ld = np.ascontiguousarray(np.stack([t[2] for t in items]))
            pool.batch.backward(ld, eps, rho_backward, mu_backward, first=slots[0], count=len(slots))
        else:
            for s, _, v in items:
                pool.batch.backward(v.reshape(1, -1), eps, rho_backward, mu_backward, first=s, count=1)
        rows = pool.batch.backward_results(-1)
        for s, qp, _ in items:
            _store_backward(qp, rows, s)
        pool.touch()


_SOLVE_TAIL = ("x", "y", "z", "eps_abs", "eps_rel", "rho", "mu_eq", "mu_in", "verbose", "compute_preconditioner",
               "compute_timings", "max_iter", "initial_guess", "check_duality_gap", "eps_duality_gap_abs",
               "eps_duality_gap_rel", "primal_infeasibility_solving", "default_H_eigenvalue_estimate")
_SOLVE_HEAD = ("H", "g", "A", "b", "C", "l", "u")


def solve(*args, **kwargs):
    """One-shot `dense::solve`: both overloads of the reference's binding
    (bindings/python/src/expose-solve.hpp; dense/wrapper.hpp:1000-1092 and, with boxes, :1133-1236):

        solve(H, g, A, b, C, l, u, x, y, z, eps_abs, ...)
        solve(H, g, A, b, C, l, u, l_box, u_box, x, y, z, eps_abs, ...)

    Python cannot overload on types.  A positional call is read as the box overload when its 8th and
    9th arguments can only be (l_box, u_box): both vectors of the primal dimension -- or None followed by
    such a vector -- while n_eq != dim, so that the 9th cannot be `y`.  When n_eq == dim the 11th argument decides
    (an array: `y` of the box overload; a scalar: `eps_abs` of the plain one); a shorter call is read as the plain
    overload (x, y).  Keywords (`l_box=`, `u_box=`, `x=`, `y=`) always work."""
    def is_vec(v, length):
        try:
            return v is not None and not np.isscalar(v) and np.asarray(v).ndim == 1 and np.asarray(v).shape[0] == length
        except Exception:
            return False

    box_names = _SOLVE_HEAD + ("l_box", "u_box") + _SOLVE_TAIL
    plain_names = _SOLVE_HEAD + _SOLVE_TAIL + ("l_box", "u_box")
    names = plain_names
    if len(args) >= 9 and "l_box" not in kwargs and "u_box" not in kwargs:
        H0 = _host(args[0])
        n0 = H0.shape[0] if H0 is not None else (len(args[1]) if args[1] is not None else 0)
        A0 = _host(args[2])
        ne0 = A0.shape[0] if A0 is not None else 0
        v8, v9 = is_vec(args[7], n0), is_vec(args[8], n0)
        if v9 and (v8 or args[7] is None):
            if ne0 == n0:
                # (x, y) and (l_box, u_box) have the same shapes: the 11th argument decides -- `y` of the box overload
                # is an array there, `eps_abs` of the plain one a scalar.  Without it the call is read as the plain
                # overload (x, y): the reference-style warm start; boxes of such a QP go by keyword.
                if len(args) >= 11 and not np.isscalar(args[10]) and args[10] is not None:
                    names = box_names
            else:
                names = box_names
    if len(args) > len(names):
        raise TypeError("solve() takes at most %d positional arguments" % len(names))
    bound = dict(zip(names, args))
    for k, v in kwargs.items():
        if k in bound:
            raise TypeError("solve() got multiple values for argument %r" % k)
        bound[k] = v
    return _solve_impl(**bound)


def solve_no_gil(*args, **kwargs):
    """`dense.solve_no_gil` of the reference binding (bindings/python/src/expose-solve.hpp:144, :206): the one-shot solve
    "while releasing the Global Interpreter Lock" -- what benchmark/timings-parallel.py:102-137 hands to a
    ThreadPoolExecutor.  Here every native call already goes through ctypes, which drops the GIL for the duration of
    the call (set-up, launch, wait), so this IS `solve`: same overloads, same arguments, same Results."""
    return solve(*args, **kwargs)


def _solve_impl(H=None, g=None, A=None, b=None, C=None, l=None, u=None, x=None, y=None, z=None, eps_abs=None,
                eps_rel=None, rho=None, mu_eq=None, mu_in=None, verbose=None, compute_preconditioner=True,
                compute_timings=False, max_iter=None, initial_guess=InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS,
                check_duality_gap=False, eps_duality_gap_abs=None, eps_duality_gap_rel=None,
                primal_infeasibility_solving=False, default_H_eigenvalue_estimate=None, l_box=None, u_box=None):
    H_, A_, C_ = _host(H), _host(A), _host(C)
    n = H_.shape[0] if H_ is not None else (len(g) if g is not None else 0)
    n_eq = A_.shape[0] if A_ is not None else 0
    n_in = C_.shape[0] if C_ is not None else 0
    box = l_box is not None or u_box is not None
    qp = QP(n, n_eq, n_in, box, HessianType.Dense, DenseBackend.PrimalDualLDLT)
    st = qp.settings
    st.initial_guess = initial_guess
    st.check_duality_gap = check_duality_gap
    for name, v in (("eps_abs", eps_abs), ("eps_rel", eps_rel), ("verbose", verbose), ("max_iter", max_iter),
                    ("eps_duality_gap_abs", eps_duality_gap_abs), ("eps_duality_gap_rel", eps_duality_gap_rel)):
        if v is not None:
            setattr(st, name, v)
    st.compute_timings = compute_timings
    st.primal_infeasibility_solving = primal_infeasibility_solving
    kw = dict(compute_preconditioner=compute_preconditioner, rho=rho, mu_eq=mu_eq, mu_in=mu_in,
              manual_minimal_H_eigenvalue=default_H_eigenvalue_estimate)
    if box:
        kw.update(l_box=l_box, u_box=u_box)
    qp.init(H_, g, A_, b, C_, l, u, **kw)
    qp.solve(x, y, z)
    return qp.results
