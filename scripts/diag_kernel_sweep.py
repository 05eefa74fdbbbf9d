"""Randomised sweep of the one-wavefront diagonal kernel against its 256-thread partner (PQP_DIAG_KERNEL switches the
dispatch per launch) and the oracle over rarely-used settings: Martinez update rule, duality-gap criterion, relative tolerance,
infeasibility check frequency, closest-feasible solving on empty boxes, small iteration caps, no preconditioner, alpha_gpdal,
both forms of the bounds, zero / diagonal Hessian.   python scripts/diag_kernel_sweep.py [seed] [count]   (LIB=<so> for the emulator)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as randqp
from oracle import oracle
from proxsuite_amd._ctypes_defs import HessianType, InitialGuess

lib = N.NativeLib(os.environ["LIB"], legacy=True) if os.environ.get("LIB") else N.load()
seed, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
bad = 0
for it in range(count):
    dim = int(rng.integers(1, 200 if not os.environ.get("LIB") else 60))
    box = bool(rng.integers(0, 2))
    hess = HessianType.Zero if rng.integers(0, 5) == 0 else HessianType.Diagonal
    B = 3
    H, g, Cm, l, u = pc.c5_models(randqp, B, dim, seed0=int(rng.integers(0, 1000)))
    if hess == HessianType.Zero:
        H = np.zeros_like(H)
    infeasible = rng.integers(0, 4) == 0
    if infeasible:
        k = int(rng.integers(0, dim)); l = l.copy(); u = u.copy(); l[:, k], u[:, k] = u[:, k] + 1.0, l[:, k]  # an empty interval
    st = dict(eps_abs=float(10.0 ** rng.integers(-9, -4)), eps_rel=float(rng.choice([0.0, 1e-6])),
              initial_guess=int(rng.choice([0, 1, 2, 3, 4])) if False else int(rng.choice([int(InitialGuess.NO_INITIAL_GUESS), int(InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS)])),
              bcl_update=int(rng.integers(0, 2)), check_duality_gap=int(rng.integers(0, 2)), frequence_infeasibility_check=int(rng.integers(1, 4)),
              primal_infeasibility_solving=int(infeasible and rng.integers(0, 2)), max_iter=int(rng.choice([3, 50, 400])), max_iter_in=int(rng.choice([2, 1500])),
              compute_preconditioner=int(rng.integers(0, 3) > 0), alpha_gpdal=float(rng.choice([0.95, 0.5])), merit_function_type=0,
              compute_timings=int(rng.integers(0, 2)), nb_iterative_refinement=int(rng.choice([1, 10])))
    ni = 0 if box else dim
    res = {}
    for kernel in ("wave", "workgroup"):
        os.environ["PQP_DIAG_KERNEL"] = kernel
        b = N.Batch(B, dim, 0, ni, box_constraints=box, hessian_type=int(hess), lib=lib)
        pc.settings_all(b, **st)
        if box:
            b.init(-1, H, g, None, None, None, None, None, l, u)
        else:
            b.init(-1, H, g, None, None, Cm, l, u)
        outs = []
        for rep in range(2):  # cold, then dirty re-solve
            b.solve()
            x, y, z, se, si, info = b.results()
            outs.append((x.copy(), z.copy(), si.copy(), [(info[i].status, info[i].iter, info[i].iter_ext, info[i].mu_updates) for i in range(B)],
                         [(info[i].objValue, info[i].pri_res, info[i].dua_res) for i in range(B)]))
        res[kernel] = outs
        b.close()
    qs = []
    for i in range(B):
        q = oracle.QP(dim, 0, ni, box_constraints=box, hessian_type=hess)
        for k_, v_ in st.items():
            setattr(q.settings, k_, type(getattr(q.settings, k_))(v_))
        if box:
            q.init(H[i], g[i], None, None, None, None, None, l[i], u[i])
        else:
            q.init(H[i], g[i], None, None, Cm[i], l[i], u[i])
        q.solve(); q.solve()
        qs.append(q)
    for rep in range(2):
        xa, za, sia, ia, ra = res["wave"][rep]; xb, zb, sib, ib, rb = res["workgroup"][rep]
        ok = ia == ib and np.allclose(xa, xb, rtol=1e-9, atol=1e-9) and np.allclose(za, zb, rtol=1e-9, atol=1e-9) and np.allclose(sia, sib, rtol=1e-8, atol=1e-8)
        if rep == 1:
            oi = [(int(q.results.info.status), q.results.info.iter, q.results.info.iter_ext, q.results.info.mu_updates) for q in qs]
            ok = ok and ia == oi and all(np.allclose(xa[i], qs[i].results.x, rtol=1e-8, atol=1e-8) for i in range(B))
        if not ok:
            bad += 1
            print("MISMATCH it", it, "rep", rep, dict(dim=dim, box=box, hess=hess.name, infeasible=bool(infeasible)), st, "\n  wave", ia, "\n  wg  ", ib,
                  "\n  orc ", oi if rep == 1 else "", "dx", float(np.max(np.abs(xa - xb))), flush=True)
print("diag kernel sweep seed %d: %d shapes, %d mismatches" % (seed, count, bad))
