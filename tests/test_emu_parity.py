"""`-m "not gpu"`: the HIP device sources run on the CPU SIMT emulator (tests/emu) and are
checked against the oracle.  Validates kernel LOGIC before any GPU minute is spent; the
numerics on the real device are covered by tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import HessianType, InitialGuess

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def lib():
    import build as emu_build
    return N.NativeLib(emu_build.build())


def test_library_exports_every_declared_symbol(lib):
    for name in N.NativeLib.SYMBOLS:
        assert hasattr(lib.L, name), name


def test_known_answers(lib):
    pc.case_known_answers(lib)


def test_ruiz(lib, oracle, randqp):
    pc.case_ruiz(lib, oracle, randqp)


@pytest.mark.parametrize("shape", [(10, 2, 3), (30, 7, 9), (33, 8, 11), (50, 25, 50), (100, 50, 100)])  # odd n: 8-byte-load path of gemv_dual
def test_random_batch(lib, oracle, randqp, shape):
    n, ne, ni = shape
    pc.case_random_batch(lib, oracle, randqp, n, ne, ni, B=8 if n < 100 else 4)


@pytest.mark.parametrize("shape", [(30, 7, 9, 8, 2), (40, 5, 300, 3, 1)])
def test_launch_size_invariance(lib, randqp, shape, monkeypatch):
    """the launcher's choice between the two register-budget builds of a solve kernel (by launch
    size; the emulated device has one CU) must not change a result"""
    n, ne, ni, B, chunk = shape
    monkeypatch.setenv("PQP_DENSE_KERNEL", "workgroup")  # (the one-wavefront kernel, which launches of 64 QPs and more may take, sums in another order: tests/test_gpu_parity.py::test_launch_size_wave_vs_workgroup)
    pc.case_launch_size_invariance(lib, randqp, n, ne, ni, B, chunk)


@pytest.mark.parametrize("kernel", ["wave", "workgroup"])
def test_dense_wave_kernel_flows(lib, oracle, randqp, monkeypatch, kernel):
    """The dense solver as ONE wavefront per QP (csrc/pqp_dwave.hpp behind the factorisation prologue) -- and, as its A/B
    partner, the 256-thread workgroup kernel on the same cases (PQP_DENSE_KERNEL forces either): random batches over the
    shapes its layout distinguishes (odd n, no equalities, no inequalities, a Schur block beyond the 96 slots that are
    factorised in registers), every initial guess of the state machine (restored / edited factors, warm starts, updates),
    the infeasibility statuses, the verbose trace: each against the oracle, 1e-10 and equal Info."""
    monkeypatch.setenv("PQP_DENSE_KERNEL", kernel)
    for (n, ne, ni, B) in [(10, 2, 3, 3), (33, 8, 11, 3), (12, 0, 9, 2), (9, 5, 0, 2), (50, 25, 50, 2), (64, 60, 70, 2), (100, 50, 100, 3)]:
        pc.case_random_batch(lib, oracle, randqp, n, ne, ni, B)
    for guess in InitialGuess:
        pc.case_state_machine(lib, oracle, randqp, guess)
    pc.case_infeasibility_statuses(lib, oracle)
    pc.case_verbose_round_trip(lib, oracle, randqp)
    # QPLayer backward on the state the forward kernel left in HBM (slot list, W_S, D_S, active-set flags)
    pc.case_backward(lib, oracle, randqp)
    pc.case_backward(lib, oracle, randqp, with_dual_terms=False)


@pytest.mark.parametrize("shape", [(120, 100, 100), (40, 5, 300), (130, 10, 20)])
def test_matrix_core_fallback_paths(lib, oracle, randqp, shape):
    """shapes that leave the register-resident factorisations: a dual Schur block above 112 rows
    (256 threads), a 512-thread workgroup, and a primal block above 112 columns -- the blocked
    LDL^T and the row-wise triangular inverse on the matrix cores (ldlt_factor_mfma,
    tri_inverse_mfma_rows)."""
    n, ne, ni = shape
    pc.case_random_batch(lib, oracle, randqp, n, ne, ni, B=2)


@pytest.mark.parametrize("shape", [(10, 4, 7), (12, 0, 9), (9, 5, 0)])
def test_backward(lib, oracle, randqp, shape):
    pc.case_backward(lib, oracle, randqp, *shape, B=4)


@pytest.mark.parametrize("guess", list(InitialGuess))
def test_state_machine(lib, oracle, randqp, guess):
    pc.case_state_machine(lib, oracle, randqp, guess)


@pytest.mark.parametrize("hessian", [HessianType.Dense, HessianType.Diagonal])
def test_box_constraints(lib, oracle, randqp, hessian):
    pc.case_box_constraints(lib, oracle, randqp, seeds=8, hessian=hessian)


@pytest.mark.parametrize("dim", [10, 40])
def test_families(lib, oracle, randqp, dim):
    pc.case_families(lib, oracle, randqp, dim)


def test_maros_meszaros_small(lib):
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maros_meszaros_small.npz")
    d = np.load(gold)
    for name in ("HS21", "HS35", "HS76", "TAME", "ZECEVIC2", "HS118", "LOTSCHD", "GENHS28", "QPTEST"):
        pc.case_maros_meszaros(lib, *(d["%s/%s" % (name, k)] for k in "PqAlu"))


def test_errors(lib):
    pc.case_errors(lib)


def test_determinism(lib, randqp):
    pc.case_determinism(lib, randqp)


def test_infeasibility_statuses(lib, oracle):
    pc.case_infeasibility_statuses(lib, oracle)


def test_rows_beyond_the_workgroup(oracle, randqp):
    """Problems above 1024 rows run the 1024-thread kernel with its one-thread-per-row stages (row deletion from the
    Schur factor, the rank-1 update of the PrimalLDLT factor) walking the rows in chunks.  Here the same chunk loops
    (emulator variant -DPQP_CHUNK_ALL=1) under a deliberately narrow workgroup (PQP_TEST_NT_MAX=256 for shapes of 600
    constraint rows / 300 variables), against the oracle."""
    import build as emu_build
    lib2 = N.NativeLib(emu_build.build_variant("chunkall", ["PQP_CHUNK_ALL=1"]))
    os.environ["PQP_TEST_NT_MAX"] = "256"
    try:
        b = N.Batch(1, 30, 4, 600, lib=lib2)
        assert b.launch_config()[0] == 256
        b.close()
        pc.case_random_batch(lib2, oracle, randqp, 30, 4, 600, B=2)
        pc.case_primal_ldlt(lib2, oracle, randqp, dim=280, B=1)
    finally:
        del os.environ["PQP_TEST_NT_MAX"]


def test_seed14_mechanism_on_the_emulated_device(lib, randqp):
    """(two phases of the cycle here -- the emulator is slow; the MI355X run sweeps the full period)"""
    n_ok, n = pc.case_seed14_mechanism(lib, randqp, guards=(41, 52), max_iter=2500)
    assert n == 2


def test_verbose_round_trip(lib, oracle, randqp, capfd):
    pc.case_verbose_round_trip(lib, oracle, randqp)
    capfd.readouterr()  # (the host-side report of the verbose QPs)


def test_verbose_trace(lib, oracle, randqp, capfd):
    """per-iteration lines of settings.verbose: recorded by the kernel, equal to the oracle's line by line, printed"""
    assert pc.case_verbose_trace(lib, oracle, randqp, capfd) > 20


def test_closest_feasible(lib, oracle, randqp):
    """seeds whose oracle run is short enough for the emulator (the GPU test runs all 20)"""
    seen = pc.case_closest_feasible(lib, oracle, randqp, seeds=range(6), max_oracle_iter_ext=60)
    assert 0 in seen


def test_c5_forms_small(lib, oracle, randqp):
    """BASELINE.json configs[4] generator at a small dimension: C = I form and box form."""
    xa, za = pc.case_c5(lib, oracle, randqp, B=2, sample=2, box=False, dim=24)
    xb, zb = pc.case_c5(lib, oracle, randqp, B=2, sample=2, box=True, dim=24)
    assert np.max(np.abs(xa - xb)) <= 1e-7 * (1 + np.max(np.abs(xa)))


def test_diag_mixed_handle(lib, oracle, randqp):
    """range / subset launches of structured QPs out of a handle that also holds a general one (parity_cases.case_diag_mixed_handle)"""
    pc.case_diag_mixed_handle(lib, oracle, randqp)


@pytest.mark.parametrize("kernel", ["wave", "workgroup"])
def test_diag_wave_kernel_flows(lib, oracle, randqp, monkeypatch, kernel):
    """the diagonal-structure solver as one wavefront per QP with its vectors in registers (csrc/pqp_diag.hpp) -- and, as
    its A/B partner (PQP_DIAG_KERNEL=workgroup), the 256-thread form of the same solver -- through the whole solve
    state machine against the oracle: both forms of the bounds, diagonal and zero Hessian, no inequality at all, PDAL"""
    monkeypatch.setenv("PQP_DIAG_KERNEL", kernel)
    pc.case_diag_wave_flows(lib, oracle, randqp, 24, box=False)
    pc.case_diag_wave_flows(lib, oracle, randqp, 70, box=True)
    pc.case_diag_wave_flows(lib, oracle, randqp, 130, box=False, B=2)
    pc.case_diag_wave_flows(lib, oracle, randqp, 40, box=True, hessian=HessianType.Zero)
    pc.case_diag_wave_flows(lib, oracle, randqp, 30, box=False, constrained=False)
    forked = pc.case_diag_wave_flows(lib, oracle, randqp, 36, box=True, merit=1) + pc.case_diag_wave_flows(lib, oracle, randqp, 36, box=False, merit=1)
    assert forked <= 6, forked
    assert pc.case_diag_wave_infeasible(lib, oracle) != int(pc.QPSolverOutput.PROXQP_SOLVED)
    pc.case_diag_wave_backward(lib, oracle, randqp)


def test_primal_ldlt_engine(lib, oracle, randqp):
    pc.case_primal_ldlt(lib, oracle, randqp, dim=12, B=2)


def test_primal_ldlt_engine_without_box(lib, oracle, randqp):
    """forced PrimalLDLT on an ordinary shape (dense Hessian, no box): same answers as the default engine"""
    from proxsuite_amd._ctypes_defs import DenseBackend
    n, ne, ni, B = 30, 7, 9, 3
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    out = []
    for backend in (DenseBackend.PrimalLDLT, DenseBackend.PrimalDualLDLT):
        b = N.Batch(B, n, ne, ni, dense_backend=int(backend), lib=lib)
        pc.settings_all(b, eps_abs=pc.EPS, eps_rel=0)
        b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
        b.solve()
        x, y, z, se, si, info = b.results()
        for i in range(B):
            assert info[i].status == 0
            pri, dua = pc.kkt(oracle, m, i, x[i], y[i], z[i])
            assert pri <= pc.EPS and dua <= pc.EPS
        out.append((x, y, z))
        b.close()
    for a, r in zip(out[0], out[1]):
        assert pc.close(a, r)


def test_refinement_fallback(lib, oracle):
    pc.case_refinement_fallback(lib, oracle, names=("QADLITTL", "QSHARE2B"))


def test_schur_factor_identity(lib, randqp):
    pc.case_schur_factor_identity(lib, randqp, n=30, ne=7, ni=30, B=8)


def test_vectors_in_hbm_path(lib, oracle, randqp, monkeypatch):
    """shapes whose per-QP vectors exceed the 160 KiB of LDS run the solver on an HBM slice per workgroup
    (pqp_solve_hbm_kernel, 1024 threads); PQP_FORCE_HBM_VECTORS sends a small batch down that path"""
    monkeypatch.setenv("PQP_FORCE_HBM_VECTORS", "1")
    pc.case_random_batch(lib, oracle, randqp, 30, 7, 9, B=2)
    pc.case_box_constraints(lib, oracle, randqp, seeds=2)


def test_line_search_bracket_path(lib, oracle, randqp):
    """more breakpoints than threads (2 n_c > 256 in a 256-thread kernel of the general signature): the line search
    brackets the zero of phi' and evaluates exactly only around it -- iterates and Info counters must stay the oracle's.
    C5 forms at dim 140 (diagonal-structure path), then a dense boxed shape and a diagonal-Hessian shape with general C,
    cold solve and warm re-solve"""
    pc.case_c5(lib, oracle, randqp, B=2, sample=2, box=False, dim=140)
    pc.case_c5(lib, oracle, randqp, B=2, sample=2, box=True, dim=140)
    r = pc.case_random_sweep(lib, oracle, randqp, 3, 3, verbose=True,
                             shapes=[(60, 10, 80, True, 1, 1), (70, 0, 150, False, 2, 1), (60, 10, 80, True, 1, 1)])
    # (the third shape runs with the PDAL merit function: sixteen-value bracket rounds, exact values through ls_ineq_terms)
    assert r["failures"] == 0 and r["info_mismatch"] == 0 and r["solved"] >= 6, r


def test_line_search_bracket_is_bit_identical_to_the_full_evaluation(randqp):
    """The bracketing line search claims the SAME floating-point step as the all-breakpoints evaluation: the same device
    sources compiled with -DPQP_LS_BRACKET=0 must give bit-identical x, y, z and Info counters on shapes that take the
    bracket (the C5 form at dim 130 with both merit functions, a dense boxed shape)."""
    import build as emu_build
    from proxsuite_amd._ctypes_defs import HessianType
    full = N.NativeLib(emu_build.build_variant("nobracket", ["PQP_LS_BRACKET=0"]))
    brk = N.NativeLib(emu_build.build())

    def run(libx, n, ne, ni, box, hess, merit, seed):
        B = 2
        m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.4, 1e-2, seed0=seed)
        H = m.H if hess == 1 else np.stack([np.diag(np.diag(h)) for h in m.H])
        rng = np.random.default_rng(seed)
        kw = {}
        if box:
            xs = rng.standard_normal((B, n)); sh = rng.uniform(0.1, 1.0, (B, n))
            kw = dict(l_box=xs - sh, u_box=xs + sh)
        b = N.Batch(B, n, ne, ni, box_constraints=box, hessian_type=hess, lib=libx)
        for i in range(B):
            s = b.settings(i); s.eps_abs = 1e-9; s.eps_rel = 0; s.initial_guess = 0; s.merit_function_type = merit
        b.init(-1, H, m.g, m.A if ne else None, m.b if ne else None, m.C if ni else None, m.l if ni else None,
               m.u if ni else None, **kw)
        b.solve()
        x, y, z, se, si, info = b.results()
        out = (x.copy(), y.copy(), z.copy(), [(info[i].iter, info[i].iter_ext, info[i].status) for i in range(B)])
        b.close()
        return out

    for (n, ne, ni, box, hess, merit, seed) in [(130, 0, 130, False, 2, 0, 1), (130, 0, 130, False, 2, 1, 3),
                                                (60, 10, 80, True, 1, 0, 4)]:
        a = run(brk, n, ne, ni, box, hess, merit, seed)
        f = run(full, n, ne, ni, box, hess, merit, seed)
        assert a[3] == f[3], (n, ne, ni, box, hess, merit, a[3], f[3])
        for u, v in zip(a[:3], f[:3]):
            assert np.array_equal(u, v), (n, ne, ni, box, hess, merit, float(np.max(np.abs(u - v))))


def test_diag_kernel_settings_sweep(lib):
    """the one-wavefront diagonal kernel against its 256-thread partner and the oracle over rarely-used settings (Martinez
    rule, duality-gap criterion, relative tolerance, infeasibility-check frequency, closest-feasible solving on empty boxes,
    iteration caps, no preconditioner, alpha_gpdal; both forms of the bounds, zero / diagonal Hessian; cold and dirty solves)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_kernel_sweep.py"), "1", "40"], capture_output=True, text=True,
                       env=dict(os.environ, LIB=lib.path), timeout=1200)
    assert "40 shapes, 0 mismatches" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_rows_above_4096(lib, oracle, randqp):
    """4500 constraint rows on six variables through the emulated 1024-thread kernel (chunked row stages, per-QP vectors on a
    heap slice as in pqp_solve_hbm_kernel): the limit of a batch is PQP_MAX_ROWS = 8192 (csrc/pqp_host.hpp; round 4: 4096),
    and beyond it the library says so"""
    pc.case_random_batch(lib, oracle, randqp, 6, 0, 4500, B=1, compare="all", info_residuals=False)
    with pytest.raises(N.NativeError):
        N.Batch(1, 10, 0, 9000, lib=lib)
