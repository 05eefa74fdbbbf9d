"""The reference's dense wrapper test cases (test/src/dense_qp_wrapper.cpp, tests/wrapper_cases.py restates them
one by one) on the oracle -- the restatement must meet the reference's own acceptance lines -- and on the device
engine behind the Python facade: CPU SIMT emulator build here (`-m "not gpu"`), the real library on the GPU box
(`-m gpu`), each compared with the oracle's run of the same case solve by solve."""
import os
import sys

import pytest

import wrapper_cases as wc
from proxsuite_amd import _native as N

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

NAMES = sorted(wc.CASES)


def oracle_side(oracle, randqp):
    S = wc.Side(oracle.QP, oracle.kkt_residuals, randqp, "oracle")
    S.one_shot = wc.oracle_one_shot(oracle.QP)
    return S


def device_side(dense, oracle, randqp, name):
    S = wc.Side(dense.QP, oracle.kkt_residuals, randqp, name)
    S.one_shot = dense.solve  # the product's one-shot function
    return S


def _scaling_oracle(q):
    s = q.scaled()
    return s["delta"], s["c"]


def _scaling_device(q):
    s = q._pool.batch.scaled(q._slot)
    return s["delta"], s["c"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_meets_reference_acceptance(oracle, randqp, name):
    S = oracle_side(oracle, randqp)
    S.scaling = _scaling_oracle
    wc.CASES[name](S)
    assert S.trace


def _on_python_problem(S, name):
    """the same flow on the problem of the reference's Python suite (test/src/dense_qp_wrapper.py: its test_case_* /
    test_sparse_problem_* are these flows on `generate_mixed_qp(10)`)"""
    if name in wc.OWN_PROBLEM:
        pytest.skip("this case builds its own problem")
    S.source = "python"
    try:
        wc.CASES[name](S)
    except wc.NotForThisSource:
        pytest.skip("this case builds its own problem (C++ suite only)")


@pytest.mark.parametrize("name", NAMES)
def test_oracle_meets_reference_acceptance_python_suite_problem(oracle, randqp, name):
    S = oracle_side(oracle, randqp)
    S.scaling = _scaling_oracle
    _on_python_problem(S, name)
    assert S.trace


@pytest.fixture(scope="module")
def emu_dense():
    import build as emu_build
    saved = N._lib
    N._lib = N.NativeLib(emu_build.build())
    from proxsuite_amd.proxqp import dense as d
    yield d
    N._lib = saved


def _device_against_oracle(dense, oracle, randqp, name, label, source="cpp"):
    ref = oracle_side(oracle, randqp)
    ref.scaling = _scaling_oracle
    dev = device_side(dense, oracle, randqp, label)
    dev.scaling = _scaling_device
    if source == "python":
        _on_python_problem(ref, name)
        _on_python_problem(dev, name)
    else:
        wc.CASES[name](ref)
        wc.CASES[name](dev)
    wc.compare_traces(dev.trace, ref.trace)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_device_matches_oracle(emu_dense, oracle, randqp, name):
    _device_against_oracle(emu_dense, oracle, randqp, name, "device (emulator)")


# The same flows on the DenseBackend::PrimalLDLT engine (reference dense/solver.hpp:88-109, 171-227: no test of the
# reference runs its update / warm-start flows on that backend; oracle and device must still agree solve by solve).
ONE_SHOT = {n for n in NAMES if n.startswith("solve_")}  # dense::solve fixes PrimalDualLDLT (wrapper.hpp:1043)


def _primal_ldlt(dense, oracle, randqp, name, label):
    if name in ONE_SHOT or name == "primal_ldlt_mu_update":
        pytest.skip("backend fixed by the case")
    ref = oracle_side(oracle, randqp)
    ref.scaling = _scaling_oracle
    ref.backend = wc.PRIMAL_LDLT
    wc.CASES[name](ref)
    dev = device_side(dense, oracle, randqp, label)
    dev.scaling = _scaling_device
    dev.backend = wc.PRIMAL_LDLT
    wc.CASES[name](dev)
    wc.compare_traces(dev.trace, ref.trace)


@pytest.mark.parametrize("name", NAMES)
def test_emulated_device_matches_oracle_primal_ldlt_backend(emu_dense, oracle, randqp, name):
    _primal_ldlt(emu_dense, oracle, randqp, name, "device (emulator)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_oracle_primal_ldlt_backend(oracle, randqp, name):
    from proxsuite_amd.proxqp import dense
    N.load()
    _primal_ldlt(dense, oracle, randqp, name, "device")


@pytest.mark.parametrize("name", NAMES)
def test_emulated_device_matches_oracle_python_suite_problem(emu_dense, oracle, randqp, name):
    _device_against_oracle(emu_dense, oracle, randqp, name, "device (emulator)", source="python")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_oracle_python_suite_problem(oracle, randqp, name):
    from proxsuite_amd.proxqp import dense
    N.load()
    _device_against_oracle(dense, oracle, randqp, name, "device", source="python")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_oracle(oracle, randqp, name):
    from proxsuite_amd.proxqp import dense
    N.load()  # the real library or a loud failure
    _device_against_oracle(dense, oracle, randqp, name, "device")
