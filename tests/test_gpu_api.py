"""`-m gpu`: the Python surface (proxsuite_amd.proxqp.dense, proxsuite_amd.torch) on the real
MI355X library, same cases as tests/test_emu_api.py, plus ROCm-tensor inputs for the QPLayer."""
import pytest

import api_cases as ac

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dense():
    from proxsuite_amd import _native as N
    N.load()  # fails loudly if libproxqp_hip.so or the GPU is missing
    from proxsuite_amd.proxqp import dense as d
    return d


def test_qp_object(dense, oracle, randqp):
    ac.case_qp_object(dense, oracle, randqp)


def test_errors(dense):
    ac.case_errors(dense)


def test_box(dense, oracle, randqp):
    ac.case_box(dense, oracle, randqp)


def test_batch_and_parallel(dense, oracle, randqp):
    ac.case_batch_and_parallel(dense, oracle, randqp)


@pytest.mark.parametrize("devices", [[0, 0, 0], [0, 0]])
def test_multi_device_batch(dense, randqp, devices):
    """BatchQP over several devices from one process (logical shards: every ordinal is the same GPU here)"""
    ac.case_multi_device_batch(dense, randqp, devices)


def test_one_shot_solve(dense, oracle, randqp):
    ac.case_one_shot_solve(dense, oracle, randqp)


@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_qpfunction_forward(dense, oracle, randqp, device):
    from proxsuite_amd.torch import QPFunction
    ac.case_qpfunction(QPFunction, oracle, randqp, device=device)


def test_nonconvex_helpers(dense):
    ac.case_nonconvex_helpers(dense)


def test_backward_api(dense, oracle, randqp):
    ac.case_backward_api(dense, oracle, randqp)


@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_qpfunction_backward(dense, device):
    from proxsuite_amd.torch import QPFunction
    ac.case_qpfunction_backward(QPFunction, device=device)


def test_omp_get_max_threads(dense):
    from proxsuite_amd import proxqp
    assert proxqp.omp_get_max_threads() >= 256


def test_timings_and_verbose(dense, randqp, capfd):
    ac.case_timings_and_verbose(dense, randqp, capfd)


def test_alias_package(dense):
    ac.case_alias_package()


def test_box_calibration(dense):
    """pqp_box_calibrate (include/proxqp_hip.h): the three fixed kernels report sane figures for an MI355X, twice the same
    within a few percent, and a bad argument is an error, not a crash"""
    from proxsuite_amd import _native as N
    a, b = N.box_calibration(0), N.box_calibration(0)
    for c in (a, b):
        assert 1000.0 < c["hbm_read_gbs"] < 8000.0 and 0.2 < c["chain_ms"] < 10.0 and 500.0 < c["sclk_mhz_est"] < 3000.0, c
        assert c["n_cu"] >= 64
    assert abs(a["chain_ms"] / b["chain_ms"] - 1.0) < 0.1, (a, b)
    lib = N.load()
    assert lib.L.pqp_box_calibrate(0, None, 0) != 0
