"""`-m "not gpu"`: the Python surface (proxsuite_amd.proxqp.dense, proxsuite_amd.torch) driven
through the CPU SIMT emulator build of the device code (tests/emu) -- TEST ONLY: the emulator
library is injected in place of libproxqp_hip.so, which the product never does."""
import os
import sys

import pytest

import api_cases as ac
from proxsuite_amd import _native as N

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def dense():
    import build as emu_build
    saved = N._lib
    N._lib = N.NativeLib(emu_build.build())
    from proxsuite_amd.proxqp import dense as d
    yield d
    N._lib = saved


def test_qp_object(dense, oracle, randqp):
    ac.case_qp_object(dense, oracle, randqp)


def test_errors(dense):
    ac.case_errors(dense)


def test_box(dense, oracle, randqp):
    ac.case_box(dense, oracle, randqp)


def test_batch_and_parallel(dense, oracle, randqp):
    ac.case_batch_and_parallel(dense, oracle, randqp)


@pytest.mark.parametrize("devices", [[0, 1, 2], [2, 2]])
def test_multi_device_batch(dense, randqp, devices):
    """BatchQP over several devices from one process (logical shards: every ordinal is the same emulated device)"""
    import os
    os.environ["HIPEMU_DEVICES"] = "4"
    ac.case_multi_device_batch(dense, randqp, devices)


def test_one_shot_solve(dense, oracle, randqp):
    ac.case_one_shot_solve(dense, oracle, randqp)


def test_qpfunction_forward(dense, oracle, randqp):
    from proxsuite_amd.torch import QPFunction
    ac.case_qpfunction(QPFunction, oracle, randqp, device="cpu")


def test_nonconvex_helpers(dense):
    ac.case_nonconvex_helpers(dense)


def test_backward_api(dense, oracle, randqp):
    ac.case_backward_api(dense, oracle, randqp)


def test_qpfunction_backward(dense):
    from proxsuite_amd.torch import QPFunction
    ac.case_qpfunction_backward(QPFunction, device="cpu")


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the real library must refuse to load (no oracle, no emulator behind it)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    saved = N._lib
    N._lib = None
    try:
        with pytest.raises(N.NativeError):
            N.load()
    finally:
        N._lib = saved


def test_timings_and_verbose(dense, randqp, capfd):
    ac.case_timings_and_verbose(dense, randqp, capfd)


def test_alias_package(dense):
    ac.case_alias_package()


def test_box_calibration_is_refused_on_the_emulator():
    """pqp_box_calibrate has no machine to measure under the emulator: an error code and a message, no kernels"""
    import ctypes as C
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import build as emu_build
    from proxsuite_amd import _native as N
    lib = N.NativeLib(emu_build.build())
    out = (C.c_double * 8)()
    assert lib.L.pqp_box_calibrate(0, out, 8) != 0
    assert b"emulator" in lib.L.pqp_last_error()
