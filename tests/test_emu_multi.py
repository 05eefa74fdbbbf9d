"""`-m "not gpu"`: asynchronous solve, host-resident results and the in-process multi-device batch (pqp_multi_*)
on the CPU SIMT emulator (tests/emu; HIPEMU_DEVICES emulates several devices, all of them the host)."""
import os
import sys

import pytest

import multi_cases as mc
from proxsuite_amd import _native as N

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def lib():
    import build as emu_build
    os.environ["HIPEMU_DEVICES"] = "4"
    return N.NativeLib(emu_build.build())


def test_async_solve_and_host_results(lib, randqp):
    mc.case_async_and_host_results(lib, randqp)


def test_two_handles_in_flight(lib, randqp):
    mc.case_two_handles_in_flight(lib, randqp)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 1, 2, 3], [1, 1, 3]])
def test_multi_device_batch_is_bit_exact(lib, randqp, devices):
    mc.case_multi(lib, randqp, devices)


def test_multi_more_shards_than_qps(lib, randqp):
    mc.case_multi(lib, randqp, [0, 1, 2, 3], B=3)


def test_multi_verbose_trace(lib, randqp):
    mc.case_multi_verbose_trace(lib, randqp, [0, 1, 2])


def test_multi_errors(lib):
    mc.case_multi_errors(lib)
