"""Cases for the asynchronous solve, the host-resident results and the in-process multi-device batch (pqp_multi_*),
shared by the emulated run (tests/test_emu_multi.py, `-m "not gpu"`) and the MI355X run (tests/test_gpu_multi.py).
The reference for every check is the plain synchronous single-handle solve of the same QPs: QPs are independent and
every reduction of the kernel has a fixed order, so shards, streams and mirrors must not change one bit."""
import numpy as np

from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import InitialGuess

EPS = 1e-9


def _settings(b, guess=InitialGuess.NO_INITIAL_GUESS):
    for i in range(b.B):
        s = b.settings(i)
        s.eps_abs, s.eps_rel, s.initial_guess = EPS, 0.0, int(guess)


def _model(randqp, B, n, ne, ni, seed0=0):
    return randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2, seed0=seed0)


def _reference(lib, m, B, n, ne, ni):
    b = N.Batch(B, n, ne, ni, lib=lib)
    _settings(b)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    return b


def _same_results(a, ref, what):
    for k, (u, v) in enumerate(zip(a[:5], ref[:5])):
        assert np.array_equal(u, v), "%s: array %d differs" % (what, k)
    B = len(ref[5])
    for i in range(B):
        for f in ("status", "iter", "iter_ext", "mu_updates", "objValue", "pri_res", "dua_res", "mu_in"):
            assert getattr(a[5][i], f) == getattr(ref[5][i], f), (what, i, f)


def case_async_and_host_results(lib, randqp, n=20, ne=5, ni=8, B=6):
    m = _model(randqp, B, n, ne, ni)
    ref = _reference(lib, m, B, n, ne, ni).results()
    assert all(ref[5][i].status == 0 for i in range(B))

    b = N.Batch(B, n, ne, ni, lib=lib)
    _settings(b)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    assert not b.host_results_fresh()
    b.enable_host_results(True)
    assert not b.host_results_fresh()  # nothing solved yet
    b.solve_async()
    b.wait()
    assert b.last_solve_ms >= 0.0
    assert b.host_results_fresh() and all(b.host_results_fresh(i) for i in range(B))
    hx, hy, hz, hse, hsi, hinfo = b.host_results()
    _same_results((hx, hy, hz, hse, hsi, [b.results(i)[5] for i in range(B)]), ref, "host mirrors")
    assert [int(v) for v in hinfo["status"]] == [ref[5][i].status for i in range(B)]
    assert [int(v) for v in hinfo["iter"]] == [ref[5][i].iter for i in range(B)]
    # get_results is served from the mirrors and agrees; one QP at a time too
    _same_results(b.results(), ref, "get_results through the mirrors")
    x3 = b.results(3)
    assert np.array_equal(x3[0], ref[0][3]) and x3[5].iter == ref[5][3].iter

    # an entry that needs the device state waits for the solve in flight by itself
    b.solve_async()
    _same_results(b.results(), ref, "results() right behind solve_async()")
    # a range in flight: only that range is re-solved, the others keep their (fresh) results
    b.solve_async(2, 3)
    b.wait()
    _same_results(b.results(), ref, "range solve")

    # a warm start changes x on the device: the mirror of that QP is stale until its next solve; results() then reads
    # the device copy (the guess), and the next solve refreshes the mirror
    guess = ref[0][1] + 1.0
    b.warm_start(1, guess, ref[1][1], ref[2][1])
    assert not b.host_results_fresh(1) and b.host_results_fresh(0) and not b.host_results_fresh()
    assert np.array_equal(b.results(1)[0], guess)
    b.settings(1).initial_guess = int(InitialGuess.NO_INITIAL_GUESS)
    b.solve(1, 1)
    assert b.host_results_fresh()
    _same_results(b.results(), ref, "after the stale QP was re-solved")
    # an update makes every mirror stale
    b.update(-1, g=m.g)
    assert not b.host_results_fresh(0)
    b.solve()
    _same_results(b.results(), ref, "after update + solve")
    # switching the mirrors off returns to device-to-host copies
    b.enable_host_results(False)
    assert not b.host_results_fresh()
    _same_results(b.results(), ref, "mirrors off")
    b.solve()
    _same_results(b.results(), ref, "solve with mirrors off")


def case_two_handles_in_flight(lib, randqp, n=16, ne=4, ni=6, B=5):
    """two batch handles (pools of a BatchQP) launched back to back, then waited for"""
    ma, mb = _model(randqp, B, n, ne, ni, 0), _model(randqp, B, n, ne, ni, 100)
    ra = _reference(lib, ma, B, n, ne, ni).results()
    rb = _reference(lib, mb, B, n, ne, ni).results()
    a, b = N.Batch(B, n, ne, ni, lib=lib), N.Batch(B, n, ne, ni, lib=lib)
    for h, m in ((a, ma), (b, mb)):
        _settings(h)
        h.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
        h.enable_host_results(True)
        h.flush()
    a.solve_async()
    b.solve_async()
    b.wait()
    a.wait()
    _same_results(a.results(), ra, "first handle")
    _same_results(b.results(), rb, "second handle")


def case_multi(lib, randqp, devices, n=20, ne=5, ni=8, B=7, gather_alloc=None):
    """the batch split over len(devices) shards is bit-exact against the single-handle solve"""
    m = _model(randqp, B, n, ne, ni)
    refb = _reference(lib, m, B, n, ne, ni)
    ref = refb.results()
    G = len(devices)
    mb = N.MultiBatch(B, n, ne, ni, devices, lib=lib)
    assert mb.shard_count == G
    spans = [mb.shard(g) for g in range(G)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == B
    assert all(spans[g][0] + spans[g][1] == spans[g + 1][0] for g in range(G - 1))
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    for i in range(B):
        s = mb.settings(i)
        s.eps_abs, s.eps_rel, s.initial_guess = EPS, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
    mb.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    mb.flush()
    mb.solve()
    assert mb.last_solve_ms >= 0.0
    _same_results(mb.results(), ref, "multi solve, %d shards" % G)
    one = mb.results(B - 1)
    assert np.array_equal(one[0], ref[0][B - 1]) and one[5].iter == ref[5][B - 1].iter
    # asynchronous form + the gathered device buffer (pack kernel per shard, copies into place)
    width = n + ne + ni + 2
    out = gather_alloc(B * width) if gather_alloc else np.zeros(B * width)
    mb.solve_async()
    mb.gather_device(out, root_shard=G - 1)
    got = (out.cpu().numpy() if hasattr(out, "cpu") else out).reshape(B, width)
    assert np.array_equal(got[:, :n], ref[0]) and np.array_equal(got[:, n:n + ne], ref[1])
    assert np.array_equal(got[:, n + ne:n + ne + ni], ref[2])
    assert [int(v) for v in got[:, -2]] == [ref[5][i].status for i in range(B)]
    assert [int(v) for v in got[:, -1]] == [ref[5][i].iter for i in range(B)]
    # per-QP addressing across the shard boundaries: update one QP of every shard, warm-start another, solve a range
    g2 = m.g.copy()
    touched = sorted({f for f, c in spans if c > 0} | {f + c - 1 for f, c in spans if c > 0})
    for i in touched:
        g2[i] = m.g[i] * 1.5
        mb.update(i, g=g2[i])
        refb.update(i, g=g2[i])
    refb.solve()
    ref2 = refb.results()
    lo, hi = touched[0], touched[-1] + 1
    mb.solve(lo, hi - lo)
    _same_results(mb.results(), ref2, "multi: per-QP update + range solve")
    # a whole-batch update through the [B]-leading arrays, cleanup of one QP
    mb.update(-1, g=m.g)
    refb.update(-1, g=m.g)
    mb.solve()
    refb.solve()
    _same_results(mb.results(), refb.results(), "multi: batch update")
    mb.close()


def case_multi_verbose_trace(lib, randqp, devices, n=16, ne=4, ni=6, B=5):
    """settings.verbose across shards: the per-iteration trace of a QP is the one its single-handle solve records
    (pqp_multi_get_trace reaches into the shard that holds it); a QP that is not verbose has none"""
    import os
    m = _model(randqp, B, n, ne, ni)
    out = []
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)  # (the library prints the lines)
    try:
        for multi in (False, True):
            b = N.MultiBatch(B, n, ne, ni, devices, lib=lib) if multi else N.Batch(B, n, ne, ni, lib=lib)
            for i in range(B):
                s = b.settings(i)
                s.eps_abs, s.eps_rel, s.initial_guess = EPS, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
                s.verbose = 0 if i == 1 else 1
            b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
            os.dup2(devnull, 1)
            try:
                b.solve()
            finally:
                os.dup2(saved, 1)
            out.append([b.trace(i) for i in range(B)])
            b.close()
    finally:
        os.close(devnull)
        os.close(saved)
    for i in range(B):
        assert np.array_equal(out[0][i], out[1][i]), i
        assert (out[0][i].shape[0] == 0) == (i == 1)


def case_multi_errors(lib):
    import pytest
    with pytest.raises(ValueError):
        N.MultiBatch(4, 5, 1, 1, [], lib=lib)
    mb = N.MultiBatch(3, 5, 1, 1, [0, 0], lib=lib)
    with pytest.raises(IndexError):
        mb.settings(3)
    with pytest.raises(ValueError):
        mb.solve(2, 5)
    with pytest.raises((ValueError, N.NativeError)):
        N.MultiBatch(3, 5, 1, 1, [0, 99], lib=lib)
