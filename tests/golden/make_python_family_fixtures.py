"""Writes tests/golden/python_infeasible_family.npz: the 20 instances of the reference's Python test
test/src/dense_qp_wrapper.py:4775-4821 (generate_mixed_qp(20, i), :19-48, then b += 10, u -= 100).
The generator below draws from numpy's legacy global RNG in the same order as the reference's
function; scipy.sparse.random's sampling can differ between scipy versions, which is why the instances
are committed instead of being regenerated at test time.  Needs numpy + scipy only."""
import os

import numpy as np
import scipy.sparse as spa


def generate_mixed_qp(n, seed=1, reg=0.01):
    np.random.seed(seed)
    m = int(n / 4) + int(n / 4)
    n_eq = int(n / 4)
    n_in = int(n / 4)
    P = spa.random(n, n, density=0.075, data_rvs=np.random.randn, format="csc").toarray()
    P = (P + P.T) / 2.0
    s = max(np.absolute(np.linalg.eigvals(P)))
    P += (abs(s) + reg) * spa.eye(n)
    P = spa.coo_matrix(P)
    q = np.random.randn(n)
    A = spa.random(m, n, density=0.15, data_rvs=np.random.randn, format="csc").toarray(order="C")
    v = np.random.randn(n)
    _delta = np.random.rand(m)  # drawn and unused, as in the reference
    u = A @ v
    l = -1.0e20 * np.ones(m)
    return P.toarray(), q, A[:n_eq, :], u[:n_eq], A[n_in:, :], u[n_in:], l[n_in:]


if __name__ == "__main__":
    out = {}
    for i in range(20):
        H, g, A, b, C, u, l = generate_mixed_qp(20, i)
        b = b + 10.0
        u = u - 100.0
        for k, v in zip("HgAbCul", (H, g, A, b, C, u, l)):
            out["%s_%d" % (k, i)] = np.ascontiguousarray(v, dtype=np.float64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "python_infeasible_family.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; scipy", __import__("scipy").__version__, "numpy", np.__version__)
