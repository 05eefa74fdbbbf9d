"""Diagnostic: run one Maros-Meszaros fixture with small max_iter / max_iter_in and dump the state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import split_maros
from proxsuite_amd import _native as N

def main(lib, name="DUALC5"):
    d = np.load(os.path.join(ROOT, "tests", "golden", "maros_meszaros_small.npz"))
    P, q, A, l, u = (d["%s/%s" % (name, k)] for k in "PqAlu")
    H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
    n, ne, ni = H.shape[0], Aeq.shape[0], C.shape[0]
    for (mi, mii) in [(1, 1), (1, 2), (1, 3), (1, 5), (2, 5), (3, 5), (8, 1500)]:
        bt = N.Batch(1, n, ne, ni, lib=lib)
        bt.init(0, H, g, Aeq, b, C, lin, uin)
        s = bt.settings(0)
        s.eps_abs, s.eps_rel, s.eps_primal_inf, s.eps_dual_inf, s.max_iter, s.max_iter_in = 2e-8, 0, 1e-12, 1e-12, mi, mii
        bt.solve()
        x, y, z, se, si, info = bt.results(0)
        st = bt.stats()[0]
        print("max_iter=%d/%d status=%d iter=%d ext=%d pri=%.6e dua=%.6e itres=%.2e |x|=%.12e |z|=%.12e sum(z)=%.12e nact=%d schur=%d"
              % (mi, mii, info.status, info.iter, info.iter_ext, info.pri_res, info.dua_res, info.iterative_residual,
                 np.abs(x).max(), np.abs(z).max(), z.sum(), st[15], st[11]), flush=True)
        bt.close()

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "emu":
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build as emub
        main(N.NativeLib(emub.build()))
    else:
        main(N.load())
