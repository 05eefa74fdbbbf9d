"""Diagnostic: per-problem status / iteration counts on the small Maros-Meszaros fixtures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import split_maros
from proxsuite_amd import _native as N

def main(lib, max_iter=300):
    d = np.load(os.path.join(ROOT, "tests", "golden", "maros_meszaros_small.npz"))
    for name in [str(s) for s in d["names"]]:
        P, q, A, l, u = (d["%s/%s" % (name, k)] for k in "PqAlu")
        H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
        n, ne, ni = H.shape[0], Aeq.shape[0], C.shape[0]
        bt = N.Batch(1, n, ne, ni, lib=lib)
        bt.init(0, H, g, Aeq, b, C, lin, uin)
        s = bt.settings(0)
        s.eps_abs, s.eps_rel, s.eps_primal_inf, s.eps_dual_inf, s.max_iter = 2e-8, 0, 1e-12, 1e-12, max_iter
        t = time.time(); bt.solve(); dt = time.time() - t
        x, y, z, se, si, info = bt.results(0)
        st = bt.stats()[0]
        print("%-10s n=%3d ne=%3d ni=%3d status=%d iter=%5d ext=%4d mu_upd=%3d pri=%.2e dua=%.2e itres=%.1e newton=%d schur=%d rows=%d %.2fs"
              % (name, n, ne, ni, info.status, info.iter, info.iter_ext, info.mu_updates, info.pri_res, info.dua_res,
                 info.iterative_residual, st[10], st[11], st[12], dt), flush=True)
        bt.close()

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "emu":
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build as emub
        main(N.NativeLib(emub.build()))
    else:
        main(N.load())
