"""Generates tests/golden/maros_meszaros_small.npz from the reference's own fixtures
(/root/reference/test/data/maros_meszaros_data/*.mat, loaded as in
test/include/maros_meszaros.hpp:121-140: keys P,q,A,l,u; equality rows are l==u).

Only the small problems (n <= 150, rows <= 320) are kept so the fixture stays a few
hundred KB; tests/test_oracle_maros_meszaros.py additionally runs the full n<=1000 set
when /root/reference is present.  Run:  python tests/golden/make_maros_meszaros_fixtures.py
"""
import glob
import os

import numpy as np
import scipy.io as sio

SRC = "/root/reference/test/data/maros_meszaros_data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maros_meszaros_small.npz")


def load(path):
    d = sio.loadmat(path)
    if d["P"].shape[0] > 1000 or d["A"].shape[0] > 1000:
        return None
    dense = lambda m: m.toarray() if hasattr(m, "toarray") else np.asarray(m)
    P, A = dense(d["P"]).astype(np.float64), dense(d["A"]).astype(np.float64)
    q, l, u = (np.asarray(d[k], dtype=np.float64).ravel() for k in ("q", "l", "u"))
    return P, q, A, l, u


def main():
    out = {}
    names = []
    for f in sorted(glob.glob(os.path.join(SRC, "*.mat"))):
        r = load(f)
        if r is None:
            continue
        P, q, A, l, u = r
        if P.shape[0] > 150 or A.shape[0] > 320:
            continue
        name = os.path.basename(f)[:-4]
        names.append(name)
        for k, v in zip("PqAlu", (P, q, A, l, u)):
            out["%s/%s" % (name, k)] = v
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(names), "problems", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
