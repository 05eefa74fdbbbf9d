"""Generates the Maros-Meszaros fixtures from the reference's own data files
(/root/reference/test/data/maros_meszaros_data/*.mat, loaded as in
test/include/maros_meszaros.hpp:121-140: keys P,q,A,l,u; equality rows are l==u).

The reference test (test/src/dense_maros_meszaros.cpp:97) runs every problem with n <= 1000 and
n_eq + n_in <= 1000: 62 problems.  They are committed in two files so that the GPU box, which has no
/root/reference, runs all of them:
  maros_meszaros_small.npz   n <= 150 and rows <= 320 (33 problems), dense arrays "<name>/<P|q|A|l|u>"
  maros_meszaros_medium.npz  the other 29, P and A as COO triplets "<name>/<P|A>_<row|col|val|shape>"
Run:  python tests/golden/make_maros_meszaros_fixtures.py
"""
import glob
import os

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

SRC = "/root/reference/test/data/maros_meszaros_data"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "maros_meszaros_small.npz")
OUT_MEDIUM = os.path.join(HERE, "maros_meszaros_medium.npz")
# beyond the reference test's own selection (it stops at 1000 rows): problems whose constraint rows exceed the 1024
# threads of the widest workgroup, for the chunked stages of the 1024-thread kernel
OUT_LARGE = os.path.join(HERE, "maros_meszaros_large.npz")
LARGE = ("GOULDQP2", "CVXQP2_M")


def load(path, any_size=False):
    d = sio.loadmat(path)
    if not any_size and (d["P"].shape[0] > 1000 or d["A"].shape[0] > 1000):
        return None
    dense = lambda m: m.toarray() if hasattr(m, "toarray") else np.asarray(m)
    P, A = dense(d["P"]).astype(np.float64), dense(d["A"]).astype(np.float64)
    q, l, u = (np.asarray(d[k], dtype=np.float64).ravel() for k in ("q", "l", "u"))
    return P, q, A, l, u


def load_medium(path=OUT_MEDIUM, only=None):
    """-> {name: (P, q, A, l, u)} with dense P and A rebuilt from the triplets"""
    d = np.load(path)
    out = {}
    for name in [str(s) for s in d["names"]]:
        if only is not None and name != only:
            continue
        mats = []
        for k in "PA":
            shape = tuple(int(v) for v in d["%s/%s_shape" % (name, k)])
            m = np.zeros(shape)
            m[d["%s/%s_row" % (name, k)], d["%s/%s_col" % (name, k)]] = d["%s/%s_val" % (name, k)]
            mats.append(m)
        out[name] = (mats[0], d[name + "/q"], mats[1], d[name + "/l"], d[name + "/u"])
    return out


def pack_coo(store, name, P, q, A, l, u):
    for k, m in (("P", P), ("A", A)):
        c = sp.coo_matrix(m)
        store["%s/%s_row" % (name, k)] = c.row.astype(np.int32)
        store["%s/%s_col" % (name, k)] = c.col.astype(np.int32)
        store["%s/%s_val" % (name, k)] = c.data
        store["%s/%s_shape" % (name, k)] = np.array(m.shape, dtype=np.int64)
    for k, v in zip("qlu", (q, l, u)):
        store["%s/%s" % (name, k)] = v


def main():
    large = {}
    for name in LARGE:
        pack_coo(large, name, *load(os.path.join(SRC, name + ".mat"), any_size=True))
    large["names"] = np.array(LARGE)
    np.savez_compressed(OUT_LARGE, **large)
    print("wrote", OUT_LARGE, len(LARGE), "problems", os.path.getsize(OUT_LARGE), "bytes")
    small, medium = {}, {}
    names_small, names_medium = [], []
    for f in sorted(glob.glob(os.path.join(SRC, "*.mat"))):
        r = load(f)
        if r is None:
            continue
        P, q, A, l, u = r
        name = os.path.basename(f)[:-4]
        if P.shape[0] <= 150 and A.shape[0] <= 320:
            names_small.append(name)
            for k, v in zip("PqAlu", (P, q, A, l, u)):
                small["%s/%s" % (name, k)] = v
        else:
            names_medium.append(name)
            for k, m in (("P", P), ("A", A)):
                c = sp.coo_matrix(m)
                medium["%s/%s_row" % (name, k)] = c.row.astype(np.int32)
                medium["%s/%s_col" % (name, k)] = c.col.astype(np.int32)
                medium["%s/%s_val" % (name, k)] = c.data
                medium["%s/%s_shape" % (name, k)] = np.array(m.shape, dtype=np.int64)
            for k, v in zip("qlu", (q, l, u)):
                medium["%s/%s" % (name, k)] = v
    small["names"] = np.array(names_small)
    medium["names"] = np.array(names_medium)
    np.savez_compressed(OUT, **small)
    np.savez_compressed(OUT_MEDIUM, **medium)
    print("wrote", OUT, len(names_small), "problems", os.path.getsize(OUT), "bytes")
    print("wrote", OUT_MEDIUM, len(names_medium), "problems", os.path.getsize(OUT_MEDIUM), "bytes")
    rt = load_medium()
    for f in sorted(glob.glob(os.path.join(SRC, "*.mat"))):
        name = os.path.basename(f)[:-4]
        if name in rt:
            P, q, A, l, u = load(f)
            assert np.array_equal(P, rt[name][0]) and np.array_equal(A, rt[name][2]) and np.array_equal(u, rt[name][4])
    print("round trip ok:", [(n, rt[n][0].shape[0], rt[n][2].shape[0]) for n in names_medium])


if __name__ == "__main__":
    main()
