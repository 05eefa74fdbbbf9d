#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Golden (x, y, z, iter, iter_ext, status, residuals) of the REAL ProxSuite dense
backend on the reference benchmark's QPs (seed i = QP i), produced by oracle/_ref/ref_batchqp
(oracle/ref/build_ref.sh: needs Eigen3) and stored as tests/golden/reference_<tag>.npz.

    python tests/golden/make_reference_fixtures.py

Runs in the authoring container only (it needs /root/reference and Eigen); the .npz files travel with
the repository and are what tests/test_reference_golden.py loads -- on the CPU for the oracle, on the
MI355X for the HIP path.  Without the binary it prints why and exits 0 (no fixture is written, the
tests skip, DESIGN.md says "parity unpinned against the ProxSuite binary").
"""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_batchqp")
# (tag, batch, n, n_eq, n_in): BASELINE.json configs[0] whole, configs[1] first 256 QPs
CASES = (("c1", 128, 50, 25, 50), ("c2_first256", 256, 100, 50, 100))


def main():
    subprocess.run([os.path.join(ROOT, "oracle", "ref", "build_ref.sh")], check=False)
    if not os.path.exists(BIN):
        print("make_reference_fixtures: %s is not there (Eigen3 absent?): no fixture written" % BIN)
        return 0
    for tag, B, n, ne, ni in CASES:
        tmp = os.path.join(HERE, "_ref_%s.bin" % tag)
        subprocess.run([BIN, "golden", str(B), str(n), str(ne), str(ni), tmp], check=True)
        raw = open(tmp, "rb").read()
        os.remove(tmp)
        hdr = struct.unpack("4q", raw[:32])
        assert hdr == (B, n, ne, ni), hdr
        rec = n + ne + ni + 5
        a = np.frombuffer(raw[32:], dtype=np.float64).reshape(B, rec)
        np.savez_compressed(os.path.join(HERE, "reference_%s.npz" % tag), n=n, n_eq=ne, n_in=ni,
                            x=a[:, :n], y=a[:, n:n + ne], z=a[:, n + ne:n + ne + ni],
                            iter=a[:, -5].astype(np.int64), iter_ext=a[:, -4].astype(np.int64),
                            status=a[:, -3].astype(np.int64), pri_res=a[:, -2], dua_res=a[:, -1])
        print("wrote reference_%s.npz (%d QPs)" % (tag, B))
    return 0


if __name__ == "__main__":
    sys.exit(main())
