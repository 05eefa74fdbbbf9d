import os, sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
import parity_cases as pc
lib = N.load()
for (n,ne,ni,B) in ((1500,300,600,2),(60,10,1500,2),(40,0,2100,1)):
    m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    pc.settings_all(b, eps_abs=1e-9, eps_rel=0, initial_guess=0)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    t0=time.time(); b.solve(); t1=time.time()
    x,y,z,se,si,info=b.results()
    t2=time.time()
    qs = pc.oracle_solve_many(O, [(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i]) for i in range(B)], n, ne, ni)
    t3=time.time()
    for i in range(B):
        q=qs[i].results
        pri,dua=pc.kkt(O,m,i,x[i],y[i],z[i])
        print((n,ne,ni), "nt", b.launch_config(), "status", info[i].status, q.info.status, "iter", info[i].iter, q.info.iter, "kkt %.1e %.1e"%(pri,dua),
              "dx %.1e dz %.1e"%(np.abs(x[i]-q.x).max(), np.abs(z[i]-q.z).max()), "gpu %.2fs oracle %.2fs"%(t1-t0,(t3-t2)/B))
