"""QPLayer forward + backward throughput on MI355X at the C2 shape (2048 QPs, n=100, n_eq=50,
n_in=100): proxsuite_amd.torch.QPFunction on ROCm tensors, loss = sum(x)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from proxsuite_amd.torch import QPFunction
from proxsuite_amd.utils import random_qp as R

B, n, ne, ni = 2048, 100, 50, 100
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
dev = "cuda"
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
Q, p, A, b, G, u = t(m.H), t(m.g).requires_grad_(True), t(m.A), t(m.b), t(m.C), t(m.u)
l = torch.full_like(u, -1e20)
f = QPFunction(eps=1e-9, maxIter=1000)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, lam, nu = f(Q, p, A, b, G, l, u)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    x.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d: forward %.2f ms (%.0f QPs/s incl. batch create + init + Ruiz), backward %.2f ms (%.0f QPs/s)"
          % (rep, 1e3 * (t1 - t0), B / (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t1)), flush=True)
    p.grad = None
