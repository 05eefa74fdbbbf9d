"""QPFunction: the forward of `proxsuite.torch.qplayer.QPFunction` on MI355X.

Mirrors reference bindings/python/proxsuite/torch/qplayer.py:12-167 (feasible QPs) and
:255-369 (closest-feasible QPs, `structural_feasibility=False`): same factory arguments,
same solver settings (max_iter = maxIter, max_iter_in = 100, rho = 5e-5 with
refactor_rho_threshold = rho, eps_abs = eps), same outputs.  The reference loops over the
batch in Python, copies every matrix to numpy, and calls init/solve per QP; here the whole
batch goes through ONE pqp_batch_init (pointers of the torch tensors, host or ROCm, are
handed to the C-ABI as they are), ONE solve launch with one workgroup per QP, and the
results are copied device-to-device into the output tensors.

The backward pass (reference qplayer.py:172-253) is ONE pqp_batch_backward launch (the device
form of dense/compute_ECJ.hpp's compute_backward for every QP of the batch) whose seven jacobians
are copied device-to-device into the gradient tensors.  The closest-feasible variant
(`structural_feasibility=False`) differentiates through the sparse backend in the reference
(qplayer.py:371-552); that path is outside this repository's scope and raises.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _native
from .._ctypes_defs import DenseBackend, HessianType, InitialGuess


def _extract_nbatch(*params_and_dims):
    for param, dim in params_and_dims:
        if param.ndimension() == dim:
            return param.size(0)
    return 1


def _expand(X, nbatch, ndim):
    # reference bindings/python/proxsuite/torch/utils.py:57-63
    if X.ndimension() in (0, ndim) or X.nelement() == 0:
        return X
    if X.ndimension() == ndim - 1:
        return X.unsqueeze(0).expand(*([nbatch] + list(X.size())))
    raise RuntimeError("Unexpected number of dimensions.")


def _dense64(t):
    return t.detach().to(torch.float64).contiguous()


# Native batch handles are kept per (batch, dim, n_eq, n_in, device) signature: creating one costs
# ~45 device allocations + memsets (25 ms at 2048 x (100, 50, 100), against 10 ms for the solve).  A
# forward checks a handle out; it goes back when the autograd context that owns it dies (after the
# backward, or with the graph), so two layers of one shape in the same graph never share a handle.
_FREE = {}
_KEEP = 4


def _checkout(key):
    free = _FREE.setdefault(key, [])
    if free:
        batch = free.pop()
        # a reused handle must look like the fresh QP objects the reference builds at every forward:
        # QP::cleanup() resets results and the proximal parameters (the previous backward left
        # rho = rho_backward, mu = mu_backward in results.info, and init(..., rho=...) keeps mu)
        batch.cleanup(-1)
        return batch
    nbatch, nz, neq, nineq, index = key
    return _native.Batch(nbatch, nz, neq, nineq, box_constraints=False, hessian_type=int(HessianType.Dense),
                         dense_backend=int(DenseBackend.Automatic), device=index)


def _give_back(key, batch):
    free = _FREE.setdefault(key, [])
    if len(free) < _KEEP:
        free.append(batch)
    else:
        batch.close()


class _Lease:
    """ties a checked-out handle to the lifetime of its autograd context"""

    def __init__(self, key, batch):
        self.key, self.batch = key, batch

    def __del__(self):
        try:
            _give_back(self.key, self.batch)
        except Exception:
            pass


def _solve_batch(Q, p, A, b, G, l, u, eps, max_iter, infeasible):
    nbatch, nineq, nz = G.size()
    neq = A.size(1) if A.nelement() > 0 else 0
    assert neq > 0 or nineq > 0
    dev = Q.device
    index = dev.index if dev.type == "cuda" and dev.index is not None else 0
    key = (int(nbatch), int(nz), int(neq), int(nineq), int(index))
    batch = _checkout(key)
    lease = _Lease(key, batch)
    rho = 5.0e-5
    # (one vectorised write per field over the handle's settings table: the per-QP Python loop cost ~5 ms at 2048 QPs)
    batch.set_all_settings(primal_infeasibility_solving=int(infeasible), max_iter=max_iter, max_iter_in=100,
                           default_rho=rho, refactor_rho_threshold=rho,  # no refactorization
                           eps_abs=eps,
                           initial_guess=int(InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS))  # (restated for reused handles)
    if dev.type == "cuda":
        # kernels go to the caller's stream; the model copies of init are blocking copies on the
        # null stream, which are ordered after torch's default stream -- only a side stream with
        # inputs still in flight needs an explicit wait
        cur = torch.cuda.current_stream(dev)
        batch.set_stream(cur.cuda_stream)
        if cur != torch.cuda.default_stream(dev):
            cur.synchronize()
    batch.init(-1, _dense64(Q), _dense64(p), _dense64(A) if neq else None, _dense64(b) if neq else None,
               _dense64(G) if nineq else None, _dense64(l) if nineq else None, _dense64(u) if nineq else None,
               rho=rho)
    batch.solve()
    opts = dict(dtype=torch.float64, device=dev)
    x = torch.empty((nbatch, nz), **opts)
    y = torch.empty((nbatch, neq), **opts)
    z = torch.empty((nbatch, nineq), **opts)
    se = torch.empty((nbatch, neq), **opts)
    si = torch.empty((nbatch, nineq), **opts)
    batch.results_into(x, y, z, se, si)
    return lease, x, y, z, se, si


def QPFunction(eps=1e-9, maxIter=1000, eps_backward=1.0e-4, rho_backward=1.0e-6, mu_backward=1.0e-6,
               omp_parallel=False, structural_feasibility=True):
    """Factory with the reference's signature (qplayer.py:12-20).  `omp_parallel` is accepted and
    ignored: the batch is always solved in one launch."""

    class QPFunctionFn(Function):
        @staticmethod
        def forward(ctx, Q_, p_, A_, b_, G_, l_, u_):
            nbatch = _extract_nbatch((Q_, 3), (p_, 2), (A_, 3), (b_, 2), (G_, 3), (l_, 2), (u_, 2))
            Q, p, G = _expand(Q_, nbatch, 3), _expand(p_, nbatch, 2), _expand(G_, nbatch, 3)
            u, l = _expand(u_, nbatch, 2), _expand(l_, nbatch, 2)
            A, b = _expand(A_, nbatch, 3), _expand(b_, nbatch, 2)
            lease, x, y, z, _, _ = _solve_batch(Q, p, A, b, G, l, u, eps, maxIter, infeasible=False)
            ctx.lease = lease  # the handle returns to the cache when this context is collected
            ctx.batch = lease.batch
            ctx.dev = Q.device
            ctx.dtype = Q.dtype
            ctx.shapes = tuple(tuple(t_.shape) if t_.numel() else () for t_ in (Q_, p_, A_, b_, G_, l_, u_))
            ctx.batched = (Q_.ndimension() == 3, p_.ndimension() == 2, A_.ndimension() == 3, b_.ndimension() == 2,
                           G_.ndimension() == 3, l_.ndimension() == 2, u_.ndimension() == 2)
            return x.to(Q.dtype), y.to(Q.dtype), z.to(Q.dtype)

        @staticmethod
        def backward(ctx, dl_dzhat, dl_dlams, dl_dnus):
            batch, dev = ctx.batch, ctx.dev
            B, n, ne, ni = batch.B, batch.n, batch.n_eq, batch.n_in
            ld = torch.zeros((B, n + ne + ni), dtype=torch.float64, device=dev)
            ld[:, :n] = dl_dzhat
            if dl_dlams is not None and ne:
                ld[:, n:n + ne] = dl_dlams
            if dl_dnus is not None and ni:
                ld[:, n + ne:] = dl_dnus
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            batch.backward(ld, eps_backward, rho_backward, mu_backward)
            opts = dict(dtype=torch.float64, device=dev)
            out = dict(dL_dH=torch.empty((B, n, n), **opts), dL_dg=torch.empty((B, n), **opts),
                       dL_dA=torch.empty((B, ne, n), **opts), dL_db=torch.empty((B, ne), **opts),
                       dL_dC=torch.empty((B, ni, n), **opts), dL_du=torch.empty((B, ni), **opts),
                       dL_dl=torch.empty((B, ni), **opts))
            batch.backward_results(-1, into=out)
            t = ctx.dtype

            def shaped(g, like_batched, shape):
                if g.numel() == 0 or len(shape) == 0:
                    return None
                # parameters shared by the batch receive the sum of the per-QP gradients
                return (g if like_batched else g.sum(dim=0)).to(t).reshape(shape)

            bt, sh = ctx.batched, ctx.shapes
            names = ("dL_dH", "dL_dg", "dL_dA", "dL_db", "dL_dC", "dL_dl", "dL_du")
            return tuple(shaped(out[k], bt[i], sh[i]) for i, k in enumerate(names))

    class QPFunctionFn_infeas(Function):
        @staticmethod
        def forward(ctx, Q_, p_, A_, b_, G_, l_, u_):
            n_in, nz = G_.size()[-2:]
            nbatch = _extract_nbatch((Q_, 3), (p_, 2), (A_, 3), (b_, 2), (G_, 3), (l_, 2), (u_, 2))
            Q, p, G = _expand(Q_, nbatch, 3), _expand(p_, nbatch, 2), _expand(G_, nbatch, 3)
            u, l = _expand(u_, nbatch, 2), _expand(l_, nbatch, 2)
            A, b = _expand(A_, nbatch, 3), _expand(b_, nbatch, 2)
            # single-sided restatement, as the reference does (qplayer.py:270-271)
            h = torch.cat((-l, u), dim=1)
            G1 = torch.cat((-G, G), dim=1)
            lo = torch.full_like(h, -1.0e20)
            lease, x, y, z, se, si = _solve_batch(Q, p, A, b, G1, lo, h, eps, maxIter, infeasible=True)
            ctx.lease = lease
            nus_sol = -z[:, :n_in] + z[:, n_in:]
            s_i = -si[:, :n_in] + si[:, n_in:]
            t = Q.dtype
            return x.to(t), y.to(t), nus_sol.to(t), se.to(t), s_i.to(t)

        @staticmethod
        def backward(ctx, dl_dzhat, dl_dlams, dl_dnus, dl_ds_e, dl_ds_i):
            raise NotImplementedError(
                "the backward of the closest-feasible QPFunction differentiates through the sparse backend in "
                "the reference (qplayer.py:371-552), which is outside this repository's scope")

    return QPFunctionFn.apply if structural_feasibility else QPFunctionFn_infeas.apply
