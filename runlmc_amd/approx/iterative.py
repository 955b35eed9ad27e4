"""Krylov solve front end (mirror of reference
runlmc/approx/iterative.py:19-62), batched on the device.

``Iterative.solve(K, y, verbose=False, minres=True, tol=1e-4)`` keeps the
reference's contract: inner method tolerance min(1e-10, tol), at most n
iterations, an explicit ``||y - K x|| < tol`` exit every 100 iterations, a
CRITICAL log (not an exception) on non-convergence.  ``y`` may also be an
(k, n) block of right-hand sides (rows), which is how the N+1 solves of a
gradient step are issued as ONE call (rl_solve_batch)."""
import logging

import numpy as np
import torch

from .._native import solve_batch, MINRES, CG, MINRES_RULE
from .._lib import as_f64

_LOG = logging.getLogger(__name__)


def _device_operator(K):
    op = getattr(K, 'device_operator', None)
    if op is None:
        raise TypeError(
            'Iterative.solve needs a device-resident operator (the result of '
            'runlmc_amd.lmc.grid_kernel.gen_grid_kernel); got {}. runlmc_amd '
            'has no host solver.'.format(type(K).__name__))
    return op() if callable(op) else op


class Iterative:
    """Target solve() tolerance. Only errors > tol reported."""

    CHECK_EVERY = 100      # reference iterative.py:39
    # SciPy's minres ends a solve on its own tests (||r|| <= rtol ||A|| ||x|| with the
    # running Frobenius-like ||A||, Acond, ...) -- in SciPy 1.15 long before the
    # reference's residual rule on ill-conditioned kernels.  The reference's published
    # logs end on the rule (counts are multiples of 100, residuals < 1e-4).  False runs
    # MINRES with those tests off (RL_MINRES_RULE): the rule or n iterations end a solve.
    SCIPY_EXITS = True
    # The reference hands ``getattr(K, 'preconditioner', None)`` to SciPy's method
    # (iterative.py:47-51); no reference operator has one.  The device operator has, when
    # all its top rows are in the polynomial form: K~^-1 itself through the Woodbury
    # identity (csrc/rl_direct.h), with which a preconditioned iteration is iterative
    # refinement -- ended by the reference's own residual rule after one or two steps,
    # where the unpreconditioned solve of the benchmark systems never meets it.  False
    # (or ``precondition=False`` per call) ignores the attribute: the Krylov solve as before.
    PRECONDITION = True

    @staticmethod
    def solve_device(K, B, minres=True, tol=1e-4, maxiter=0, lanczos_cap=0, scipy_exits=None,
                     precondition=None):
        """B: (k, n) float64 tensor on the operator's device.  Returns
        (X tensor, iterations, residuals, istop[, lanczos]) without the
        vectors leaving the GPU (lanczos is None when the operator's preconditioner
        answered: no Krylov recurrence ran)."""
        ski = _device_operator(K)
        if precondition is None:
            precondition = Iterative.PRECONDITION
        if precondition:
            M = getattr(K, 'preconditioner', None)          # reference iterative.py:47
            if M is not None:
                out = M.solve(B.contiguous(), tol=tol, maxiter=maxiter)
                return out + (None,) if lanczos_cap > 0 else out
        if scipy_exits is None:
            scipy_exits = Iterative.SCIPY_EXITS
        method = CG if not minres else (MINRES if scipy_exits else MINRES_RULE)
        return solve_batch(ski, B.contiguous(), method,
                           tol=tol, check_every=Iterative.CHECK_EVERY,
                           maxiter=maxiter, lanczos_cap=lanczos_cap)

    @staticmethod
    def solve(K, y, verbose=False, minres=True, tol=1e-4, scipy_exits=None, precondition=None):
        ski = _device_operator(K)
        y = as_f64(y)
        single = y.ndim == 1
        B = torch.from_numpy(np.atleast_2d(y)).to(ski.device)
        if B.shape[1] != K.shape[0]:
            raise ValueError('right-hand side has length {}, operator is {}'
                             .format(B.shape[1], K.shape))
        X, iters, resid, istop = Iterative.solve_device(K, B, minres, tol, scipy_exits=scipy_exits,
                                                        precondition=precondition)
        X = X.cpu().numpy()
        n = K.shape[0]
        for r, code in zip(resid, istop):
            if r > tol or code == 6:
                _LOG.critical('MINRES (n = %d) did not converge in n iterations.'
                              ' Reconstruction error %e', n, r)
        if single:
            return (X[0], int(iters[0]), float(resid[0])) if verbose else X[0]
        return (X, iters, resid) if verbose else X

    @staticmethod
    def solve_sharded(K, B, minres=True, tol=1e-4, group=None):
        """A block of right-hand sides split over the ranks of `group` (every
        rank holds a replica of the operator; rows rank, rank + world, ... are
        solved locally, no communication during the solve), results put together
        on every rank with ONE all-gather (each rank sends its own rows once) --
        SURVEY section 8e, BASELINE config 4 (block of 8 right-hand sides over
        1-4 GPUs).  B: (k, n) array.  Returns (X, iterations, residuals), all (k, .)."""
        from ..util.dist import shard_rows, rank_world, all_gather_rows
        ski = _device_operator(K)
        B = np.atleast_2d(as_f64(B))
        k, n = B.shape
        if n != K.shape[0]:
            raise ValueError('right-hand side has length {}, operator is {}'
                             .format(n, K.shape))
        rank, world = rank_world(group)
        mine = shard_rows(k, group)
        # solution rows with their iteration count and residual in two extra columns
        local = torch.zeros((len(mine), n + 2), dtype=torch.float64, device=ski.device)
        if mine:
            Bl = torch.from_numpy(B[mine]).to(ski.device)
            Xl, iters, resid, _ = Iterative.solve_device(K, Bl, minres, tol)
            local[:, :n] = Xl
            local[:, n] = torch.from_numpy(np.asarray(iters, dtype=np.float64)).to(ski.device)
            local[:, n + 1] = torch.from_numpy(np.asarray(resid, dtype=np.float64)).to(ski.device)
        counts = [len(range(r, k, world)) for r in range(world)]
        full = all_gather_rows(local, counts, group).cpu().numpy()
        # rank r's rows are r, r + world, ...: undo the interleave
        order = np.concatenate([np.arange(r, k, world) for r in range(world)]).astype(np.int64)
        out = np.empty_like(full)
        out[order] = full
        return out[:, :n], out[:, n].astype(np.int64), out[:, n + 1]
