"""Probe generation, the N+1 solves and the Hutchinson estimator (mirror of
reference runlmc/lmc/stochastic_deriv.py:12-78).

``StochasticDerivService(metrics, pool, n_it, tol).generate(K, y)`` keeps the
reference's signature; ``pool`` is accepted and ignored (the solves are one
batched device call), and an optional torch.distributed ``group`` shards the
probes over GPUs.
"""
import ctypes
import threading

import numpy as np
import torch

from .. import _lib

from .derivative import Derivative
from ..approx.iterative import Iterative
from ..util.dist import rank_world, shard_rows, all_reduce_sum_, broadcast_


def _host_cores():
    import os
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


class StochasticDerivService:
    # Lanczos steps kept per system for the log-determinant quadrature.  None: as many as a
    # solve may run (maxiter, else n), at most LANCZOS_MAX -- until round 5 a fixed 256 cut the
    # quadrature below the 400-500 iterations a C5 solve runs.  (16 bytes per step and system.)
    LANCZOS_CAP = None
    LANCZOS_MAX = 4096

    def __init__(self, metrics, pool, n_it, tol, group=None, scipy_exits=None, maxiter=0,
                 precondition=None, device_probes=None):
        # (the first four arguments are the reference's; scipy_exits / maxiter / precondition go
        # to Iterative.solve_device unchanged: None = Iterative.SCIPY_EXITS, 0 = n iterations,
        # None = Iterative.PRECONDITION)
        self.metrics = metrics
        self._pool = pool          # interface compatibility only
        self._n_it = int(n_it)
        self._tol = tol
        self._group = group
        self._scipy_exits = scipy_exits
        self._maxiter = int(maxiter)
        self._precondition = precondition
        self._side = None          # copy stream and device-side byte buffer of the probe hand-over
        self._dev8 = None
        # device_probes: None -- probes come from NumPy's global legacy RNG exactly as the
        # reference draws them (stochastic_deriv.py:35; 0.6 s of host time for C5's 128 x 10^6
        # int64 matrix, twenty times the step that uses them); an int -- a seed: every call of
        # generate() without explicit probes draws them ON the device (torch's generator, seed +
        # call count, the same on every rank): the same estimator, another stream of draws
        self._device_probes = device_probes
        self._draws = 0
        self._stage = None         # pinned host bytes of the narrowed probes (kept between steps)

    def draw_probes(self, n):
        """+-1 probes from NumPy's legacy global RNG exactly as the reference
        draws them (stochastic_deriv.py:35); every rank draws the same
        matrix (seed the RNG identically) and keeps its own rows."""
        return np.random.randint(0, 2, (self._n_it, n)) * 2 - 1

    def draw_probes_device(self, n, device, seed=None):
        """+-1 probes drawn ON the device (int8, n_it x n; torch's generator, not NumPy's
        stream: the same estimator, another draw).  Every rank must pass the same seed."""
        gen = torch.Generator(device=device)
        if seed is not None:
            gen.manual_seed(int(seed))
        bits = torch.randint(0, 2, (self._n_it, n), dtype=torch.int8, device=device, generator=gen)
        return bits * 2 - 1

    def generate(self, K, y, rs=None):
        """Solve K alpha = y and K s_i = r_i for this rank's probes.  `rs`
        (n_it x n, entries +-1) may be passed explicitly (parity tests)."""
        n = K.shape[0]
        if rs is None and self._device_probes is not None:
            rs = self.draw_probes_device(n, K.device, seed=int(self._device_probes) + self._draws)
            self._draws += 1
        if rs is None:
            rs = self.draw_probes(n)
        # (a torch tensor on the operator's device is taken as it is -- +-1 entries of any
        # dtype, e.g. drawn there with draw_probes_device: no host pass over N x n entries,
        # which at C5 costs ten times what the solves through the factorisation do)
        on_device = isinstance(rs, torch.Tensor)
        if not on_device:
            rs = np.asarray(rs)
        if tuple(rs.shape) != (self._n_it, n):
            raise ValueError('probes must have shape {}'.format((self._n_it, n)))
        mine = shard_rows(self._n_it, self._group)
        dev = K.device
        # Every rank solves for alpha next to its own probes (no rank waits
        # for another during the solve); y rides in a transform pair of its
        # own -- last in an odd batch, or next to a zero vector -- so that its
        # roundoff does not depend on a probe.  That alone does not make the
        # bits equal across ranks: which kernels a batch runs on (polynomial /
        # filter forms above a batch gate, fused products below a size,
        # projection chunk lengths) depends on the batch size, and ranks
        # whose probe counts differ by one carry batches that differ by two.
        # Rank 0's alpha is therefore BROADCAST after the solve (n doubles,
        # one collective) and its Gram terms ride in the gradient's one
        # all-reduce (likelihood.py): every rank ends the step with the same
        # bits by construction, whatever the shard sizes.
        # The right-hand sides are put together ON THE DEVICE: +-1 probes drawn as the
        # reference draws them are an int64 matrix (1 GB at C5) -- converted to float64
        # and stacked on the host they cost 0.4 s of a 1.9 s step on the GPU box's
        # cores; they cross as one byte per entry and are widened there.
        nm = len(mine)
        rank, world = rank_world(self._group)
        mine_rows = rs[rank::world]                               # (a view: rows rank, rank + world, ...)
        if nm % 2 == 0:
            nrow, first, order = nm + 1, 0, [nm] + list(range(nm))    # y alone in the last pair
            yat = nm
        else:
            nrow, first, order = nm + 2, 2, [0] + list(range(2, 2 + nm))   # y paired with zeros
            yat = 0
        Bfull = torch.zeros((nrow, n), dtype=torch.float64, device=dev)
        Bfull[yat] = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(dev)
        if nm and on_device:
            Bfull[first:first + nm] = mine_rows.to(device=dev, dtype=torch.float64)
        elif nm:
            # +-1 probes as the reference draws them are an int64 matrix (1 GB at C5).  ONE
            # pass over it on the host's cores (the library's helper: every entry checked to be
            # +-1 BEFORE it is narrowed -- 255 or 257 would wrap to -1 / +1 in one byte) into a
            # pinned staging buffer kept on the service, one byte per entry across the bus,
            # widened on the device.  Anything else (another dtype, other values): the plain way.
            narrow = None
            if (mine_rows.dtype == np.int64 and mine_rows.ndim == 2 and mine_rows.strides[1] == 8
                    and mine_rows.strides[0] % 8 == 0 and mine_rows.strides[0] > 0):
                stage = self._stage
                if stage is None or stage.numel() < nm * n:
                    stage = torch.empty(nm * n, dtype=torch.int8)
                    if dev.type == 'cuda':
                        stage = stage.pin_memory()
                    self._stage = stage
                lib = _lib.get_library()
                # The pass over the probes is host work only (ctypes drops the GIL around it) and the
                # bytes cross the bus on a stream of their own, a few rows' worth at a time: while
                # both run, this thread has the operator's forms verified and its factorisation
                # built -- what the solve below would otherwise start with (C5: ~5 ms of a 30 ms step).
                on_gpu = dev.type == 'cuda'
                # (stream and device-side byte buffer are the service's, like the pinned one: a
                # fresh 128 MB block and a fresh stream per step cost the first steps of a fit
                # 20 ms each until the allocator's pool had grown)
                side = None
                if on_gpu:
                    if self._side is None or self._side.device != dev:
                        self._side = torch.cuda.Stream(device=dev)
                    side = self._side
                    side.wait_stream(torch.cuda.current_stream(dev))   # (the last step's widening read it)
                if self._dev8 is None or self._dev8.device != dev or self._dev8.numel() < nm * n:
                    self._dev8 = torch.empty(nm * n, dtype=torch.int8, device=dev)
                dev8 = self._dev8[:nm * n].view(nm, n)
                pieces = [(a, min(nm, a + max(1, (nm + 3) // 4))) for a in range(0, nm, max(1, (nm + 3) // 4))]
                state = {'ok': True, 'error': None}

                def narrow_on_host():
                    try:
                        for a, b in pieces:
                            ok = ctypes.c_int()
                            lib.call('rl_probes_to_int8',
                                     ctypes.c_void_p(mine_rows.ctypes.data + a * mine_rows.strides[0]), b - a,
                                     mine_rows.strides[0] // 8, n, ctypes.c_void_p(stage.data_ptr() + a * n),
                                     max(1, min(_host_cores(), 32)), ctypes.byref(ok))
                            if not ok.value:
                                state['ok'] = False
                                return
                            src = stage[a * n:b * n].view(b - a, n)
                            if on_gpu:
                                with torch.cuda.stream(side):
                                    dev8[a:b].copy_(src, non_blocking=True)
                            else:
                                dev8[a:b].copy_(src)
                    except Exception as e:              # (re-raised on the caller's thread)
                        state['error'] = e
                worker = threading.Thread(target=narrow_on_host)
                worker.start()
                try:
                    if Iterative.PRECONDITION if self._precondition is None else self._precondition:
                        getattr(K, 'preconditioner', None)
                finally:
                    worker.join()
                if state['error'] is not None:
                    raise state['error']
                if on_gpu:
                    torch.cuda.current_stream(dev).wait_stream(side)
                if state['ok']:
                    narrow = dev8
            if narrow is not None:
                Bfull[first:first + nm] = narrow
            else:
                Bfull[first:first + nm] = torch.from_numpy(
                    np.ascontiguousarray(mine_rows, dtype=np.float64)).to(dev)
        Xf, iters, resid, istop, lanczos = Iterative.solve_device(
            K, Bfull, minres=True, tol=self._tol, maxiter=self._maxiter,
            lanczos_cap=(self.LANCZOS_CAP or min(self._maxiter or n, self.LANCZOS_MAX)),
            scipy_exits=self._scipy_exits,
            precondition=self._precondition)
        idx = torch.tensor(order, device=dev)
        X, B = Xf[idx], Bfull[idx]
        iters, resid, istop = (np.asarray(a)[order] for a in (iters, resid, istop))
        # (no Lanczos coefficients when the operator's preconditioner answered -- its
        # factorisation holds log det K~ exactly instead)
        logdet_exact = None
        logdet_fn = None
        if lanczos is None:
            M = K.preconditioner
            if M is not None and M.exact:
                logdet_exact = M.logdet()
            elif M is not None:
                # (preconditioned CG through an INEXACT factorisation: no Lanczos recurrence of K~
                # ran and the factorisation's log det is not the operator's -- asked for, the log
                # det comes from a few extra preconditioned solves, FactoredInverse.logdet_estimate)
                tol, grp = self._tol, self._group
                logdet_fn = lambda: M.logdet_estimate(tol=tol, group=grp)    # noqa: E731
        else:
            lanczos = lanczos[order]
        if self.metrics is not None:
            # mean over the N+1 systems; alpha (solved everywhere) counted once
            lo = 0 if rank == 0 else 1
            stats = torch.tensor([float(np.sum(iters[lo:])),
                                  float(np.sum(resid[lo:]))], dtype=torch.float64)
            all_reduce_sum_(stats, self._group)
            self.metrics.iterations.append(float(stats[0]) / (self._n_it + 1))
            self.metrics.solv_error.append(float(stats[1]) / (self._n_it + 1))
        alpha = X[0].clone()
        broadcast_(alpha, src=0, group=self._group)        # (no-op in a world of one)
        return StochasticDeriv(alpha, B[1:], X[1:], self._n_it, group=self._group,
                               iterations=iters, residuals=resid, istop=istop,
                               lanczos=lanczos, logdet_exact=logdet_exact, logdet_fn=logdet_fn)

    def _concurrent_solve(self, ls):
        """Reference entry point (stochastic_deriv.py:51-52): a list of
        (K, rhs, verbose, minres, tol) tuples sharing one K."""
        K = ls[0][0]
        verbose, minres, tol = ls[0][2], ls[0][3], ls[0][4]
        out = Iterative.solve(K, np.vstack([t[1] for t in ls]), verbose=True,
                              minres=minres, tol=tol)
        X, iters, resid = out
        if verbose:
            return [(X[i], int(iters[i]), float(resid[i])) for i in range(len(ls))]
        return list(X)


class StochasticDeriv(Derivative):
    """alpha = K^-1 y, this rank's probes r_i and K^-1 r_i, all on the device.

    ``n_it`` is the GLOBAL probe count (the 1/N of the estimator)."""

    def __init__(self, alpha, rs, inv_rs, n_it, group=None, iterations=None,
                 residuals=None, istop=None, lanczos=None, logdet_exact=None, logdet_fn=None):
        to_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(a, dtype=np.float64))
        self.alpha_dev = to_t(alpha)
        self.rs_dev = to_t(rs).to(self.alpha_dev.device)
        self.inv_rs_dev = to_t(inv_rs).to(self.alpha_dev.device)
        self._n_it = int(n_it)
        self._group = group
        self.iterations, self.residuals, self.istop = iterations, residuals, istop
        self.lanczos = lanczos
        self.logdet_exact = logdet_exact
        self._logdet_fn = logdet_fn          # () -> (estimate, sem, iterations), run on first use
        self.logdet_precond = None           # its result, once asked for
        self._alpha_host = None

    @property
    def alpha(self):
        if self._alpha_host is None:
            self._alpha_host = self.alpha_dev.cpu().numpy()
        return self._alpha_host

    @property
    def _rs(self):
        return self.rs_dev.cpu().numpy()

    @property
    def _inv_rs(self):
        return self.inv_rs_dev.cpu().numpy()

    def logdet_probe_estimates(self):
        """Per-probe stochastic-Lanczos-quadrature values r_i^T log(K) r_i of
        THIS rank's probes, from the Lanczos tridiagonals the probe solves
        built (no extra operator products)."""
        if self.lanczos is None:
            raise ValueError('no Lanczos coefficients were recorded (the solves ran preconditioned: '
                             'pass precondition=False for a Lanczos-quadrature log det)')
        from .._native import slq_quadratic_forms
        n = self.alpha_dev.shape[0]
        its = np.asarray(self.iterations)[1:]
        sq = np.full(len(its), float(n))         # ||r||^2 = n for +-1 probes
        return slq_quadratic_forms(self.lanczos[1:], its, sq)

    def logdet_K(self):
        """log det K: exact (determinant lemma, rl_ski_factor) when the solves went
        through the operator's factorisation; when that factorisation was a preconditioner, its
        exact log det plus the preconditioned Lanczos-quadrature estimate of the rest (a few extra
        solves, run on first use); else the Hutchinson + Lanczos-quadrature estimate from the
        probe solves (mean over all ranks' probes)."""
        if self.logdet_exact is not None:
            return self.logdet_exact
        if self.lanczos is None and self._logdet_fn is not None:
            if self.logdet_precond is None:
                self.logdet_precond = self._logdet_fn()
            return self.logdet_precond[0]
        local = self.logdet_probe_estimates()
        tot = torch.tensor([float(local.sum())], dtype=torch.float64)
        all_reduce_sum_(tot, self._group)
        return float(tot[0]) / self._n_it

    # generic operator form, any Matrix dKdt (reference :69-78)
    def d_normal_quadratic(self, dKdt):
        return float(self.alpha.dot(dKdt.matvec(self.alpha)))

    def d_logdet_K(self, dKdt):
        rs, inv = self._rs, self._inv_rs
        local = 0.0
        if len(rs):
            local = float(np.einsum('ij,ij->', inv, dKdt.matmat(rs.T.astype(float)).T))
        tot = torch.tensor([local], dtype=torch.float64)
        all_reduce_sum_(tot, self._group)
        return float(tot[0]) / self._n_it
