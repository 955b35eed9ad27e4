"""Oracle: the reference's Krylov solve wrapper and the Krylov methods under it.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The reference calls scipy.sparse.linalg.minres / cg (third-party; unpinned in
the reference's setup.py:11-19; SciPy 1.15.3 in this image).  Two things live
here:

* ``minres_ps`` / ``cg_hs``: NumPy restatements of the published algorithms
  (Paige & Saunders 1975 MINRES with the stopping tests of SciPy's
  ``_isolve/minres.py``; Hestenes-Stiefel CG with SciPy's ``rtol*||b||``
  test).  These are what the HIP batched solver is compared with iterate by
  iterate.  tests/test_oracle_golden.py checks them against SciPy itself and against
  the reference's stored iterates.
* ``iterative_solve``: the reference wrapper's stopping rule on top
  (reference runlmc/approx/iterative.py:23-62): inner tolerance
  min(1e-10, tol), maxiter n, and every 100th iteration an explicit residual
  ||y - K x||_2 < tol exit.
"""
import numpy as np

_EPS = np.finfo(np.float64).eps


def minres_ps(matvec, b, rtol=1e-10, maxiter=None, callback=None, own_exits=True):
    """MINRES for symmetric A, x0 = 0, no preconditioner, no shift.

    Returns (x, info, itn, istop).  Follows SciPy 1.15.3
    scipy/sparse/linalg/_isolve/minres.py statement by statement (same
    operation order, so iterates agree with SciPy's to roundoff).

    own_exits=False switches SciPy's own stopping tests off (istop 1-4: test1 /
    test2 against rtol and against 1, Acond, epsx) so that only the caller's
    callback or maxiter ends the iteration -- the mode in which the reference's
    rule (iterative.py:36-42: explicit residual < tol every 100th iteration) is
    the one that stops a solve, as in its published logs
    (benchmarks/representation-cmp/out/inv-run-1.txt: counts are multiples of
    100, residuals < 1e-4)."""
    b = np.asarray(b, dtype=np.float64)
    n = b.shape[0]
    if maxiter is None:
        maxiter = 5 * n
    x = np.zeros(n)
    r1 = b.copy()
    y = r1
    beta1 = float(r1 @ y)
    if beta1 == 0:
        return x, 0, 0, 0
    beta1 = np.sqrt(beta1)

    oldb, beta, dbar, epsln = 0.0, beta1, 0.0, 0.0
    phibar, rhs1, rhs2, tnorm2 = beta1, beta1, 0.0, 0.0
    gmax, gmin = 0.0, np.finfo(np.float64).max
    cs, sn = -1.0, 0.0
    w = np.zeros(n)
    w2 = np.zeros(n)
    r2 = r1
    istop, itn = 0, 0
    while itn < maxiter:
        itn += 1
        v = (1.0 / beta) * y
        y = matvec(v)
        if itn >= 2:
            y = y - (beta / oldb) * r1
        alfa = float(v @ y)
        y = y - (alfa / beta) * r2
        r1 = r2
        r2 = y
        oldb = beta
        beta = float(r2 @ y)
        if beta < 0:
            raise ValueError('non-symmetric matrix')
        beta = np.sqrt(beta)
        tnorm2 += alfa ** 2 + oldb ** 2 + beta ** 2
        if itn == 1 and beta / beta1 <= 10 * _EPS:
            istop = -1

        oldeps = epsln
        delta = cs * dbar + sn * alfa
        gbar = sn * dbar - cs * alfa
        epsln = sn * beta
        dbar = -cs * beta
        root = np.hypot(gbar, dbar)

        gamma = max(np.hypot(gbar, beta), _EPS)
        cs = gbar / gamma
        sn = beta / gamma
        phi = cs * phibar
        phibar = sn * phibar

        denom = 1.0 / gamma
        w1 = w2
        w2 = w
        w = (v - oldeps * w1 - delta * w2) * denom
        x = x + phi * w

        gmax = max(gmax, gamma)
        gmin = min(gmin, gamma)
        z = rhs1 / gamma
        rhs1 = rhs2 - delta * z
        rhs2 = -epsln * z

        Anorm = np.sqrt(tnorm2)
        ynorm = np.linalg.norm(x)
        epsx = Anorm * ynorm * _EPS
        rnorm = phibar
        test1 = np.inf if (ynorm == 0 or Anorm == 0) else rnorm / (Anorm * ynorm)
        test2 = np.inf if Anorm == 0 else root / Anorm
        Acond = gmax / gmin

        if istop == 0 and not own_exits:
            if itn >= maxiter:
                istop = 6
        elif istop == 0:
            if 1 + test2 <= 1:
                istop = 2
            if 1 + test1 <= 1:
                istop = 1
            if itn >= maxiter:
                istop = 6
            if Acond >= 0.1 / _EPS:
                istop = 4
            if epsx >= beta1:
                istop = 3
            if test2 <= rtol:
                istop = 2
            if test1 <= rtol:
                istop = 1
        if callback is not None:
            callback(x)
        if istop != 0:
            break
    info = maxiter if istop == 6 else 0
    return x, info, itn, istop


def cg_hs(matvec, b, rtol=1e-10, maxiter=None, callback=None):
    """Conjugate gradients, x0 = 0, no preconditioner; stops when the
    recurrence residual satisfies ||r|| < rtol * ||b|| (SciPy 1.15.3
    _isolve/iterative.py cg with atol=0).  Returns (x, info, itn)."""
    b = np.asarray(b, dtype=np.float64)
    n = b.shape[0]
    if maxiter is None:
        maxiter = 10 * n
    bnrm2 = np.linalg.norm(b)
    x = np.zeros(n)
    if bnrm2 == 0:
        return x, 0, 0
    atol = rtol * bnrm2
    r = b.copy()
    rho_prev, p = None, None
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            return x, 0, it
        z = r
        rho_cur = float(r @ z)
        if it > 0:
            p = z + (rho_cur / rho_prev) * p
        else:
            p = z.copy()
        q = matvec(p)
        alpha = rho_cur / float(p @ q)
        x = x + alpha * p
        r = r - alpha * q
        rho_prev = rho_cur
        if callback is not None:
            callback(x)
    return x, maxiter, maxiter


class _EarlyExit(Exception):
    def __init__(self, x):
        super().__init__()
        self.x = x


def iterative_solve(matvec, y, tol=1e-4, minres=True, check_every=100,
                    use_scipy=False, own_exits=True):
    """The reference's Iterative.solve (approx/iterative.py:23-62).
    own_exits=False: MINRES's own stopping tests off (see minres_ps), so the
    reference's residual rule or n iterations end the solve.

    Returns (x, iterations, final_residual_norm, converged_flag)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    ctr = 0

    def cb(x):
        nonlocal ctr
        ctr += 1
        if ctr % check_every == 0:
            if np.linalg.norm(y - matvec(x)) < tol:
                raise _EarlyExit(x)

    inner_tol = min(1e-10, tol)
    try:
        if use_scipy:
            import scipy.sparse.linalg as sla
            op = sla.LinearOperator((n, n), matvec=matvec, dtype=np.float64)
            fn = sla.minres if minres else sla.cg
            x, info = fn(op, y, rtol=inner_tol, maxiter=n, callback=cb)
        elif minres:
            x, info, _, _ = minres_ps(matvec, y, rtol=inner_tol, maxiter=n,
                                      callback=cb, own_exits=own_exits)
        else:
            x, info, _ = cg_hs(matvec, y, rtol=inner_tol, maxiter=n,
                               callback=cb)
    except _EarlyExit as e:
        x, info = e.x, 0
    err = float(np.linalg.norm(y - matvec(x)))
    return x, ctr, err, (err <= tol and info == 0)
