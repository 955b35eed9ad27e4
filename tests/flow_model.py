"""Executable specification of the device FFT flow graph (NumPy, slow).

The HIP kernels in runlmc_amd/csrc implement exactly this index arithmetic:
in-place decimation-in-frequency passes (natural order in, digit-scrambled
order out), a four-step N1 x N2 split with an inter-step twiddle, a pointwise
real mix at every scrambled position, and the adjoint passes back.  Nothing is
ever un-scrambled: the circulant spectra are produced by the same forward
graph and so live in the same scrambled positions.

Used by tests/test_flow_model.py to check the scheme against numpy.fft and
the oracle, and as a readable reference when debugging kernels.
"""
import numpy as np


def radix_plan(N, max_radix=8):
    """Radix list for a power-of-two N (largest radices first)."""
    plan, rem = [], N
    while rem > 1:
        r = max_radix
        while rem % r:
            r //= 2
        plan.append(r)
        rem //= r
    return plan


def dif_forward(a, plan):
    """In place over axis 0 of `a` (shape (N, ...)); returns nothing."""
    N = a.shape[0]
    Ns = N
    for R in plan:
        sub = Ns // R
        wR = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
        for g in range(0, N, Ns):
            for j in range(sub):
                idx = g + j + sub * np.arange(R)
                out = np.tensordot(wR, a[idx], axes=(1, 0))
                tw = np.exp(-2j * np.pi * j * np.arange(R) / Ns)
                a[idx] = out * tw.reshape((R,) + (1,) * (a.ndim - 1))
        Ns = sub


def dif_adjoint(a, plan):
    """Conjugate transpose of dif_forward (unnormalised inverse)."""
    N = a.shape[0]
    subs = []
    Ns = N
    for R in plan:
        subs.append((R, Ns))
        Ns //= R
    for R, Ns in reversed(subs):
        sub = Ns // R
        wR = np.exp(+2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
        for g in range(0, N, Ns):
            for j in range(sub):
                idx = g + j + sub * np.arange(R)
                tw = np.exp(+2j * np.pi * j * np.arange(R) / Ns)
                vin = a[idx] * tw.reshape((R,) + (1,) * (a.ndim - 1))
                a[idx] = np.tensordot(wR, vin, axes=(1, 0))


def position_to_freq(N, plan):
    """freq[p] = frequency index held at position p after dif_forward."""
    freq = np.zeros(N, dtype=np.int64)
    for p in range(N):
        rem, Ns, mult, f = p, N, 1, 0
        for R in plan:
            sub = Ns // R
            d = rem // sub
            rem -= d * sub
            f += d * mult
            mult *= R
            Ns = sub
        freq[p] = f
    return freq


def four_step_forward(z, N1, N2, plan1, plan2):
    """z: (L,) complex, L = N1*N2, n = N2*n1 + n2.  Returns S (N1, N2):
    S[r, c] = DFT(z)[k1(r) + N1*k2(c)]."""
    L = N1 * N2
    T = z.reshape(N1, N2).astype(np.complex128).copy()
    dif_forward(T, plan1)                      # kernel A: columns
    k1 = position_to_freq(N1, plan1)
    T *= np.exp(-2j * np.pi * np.outer(k1, np.arange(N2)) / L)
    S = T.T.copy()                             # kernel B works on (N2, rows)
    dif_forward(S, plan2)
    return S.T.copy()


def four_step_adjoint(S, N1, N2, plan1, plan2):
    L = N1 * N2
    U = S.T.copy()
    dif_adjoint(U, plan2)
    T = U.T.copy()
    k1 = position_to_freq(N1, plan1)
    T *= np.exp(+2j * np.pi * np.outer(k1, np.arange(N2)) / L)
    dif_adjoint(T, plan1)
    return T.reshape(L)


def circulant_column(top, L):
    m = len(top)
    c = np.zeros(L)
    c[:m] = top
    if m > 1:
        c[L - m + 1:] = top[1:][::-1]
    return c


def grid_mvm_model(tops, facA, facW, facQ, kappa, X, L, N1, N2,
                   max_radix=8):
    """Pair-packed K_UU X for X of shape (nvec, D, m), factored mix
    B_q = sum_{f: facQ[f]==q} facW[f] facA[f]^T facA[f] + diag(kappa[q])."""
    nvec, D, m = X.shape
    Q = len(tops)
    plan1, plan2 = radix_plan(N1, max_radix), radix_plan(N2, max_radix)
    spec = np.zeros((Q, N1, N2))
    for q in range(0, Q, 2):
        c = circulant_column(tops[q], L).astype(np.complex128)
        if q + 1 < Q:
            c = c + 1j * circulant_column(tops[q + 1], L)
        S = four_step_forward(c, N1, N2, plan1, plan2) / L
        spec[q] = S.real
        if q + 1 < Q:
            spec[q + 1] = S.imag
    Y = np.zeros_like(X)
    for p in range(0, nvec, 2):
        Z = np.zeros((D, N1, N2), dtype=np.complex128)
        for b in range(D):
            z = np.zeros(L, dtype=np.complex128)
            z[:m] = X[p, b]
            if p + 1 < nvec:
                z[:m] += 1j * X[p + 1, b]
            Z[b] = four_step_forward(z, N1, N2, plan1, plan2)
        dcoef = np.einsum('qd,qrc->drc', kappa, spec)
        Yh = dcoef * Z
        for f in range(len(facW)):
            s = np.einsum('b,brc->rc', facA[f], Z) * (facW[f] * spec[facQ[f]])
            Yh += facA[f][:, None, None] * s[None]
        for a in range(D):
            y = four_step_adjoint(Yh[a], N1, N2, plan1, plan2)
            Y[p, a] = y[:m].real
            if p + 1 < nvec:
                Y[p + 1, a] = y[:m].imag
    return Y
