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


# ---------------------------------------------------------------------------
# On-chip scheme (k4_product): ONE real vector per workgroup, the length-L real
# transform done as an N = L/2 point complex transform of z[n] = x[2n] + i x[2n+1]
# (itself split Na x Nb inside LDS), untangled pairwise (k, N-k) into the
# half spectrum k = 0..N, mixed, re-tangled and transformed back.
# ---------------------------------------------------------------------------
def onchip_positions(Na, Nb, planA, planB):
    """pos[k] = (row, col) of frequency k of the N-point transform after
    four_step_forward(., Na, Nb): k = kA(row) + Na * kB(col)."""
    fa = position_to_freq(Na, planA)
    fb = position_to_freq(Nb, planB)
    inva = np.argsort(fa)
    invb = np.argsort(fb)
    k = np.arange(Na * Nb)
    return inva[k % Na], invb[k // Na]


def onchip_untangle(S, Na, Nb, planA, planB, L):
    """S: scrambled N-point spectrum (Na, Nb) of the packed sequence.  Returns
    (c, X2lo, X2hi): for every pair slot c in [0, N/2], twice the real
    sequence's spectrum at k = c and at k = N - c."""
    N = Na * Nb
    r, cpos = onchip_positions(Na, Nb, planA, planB)
    c = np.arange(N // 2 + 1)
    A = S[r[c], cpos[c]]
    B = S[r[(N - c) % N], cpos[(N - c) % N]]
    w = np.exp(-2j * np.pi * c / L)
    P = A + np.conj(B)
    Qd = 1j * w * (A - np.conj(B))
    return c, P - Qd, np.conj(P + Qd)


def onchip_retangle(Ulo, Uhi, Na, Nb, planA, planB, L):
    """Inverse of the untangle for a Hermitian product spectrum: values at
    k = c (Ulo) and k = N - c (Uhi) -> scrambled N-point array (Na, Nb)."""
    N = Na * Nb
    r, cpos = onchip_positions(Na, Nb, planA, planB)
    c = np.arange(N // 2 + 1)
    w = np.exp(-2j * np.pi * c / L)
    P = Ulo + np.conj(Uhi)
    Qd = 1j * np.conj(w) * (Ulo - np.conj(Uhi))
    S = np.zeros((Na, Nb), dtype=np.complex128)
    S[r[c], cpos[c]] = P + Qd
    S[r[(N - c) % N], cpos[(N - c) % N]] = np.conj(P - Qd)
    return S


def onchip_grid_mvm_model(tops, facA, facW, facQ, kappa, X, L, Na, Nb, planA, planB):
    """K_UU X for X (nvec, D, m), one real vector at a time (see above)."""
    nvec, D, m = X.shape
    Q = len(tops)
    N = L // 2
    assert Na * Nb == N

    def half_spectrum2(x_real_L):
        z = x_real_L[0::2] + 1j * x_real_L[1::2]
        S = four_step_forward(z, Na, Nb, planA, planB)
        return onchip_untangle(S, Na, Nb, planA, planB, L)

    # spectra at k = c and k = N - c, scaled so that the unnormalised adjoint
    # returns the product: t_hat / (2 L)
    slo = np.zeros((Q, N // 2 + 1))
    shi = np.zeros((Q, N // 2 + 1))
    for q in range(Q):
        _, lo, hi = half_spectrum2(circulant_column(tops[q], L))
        assert np.abs(lo.imag).max() < 1e-9 * (1 + np.abs(lo).max())
        slo[q], shi[q] = lo.real / (4 * L), hi.real / (4 * L)
    Y = np.zeros_like(X)
    for v in range(nvec):
        Xlo = np.zeros((D, N // 2 + 1), dtype=np.complex128)
        Xhi = np.zeros_like(Xlo)
        for b in range(D):
            xp = np.zeros(L)
            xp[:m] = X[v, b]
            _, Xlo[b], Xhi[b] = half_spectrum2(xp)
        out = []
        for Z, s in ((Xlo, slo), (Xhi, shi)):
            Yh = np.einsum('qd,qk->dk', kappa, s) * Z
            for f in range(len(facW)):
                t = np.einsum('b,bk->k', facA[f], Z) * (facW[f] * s[facQ[f]])
                Yh += facA[f][:, None] * t[None]
            out.append(Yh)
        for a in range(D):
            S = onchip_retangle(out[0][a], out[1][a], Na, Nb, planA, planB, L)
            z = four_step_adjoint(S, Na, Nb, planA, planB)
            y = np.zeros(L)
            y[0::2], y[1::2] = z.real, z.imag
            Y[v, a] = y[:m]
    return Y


# ---------------------------------------------------------------------------
# Two-phase on-chip scheme (what k4_product actually runs): the length-L real
# transform of a sequence whose second half is zero padding, split by parity of
# the frequency.  With N = L/2, H = N/2:
#   even k = 2k':  the length-N real transform of x[0:N]  -> H-point complex
#                  transform of x[2n] + i x[2n+1] + pairwise untangle
#   odd  k = 4j+1: the H-point complex transform of (x[n] - i x[n+H]) W_L^n
#                  (k = 4j+3 are the conjugates of mirrored ones: never formed)
# Each phase keeps D * H complex values on chip; the result is the sum.
# ---------------------------------------------------------------------------
def twophase_grid_mvm_model(tops, facA, facW, facQ, kappa, X, L, Ha, Hb, planA, planB):
    nvec, D, m = X.shape
    Q = len(tops)
    N = L // 2
    H = N // 2
    assert Ha * Hb == H and m <= N
    n = np.arange(H)
    wL = np.exp(-2j * np.pi * n / L)

    def mix(Z, s):                       # Z (D, K) complex, s (Q, K) real
        Yh = np.einsum('qd,qk->dk', kappa, s) * Z
        for f in range(len(facW)):
            t = np.einsum('b,bk->k', facA[f], Z) * (facW[f] * s[facQ[f]])
            Yh += facA[f][:, None] * t[None]
        return Yh

    def even_forward(u):                 # u real (N,) -> 2 E[c], 2 E[H - c]
        z = u[0::2] + 1j * u[1::2]
        S = four_step_forward(z, Ha, Hb, planA, planB)
        return onchip_untangle(S, Ha, Hb, planA, planB, N)

    def odd_forward(u):                  # u real (N,) -> scrambled (Ha, Hb): O[2j]
        z = (u[:H] - 1j * u[H:]) * wL
        return four_step_forward(z, Ha, Hb, planA, planB)

    se_lo = np.zeros((Q, H // 2 + 1))
    se_hi = np.zeros_like(se_lo)
    so = np.zeros((Q, Ha, Hb))
    for q in range(Q):
        c = circulant_column(tops[q], L)
        _, lo, hi = even_forward(c[:N] + c[N:])
        se_lo[q], se_hi[q] = lo.real / (4 * L), hi.real / (4 * L)
        S = odd_forward(c[:N] - c[N:])
        assert np.abs(S.imag).max() < 1e-9 * (1 + np.abs(S).max())
        so[q] = S.real * (2.0 / L)
    Y = np.zeros_like(X)
    for v in range(nvec):
        xp = np.zeros((D, N))
        xp[:, :m] = X[v]
        # phase E
        lo = np.zeros((D, H // 2 + 1), dtype=np.complex128)
        hi = np.zeros_like(lo)
        for b in range(D):
            _, lo[b], hi[b] = even_forward(xp[b])
        ylo, yhi = mix(lo, se_lo), mix(hi, se_hi)
        for a in range(D):
            S = onchip_retangle(ylo[a], yhi[a], Ha, Hb, planA, planB, N)
            z = four_step_adjoint(S, Ha, Hb, planA, planB)
            y = np.zeros(N)
            y[0::2], y[1::2] = z.real, z.imag
            Y[v, a] = y[:m]
        # phase O
        So = np.zeros((D, Ha, Hb), dtype=np.complex128)
        for b in range(D):
            So[b] = odd_forward(xp[b])
        Yo = mix(So.reshape(D, -1), so.reshape(Q, -1)).reshape(D, Ha, Hb)
        for a in range(D):
            h = np.conj(wL) * four_step_adjoint(Yo[a], Ha, Hb, planA, planB)
            y = np.concatenate([h.real, -h.imag])
            Y[v, a] += y[:m]
    return Y
