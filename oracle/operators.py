"""Oracle: structured-operator matrix-vector products (NumPy, float64).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

All vectors are output-major: grid index = d*m + i (reference
runlmc/linalg/kronecker.py:42-46 reshapes to (-1, m)).
"""
import numpy as np


def next_pow2(x):
    """Smallest power of two >= x (reference runlmc/linalg/bttb.py:16-19)."""
    x = int(x)
    return 1 << (x - 1).bit_length() if x > 1 else 1


def circulant_embed(top, sizes):
    """Zero-padded, mirrored first column of the circulant that embeds a
    symmetric BTTB matrix (reference bttb.py:110-120).

    Per axis of length n the embedding length is next_pow2(2n); entry
    [L-k] = entry [k] for k = 1..n-1, everything between stays zero.
    """
    sizes = tuple(int(s) for s in sizes)
    ext_shape = tuple(next_pow2(2 * s) for s in sizes)
    ext = np.zeros(ext_shape)
    ext[tuple(slice(0, s) for s in sizes)] = np.asarray(
        top, dtype=np.float64).reshape(sizes)
    # mirror one axis at a time, last axis first, each time over the whole
    # already-mirrored trailing block (bttb.py:114-119)
    for ax in range(len(sizes) - 1, -1, -1):
        n, big = sizes[ax], ext_shape[ax]
        if n > 1:
            dst = [slice(None)] * len(sizes)
            src = [slice(None)] * len(sizes)
            dst[ax] = slice(big - n + 1, big)
            src[ax] = slice(n - 1, 0, -1)
            ext[tuple(dst)] = ext[tuple(src)]
    return ext


def bttb_spectrum(top, sizes):
    """rfftn of the circulant embedding (reference bttb.py:106-108).  The
    result is complex with an O(1e-16 * scale) imaginary part."""
    return np.fft.rfftn(circulant_embed(top, sizes))


def bttb_matvec(spectrum, sizes, x):
    """pad -> rfftn -> pointwise -> irfftn -> crop (reference bttb.py:144-148)."""
    sizes = tuple(int(s) for s in sizes)
    ext_shape = [next_pow2(2 * s) for s in sizes]
    xf = np.fft.rfftn(np.asarray(x, dtype=np.float64).reshape(sizes),
                      axes=tuple(range(len(sizes))),
                      s=ext_shape)
    xf *= spectrum
    full = np.fft.irfftn(xf, s=ext_shape, axes=tuple(range(len(sizes))))
    return full[tuple(slice(0, s) for s in sizes)].ravel()


def toeplitz_matvec(top, x):
    """Length-2n circulant embedding, not rounded up to a power of two
    (reference runlmc/linalg/toeplitz.py:45-67)."""
    top = np.asarray(top, dtype=np.float64)
    n = len(top)
    col = np.zeros(2 * n)
    col[:n] = top
    col[n + 1:] = top[1:][::-1]
    spec = np.fft.rfft(col)
    xf = np.fft.rfft(np.asarray(x, dtype=np.float64), n=2 * n)
    return np.fft.irfft(xf * spec, n=2 * n)[:n]


def bttb_dense(top, sizes):
    """Dense symmetric BTTB from its first row (what reference
    bttb.py:122-142 as_numpy() builds): entry (i, j) is top at the
    per-axis absolute index difference."""
    sizes = tuple(int(s) for s in sizes)
    top = np.asarray(top, dtype=np.float64).reshape(sizes)
    idx = np.indices(sizes).reshape(len(sizes), -1)  # (P, N)
    diff = np.abs(idx[:, :, None] - idx[:, None, :])  # (P, N, N)
    return top[tuple(diff)]


class BTTBOracle:
    """Holder mirroring the reference BTTB object's matvec/matmat/as_numpy
    (reference bttb.py:91-148, matrix.py:55-67)."""

    def __init__(self, top, sizes=None):
        top = np.asarray(top)
        if top.ndim != 1:
            raise ValueError('top must be 1-D')
        if top.size == 0:
            raise ValueError('top is empty')
        sizes = (len(top),) if sizes is None else tuple(int(s) for s in sizes)
        if int(np.prod(sizes)) != top.size:
            raise ValueError('sizes do not match top')
        self.top = top.astype(np.float64, casting='safe')
        self.sizes = sizes
        self.shape = (top.size, top.size)
        self.spectrum = bttb_spectrum(self.top, sizes)

    def matvec(self, x):
        return bttb_matvec(self.spectrum, self.sizes, x)

    def matmat(self, X):
        return np.stack([self.matvec(c) for c in np.asarray(X).T], axis=1)

    def as_numpy(self):
        return bttb_dense(self.top, self.sizes)


def kron_matvec(B, toep, x):
    """(B kron T) x for dense B (a x b) and a BTTBOracle T (reference
    kronecker.py:39-46 with numpy_matrix.py:30-31): apply T to each of the
    b row-blocks of x, then mix the blocks with B."""
    B = np.asarray(B, dtype=np.float64)
    m = toep.shape[0]
    X = np.asarray(x, dtype=np.float64).reshape(B.shape[1], m)
    TX = np.stack([toep.matvec(row) for row in X], axis=0)
    return (B @ TX).reshape(-1)


# --- the three grid-kernel representations ---------------------------------

def grid_sum_matvec(Bs, toeps, x):
    """'sum' representation: sum_q (B_q kron T_q) x (reference
    runlmc/lmc/grid_kernel.py:126-136, sum_matrix.py:31-32)."""
    out = np.zeros(Bs[0].shape[0] * toeps[0].shape[0])
    for B, T in zip(Bs, toeps):
        out = out + kron_matvec(B, T, x)
    return out


def grid_bt_matvec(Bs, tops, sizes, x):
    """'bt' representation: D x D blocks T_ab = BTTB(sum_q B_q[a,b] k_q)
    (reference grid_kernel.py:115-123, block_matrix.py:32-37)."""
    Bs = np.asarray(Bs, dtype=np.float64)
    tops = np.asarray(tops, dtype=np.float64)
    D = Bs.shape[1]
    m = tops.shape[1]
    mixed = np.tensordot(Bs, tops, axes=(0, 0))  # (D, D, m)
    X = np.asarray(x, dtype=np.float64).reshape(D, m)
    out = np.zeros((D, m))
    cache = {}
    for a in range(D):
        for b in range(D):
            key = (min(a, b), max(a, b))
            if key not in cache:
                cache[key] = BTTBOracle(mixed[key[0], key[1]], sizes)
            out[a] += cache[key].matvec(X[b])
    return out.reshape(-1)


def grid_slfm_matvec(coreg_vecs, coreg_diags, tops, sizes, x):
    """'slfm' representation for a model whose Q kernels all carry a
    coregionalisation vector block A_q (R_q x D) and a diagonal kappa_q:
    (A*^T kron I) blockdiag(T_{q(r)}) (A* kron I) x + blockdiag_d(T(sum_q
    kappa_qd k_q)) x  (reference grid_kernel.py:77-112, block_diag.py:36-40).
    A* stacks all A_q rows (grid_kernel.py:90-92); note left = A*^T... in the
    reference `A_star = vstack(all_coreg).T` is D x R, left = A_star kron I,
    right = A_star^T kron I."""
    tops = np.asarray(tops, dtype=np.float64)
    Q, m = tops.shape
    D = np.asarray(coreg_diags[0]).shape[0]
    X = np.asarray(x, dtype=np.float64).reshape(D, m)
    toeps = [BTTBOracle(t, sizes) for t in tops]
    out = np.zeros((D, m))
    for q in range(Q):
        A = np.atleast_2d(np.asarray(coreg_vecs[q], dtype=np.float64))
        for r in range(A.shape[0]):
            latent = A[r] @ X               # (m,)
            out += np.outer(A[r], toeps[q].matvec(latent))
    diag_tops = np.column_stack(coreg_diags) @ tops   # (D, m)
    for d in range(D):
        out[d] += BTTBOracle(diag_tops[d], sizes).matvec(X[d])
    return out.reshape(-1)


def coreg_mats(coreg_vecs, coreg_diags):
    """B_q = A_q^T A_q + diag(kappa_q) (reference
    runlmc/lmc/functional_kernel.py:280-287)."""
    return [np.atleast_2d(a).T @ np.atleast_2d(a) + np.diag(k)
            for a, k in zip(coreg_vecs, coreg_diags)]


def ski_matvec(W, WT, grid_mv, x):
    """W (K_UU (W^T x)) (reference runlmc/approx/ski.py:8-16,
    composition.py:14-17).  W, WT are scipy CSR matrices."""
    return W.dot(grid_mv(WT.dot(x)))


def full_matvec(W, WT, grid_mv, noise_diag, x):
    """K~ x = W K_UU W^T x + diag(eps) x (reference grid_kernel.py:70-74,
    diag.py:24-25, sum_matrix.py:31-32)."""
    return ski_matvec(W, WT, grid_mv, x) + noise_diag * x


def dense_from_matvec(mv, n):
    """Column-by-column densification (reference matrix.py:45-49)."""
    eye = np.identity(n)
    return np.stack([mv(eye[:, j]) for j in range(n)], axis=1)
