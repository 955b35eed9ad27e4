"""Direct sum of (possibly rectangular) operators (mirror of reference
runlmc/linalg/block_diag.py:11-49)."""
import numpy as np
import scipy.linalg as la

from .matrix import Matrix


class BlockDiag(Matrix):
    def __init__(self, blocks):
        blocks = list(blocks)
        rows = np.cumsum([0] + [b.shape[0] for b in blocks])
        cols = np.cumsum([0] + [b.shape[1] for b in blocks])
        super().__init__(int(rows[-1]), int(cols[-1]))
        self.blocks = blocks
        self._rows, self._cols = rows, cols

    def matvec(self, x):
        out = np.empty(self.shape[0], dtype=self.dtype)
        for i, blk in enumerate(self.blocks):
            out[self._rows[i]:self._rows[i + 1]] = blk.matvec(
                x[self._cols[i]:self._cols[i + 1]])
        return out

    def as_numpy(self):
        return la.block_diag(*[b.as_numpy() for b in self.blocks])

    def __str__(self):
        return 'BlockDiag(..., blocki, ...)\n' + '\n'.join(
            'block{}\n{!s}'.format(i, b) for i, b in enumerate(self.blocks))
