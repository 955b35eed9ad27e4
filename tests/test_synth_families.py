"""The reference benchmark's kernel families as restated in
runlmc_amd/util/synth.py (reference benchmarks/benchlib/bench.py:94,284-297,
gen_kernels): which kernels a family holds, that the data do not depend on the
family, and that the package's kernels and the oracle's evaluate to the same
top rows.  CPU only."""
import numpy as np

from runlmc_amd.util import synth
from oracle.kernels import RBFSpec, Matern32Spec, StdPeriodicSpec


def test_families_follow_gen_kernels():
    g5 = np.logspace(0, 1, 5)
    assert synth.kernel_family(5, 'rbf') == [('rbf', g) for g in g5]
    assert synth.kernel_family(5, 'periodic') == [('periodic', 1.0, g) for g in g5]
    assert synth.kernel_family(5, 'matern') == [('matern', g) for g in g5]
    # mix: one (rbf, periodic, matern) triple per gamma in logspace(0, 1, max(q // 3, 1)),
    # cut to q or padded with rbf(1)
    assert synth.kernel_family(5, 'mix') == [('rbf', 1.0), ('periodic', 1.0, 1.0), ('matern', 1.0),
                                             ('rbf', 1.0), ('rbf', 1.0)]
    assert synth.kernel_family(2, 'mix') == [('rbf', 1.0), ('periodic', 1.0, 1.0)]
    m6 = synth.kernel_family(6, 'mix')
    assert [d[0] for d in m6] == ['rbf', 'periodic', 'matern'] * 2
    assert m6[3][1] == 10.0 and m6[0][1] == 1.0


def test_data_do_not_depend_on_the_family():
    a = synth.make_problem(3, 2, 1, 60, kern='rbf')
    b = synth.make_problem(3, 2, 1, 60, kern='matern')
    assert np.array_equal(a.y, b.y) and np.array_equal(a.grid, b.grid)
    assert np.array_equal(a.coreg_vecs, b.coreg_vecs) and np.array_equal(a.noise, b.noise)
    assert (a.W != b.W).nnz == 0


def test_package_and_oracle_kernels_agree():
    for kern in synth.KERN_FAMILIES:
        p = synth.make_problem(2, 3, 1, 50, kern=kern)
        mine = synth.tops(p)
        theirs = np.array([k.from_dist(p.grid_dists) for k in synth.kernel_objects(
            p.kern_desc, rbf=RBFSpec, periodic=StdPeriodicSpec, matern=Matern32Spec)])
        assert np.array_equal(mine, theirs), kern
        fk = synth.functional_kernel(p)
        assert fk.Q == 3
        grads = fk.eval_kernel_gradients({(0,): p.grid_dists})
        assert [len(g) for g in grads] == [2 if d[0] == 'periodic' else 1 for d in p.kern_desc]
