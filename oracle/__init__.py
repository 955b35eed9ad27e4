"""CPU oracle for the matrix-free LMC hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the reference algorithm
(vlad17/runlmc: runlmc/linalg, runlmc/approx, runlmc/lmc).  It exists so that
parity tests have something to check the HIP path against on a machine where
the reference sources are absent.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
    of ``bench.py`` may import anything from here;
  * nothing under ``runlmc_amd/`` imports it, and it imports nothing from
    ``runlmc_amd/``;
  * it is never the thing that is measured as the product or shipped.

Parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference
from the read-only reference checkout (build container only) and stores its outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors.

Every function cites the reference file:line it restates.
"""
