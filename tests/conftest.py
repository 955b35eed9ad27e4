import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The tests force kernel paths through the library's A/B hooks (RUNLMC_NO_RP, RUNLMC_STAGED_WT,
# ...), which the library reads only under this switch (csrc/rl_gridop.hip: read_knobs).  Child
# processes (multi-rank tests, the C ABI demo) inherit it.
os.environ.setdefault('RUNLMC_DEBUG', '1')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
