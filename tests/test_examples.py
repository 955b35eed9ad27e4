"""examples/fit_real_data.py on the GPU: the reference's FX2007 workload fitted
end to end (AdaDelta + held-out prediction) lands in the band the reference
publishes (SMSE 0.21, NLPD -3.62 over ten runs; single runs scatter around
that: 0.19-0.24, -3.3 to -3.7)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fit_fx2007_end_to_end():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'fit_real_data.py'),
                          'fx2007', '3'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    last = [l for l in out.stdout.splitlines() if l.startswith('fx2007: n = 3054')][-1]
    smse = float(last.split('SMSE')[1].split()[0])
    nlpd = float(last.split('NLPD')[1].split()[0])
    fit = float(last.split('fit')[1].split()[0])
    assert 0.15 < smse < 0.30, last
    assert -4.0 < nlpd < -3.0, last
    assert fit < 30.0, last
