"""examples/fit_real_data.py on the GPU: the reference's benchmark workloads
fitted end to end (AdaDelta + held-out prediction) land in the bands the
reference publishes -- FX2007: SMSE 0.21, NLPD -3.62 over ten runs (single runs
scatter 0.19-0.24, -3.3 to -3.7); weather: SMSE 0.09, NLPD 1.72 / 1.69 at
m = 500 / 1000 (its own ten runs scatter 0.046-0.144 and 1.20-3.66, one random
initialisation in ten ends in a poor optimum -- hence medians here); synthetic
two-input benchmark: SMSE 0.12, NLPD 0.28 (paper/results_synth.tex)."""
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


def _run(name, runs):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'fit_real_data.py'),
                          name, str(runs)], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout + out.stderr
    last = [l for l in out.stdout.splitlines() if l.startswith(name + ': n = ')][-1]
    med = last.split('medians:')[1]
    return (last, float(med.split('fit')[1].split()[0]), float(med.split('SMSE')[1].split()[0].rstrip(',')),
            float(med.split('NLPD')[1].split()[0]))


@pytest.mark.gpu
@pytest.mark.parametrize('name,grid', [('weather', '504'), ('weather1000', '1004')])
def test_fit_weather_median_band(name, grid):
    """BASELINE config 4 (D = 4 stations, n = 15 789, 2 SLFM + 4 independent
    kernels) at both published grid sizes; median of five runs."""
    last, fit, smse, nlpd = _run(name, 5)
    assert 'grid %s ' % grid in last, last
    assert 0.04 < smse < 0.16, last
    assert 1.0 < nlpd < 2.8, last
    assert fit < 30.0, last


@pytest.mark.gpu
def test_fit_synthetic_two_input_benchmark():
    """benchmarks/synth/synth.py:30-55: D = 5, n = 47 527 on the unit square,
    25 x 25 interpolating points (29 x 29 grid), tolerance 1e-3; the reference
    publishes 161 s, SMSE 0.12, NLPD 0.28."""
    last, fit, smse, nlpd = _run('synth', 3)
    assert 'grid 29x29 ' in last, last
    assert 0.09 < smse < 0.16, last
    assert 0.15 < nlpd < 0.45, last
    assert fit < 60.0, last
