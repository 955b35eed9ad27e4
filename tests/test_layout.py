"""Repository contract checks (CPU): the C ABI library exports every symbol
the public header declares, the oracle is test-only, the product has no CPU
fallback."""
import ast
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(d):
        if '__pycache__' in base:
            continue
        for f in files:
            if f.endswith('.py'):
                yield os.path.join(base, f)


def _imports(path):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom) and node.module:
            yield ('.' * node.level) + node.module


def test_product_never_imports_oracle():
    for path in _py_files(os.path.join(ROOT, 'runlmc_amd')):
        for mod in _imports(path):
            assert not mod.lstrip('.').startswith('oracle'), (path, mod)


def test_oracle_never_imports_product():
    for path in _py_files(os.path.join(ROOT, 'oracle')):
        for mod in _imports(path):
            assert 'runlmc_amd' not in mod, (path, mod)


def test_oracle_headers_say_test_only():
    for path in _py_files(os.path.join(ROOT, 'oracle')):
        assert 'TEST INFRASTRUCTURE ONLY' in open(path).read(), path


def test_nothing_reads_reference_at_runtime():
    """/root/reference does not exist on the GPU box: only the golden
    generator (build container) may name it."""
    allowed = {os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'),
               os.path.abspath(__file__)}
    roots = [os.path.join(ROOT, d) for d in ('runlmc_amd', 'oracle', 'tests')]
    files = [f for r in roots for f in _py_files(r)]
    files += [os.path.join(ROOT, f) for f in ('bench.py', '__graft_entry__.py')]
    for path in files:
        if path in allowed:
            continue
        assert '/root/reference' not in open(path).read(), path


def test_library_exports_every_declared_symbol():
    from runlmc_amd import _lib, build
    declared = _lib.declared_symbols()
    assert len(declared) >= 18
    lib = build.build_hip()          # hipcc cross-compiles without a GPU
    cdll = ctypes.CDLL(lib)
    for name in declared:
        assert hasattr(cdll, name), 'librunlmc_hip.so lacks %s' % name
    cdll.rl_backend.restype = ctypes.c_char_p
    assert cdll.rl_backend() == b'hip-gfx950'
    # the binding knows every declared function too
    known = set(_lib._SIGNATURES) | set(_lib._RESTYPE)
    assert set(declared) <= known, set(declared) - known


def test_missing_library_fails_loudly(tmp_path):
    from runlmc_amd import _lib
    with pytest.raises(_lib.NativeError):
        _lib.NativeLib(str(tmp_path / 'nope.so'))


def test_no_gpu_means_no_operator():
    """With the HIP library but no GPU, constructing an operator raises (no
    silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import numpy as np
    from runlmc_amd import _lib
    from runlmc_amd.linalg.toeplitz import Toeplitz
    _lib.use_library(None)
    t = Toeplitz(np.ones(4))
    with pytest.raises(_lib.NativeError):
        t.matvec(np.ones(4))


def test_argument_errors_map_to_valueerror():
    from runlmc_amd import _lib, build
    from runlmc_amd._native import GridOp
    lib = _lib.use_library(build.build_emu())
    try:
        with pytest.raises(ValueError):
            GridOp(0, 5, 1)
        assert GridOp(17, 5, 1).D == 17   # (no limit on D since round 4: the wide operator)
        with pytest.raises(NotImplementedError):
            GridOp(5000, 5, 1)         # documented limit: D <= 4096
        g = GridOp(2, 5, 1)
        import numpy as np
        with pytest.raises(ValueError):
            g.set_lmc(np.ones((2, 5)), [None, None], [np.ones(2)] * 2)   # Q > max_tops
        with pytest.raises(ValueError):
            g.set_lmc(np.ones((1, 4)), [None], [np.ones(2)])             # wrong m
        with pytest.raises(ValueError):
            g.set_dense(np.ones((1, 5)), np.array([[[1.0, 2.0], [0.0, 1.0]]]))  # asymmetric
    finally:
        _lib.use_library(None)
