"""ctypes binding of the C ABI declared in include/runlmc_hip.h.

The product library is ``runlmc_amd/csrc/librunlmc_hip.so`` (hipcc, gfx950).
There is no CPU fallback: if the library is missing or cannot be loaded,
every operator constructor raises.  Tests may point the binding at the
thread-level emulator build (tests/emu) with :func:`use_library`; the package
never does that on its own.
"""
import ctypes
import os
import re

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(HERE, 'csrc', 'librunlmc_hip.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'runlmc_hip.h')

RL_OK, RL_EINVAL, RL_EHIP, RL_ENOMEM, RL_ELIMIT = 0, 1, 2, 3, 4

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_dbl_p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double

# name -> argtypes; every function returns int unless listed in _RESTYPE
_SIGNATURES = {
    'rl_abi_version': [],
    'rl_device_count': [_c_int_p],
    'rl_gridop_create': [_i, _i, _i, _i, ctypes.POINTER(_vp)],
    'rl_gridop_create_2d': [_i, _i, _i, _i, _i, ctypes.POINTER(_vp)],
    'rl_gridop_destroy': [_vp],
    'rl_gridop_info': [_vp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p],
    'rl_gridop_form': [_vp, _c_int_p, ctypes.POINTER(ctypes.c_longlong)],
    'rl_gridop_set_form_gate': [_vp, ctypes.c_longlong],
    'rl_gridop_top_forms': [_vp, _c_int_p, _c_int_p],
    'rl_gridop_set_lmc': [_vp, _i, _vp, _vp, _vp, _vp],
    'rl_gridop_set_dense': [_vp, _i, _vp, _vp],
    'rl_gridop_form_stats': [_vp, _i, _vp],
    'rl_gridop_mvm': [_vp, _vp, _vp, _i, _vp],
    'rl_gridop_mvm_top': [_vp, _i, _vp, _vp, _i, _vp],
    'rl_gridop_spectrum_host': [_vp, _i, _vp],
    'rl_ski_create': [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp)],
    'rl_ski_destroy': [_vp],
    'rl_ski_add_term': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'rl_ski_apply_wt_term': [_vp, _i, _vp, _vp, _i, _vp],
    'rl_ski_apply_w_term': [_vp, _i, _vp, _vp, _i, _vp],
    'rl_ski_set_noise': [_vp, _vp, _vp],
    'rl_ski_mvm': [_vp, _vp, _vp, _i, _vp],
    'rl_ski_apply_wt': [_vp, _vp, _vp, _i, _vp],
    'rl_ski_apply_w': [_vp, _vp, _vp, _i, _vp],
    'rl_solve_batch': [_vp, _vp, _vp, _i, _i, _d, _i, _i, _vp, _vp, _vp, _vp],
    'rl_solve_batch_lanczos': [_vp, _vp, _vp, _i, _i, _d, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    'rl_slq_log_quadrature': [_vp, _i, _i, _vp, _vp, _vp, _i],
    'rl_probes_to_int8': [_vp, _i, ctypes.c_longlong, ctypes.c_longlong, _vp, _i, _c_int_p],
    'rl_ski_factor': [_vp, _c_int_p, _c_dbl_p, _c_dbl_p],
    'rl_ski_project': [_vp, _vp, _i, _vp, _c_int_p, _vp],
    'rl_gridop_poly_coeffs': [_vp, _i, _vp, _i, _c_int_p],
    'rl_gridop_project': [_vp, _vp, _i, _i, _vp, _vp],
    'rl_gridop_set_rank_hint': [_vp, _i],
    'rl_solve_direct': [_vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp, _vp],
    'rl_solve_pcg': [_vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp, _vp],
    'rl_solve_pcg_lanczos': [_vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    'rl_ski_precond_sample': [_vp, _vp, _vp, _i, _c_dbl_p, _vp],
    'rl_cross_dots': [_vp, _vp, _i, _i, _i, _vp, _vp],
    'rl_segment_dots': [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
}
ABI_VERSION = 4      # include/runlmc_hip.h: RL_ABI_VERSION
_RESTYPE = {'rl_last_error': ctypes.c_char_p, 'rl_backend': ctypes.c_char_p}


def declared_symbols(header=HEADER):
    """Names of every function the public header declares."""
    text = open(header).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rl_[a-z0-9_]+)\s*\(', text)))


class NativeError(RuntimeError):
    pass


class NativeLib:
    """One loaded copy of the native library."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise NativeError(
                'native library %s is missing: build it with '
                '`python -m runlmc_amd.build` (hipcc, gfx950). runlmc_amd has '
                'no CPU fallback.' % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, rt in _RESTYPE.items():
            getattr(self.cdll, name).restype = rt
            getattr(self.cdll, name).argtypes = []
        for name, args in _SIGNATURES.items():
            fn = getattr(self.cdll, name, None)
            if fn is None:
                continue
            fn.restype = ctypes.c_int
            fn.argtypes = args
        self.backend = self.cdll.rl_backend().decode()
        self.is_hip = self.backend.startswith('hip')
        # the signatures above are those of ABI version ABI_VERSION (include/runlmc_hip.h:
        # RL_ABI_VERSION); a library of another version would take shifted arguments
        ver = getattr(self.cdll, 'rl_abi_version', None)
        have = int(ver()) if ver is not None else 1
        if have != ABI_VERSION:
            raise NativeError(
                '%s implements ABI version %d, this binding is written against version %d: '
                'rebuild it (`python -m runlmc_amd.build --force`)' % (path, have, ABI_VERSION))

    # -- error mapping -----------------------------------------------------
    def check(self, rc):
        if rc == RL_OK:
            return
        msg = self.cdll.rl_last_error().decode()
        if rc == RL_EINVAL:
            raise ValueError(msg)
        if rc == RL_ELIMIT:
            raise NotImplementedError(msg)
        if rc == RL_ENOMEM:
            raise MemoryError(msg)
        raise NativeError(msg)

    def call(self, name, *args):
        self.check(getattr(self.cdll, name)(*args))

    # -- device memory handles ----------------------------------------------
    def torch_device(self, index=0):
        """Where vectors handed to this library must live."""
        if self.is_hip:
            if not torch.cuda.is_available():
                raise NativeError(
                    'librunlmc_hip.so is loaded but no GPU is visible; '
                    'runlmc_amd has no CPU fallback')
            return torch.device('cuda', index)
        return torch.device('cpu')

    def stream_ptr(self, device):
        if self.is_hip:
            return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        return ctypes.c_void_p(0)


_active = None


def get_library():
    """The active native library (loads the HIP build on first use)."""
    global _active
    if _active is None:
        _active = NativeLib(HIP_LIB)
    return _active


def use_library(path):
    """Point the binding at another build of the same ABI (tests only)."""
    global _active
    _active = NativeLib(path) if path is not None else None
    return _active


def host_ptr(arr):
    """void* of a C-contiguous numpy array (kept alive by the caller)."""
    if arr is None:
        return ctypes.c_void_p(0)
    assert arr.flags['C_CONTIGUOUS']
    return ctypes.c_void_p(arr.ctypes.data)


def dev_ptr(t):
    """void* of a contiguous torch tensor."""
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def as_f64(a, name='array'):
    """Safe cast to float64, as the reference does
    (runlmc/linalg/bttb.py:104: astype('float64', casting='safe'))."""
    a = np.asarray(a)
    return np.ascontiguousarray(a.astype(np.float64, casting='safe'))
