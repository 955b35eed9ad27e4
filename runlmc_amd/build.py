"""Build the native library in-tree.

    python -m runlmc_amd.build          # hipcc, gfx950  -> runlmc_amd/csrc/librunlmc_hip.so
    python -m runlmc_amd.build --emu    # g++ + tests/emu (debug aid, tests only)

The product library is the hipcc one.  The emulator build exists so kernel
logic can be debugged and sanitised on a machine without a GPU; the package
never loads it on its own (see runlmc_amd/_lib.py).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
SOURCES = [os.path.join(CSRC, 'runlmc_hip.hip')]
import glob
HEADERS = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [
    os.path.join(ROOT, 'include', 'runlmc_hip.h')]
HIP_LIB = os.path.join(CSRC, 'librunlmc_hip.so')
EMU_DIR = os.path.join(ROOT, 'tests', 'emu')
EMU_LIB = os.path.join(EMU_DIR, 'librunlmc_emu.so')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0)')


def build_hip(force=False, extra=()):
    """Compile the gfx950 shared library (cross-compiles without a GPU)."""
    if not force and not _stale(HIP_LIB, SOURCES + HEADERS + [__file__]):
        return HIP_LIB
    cmd = [find_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17',
           '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
           '-o', HIP_LIB] + list(extra) + SOURCES
    _run(cmd)
    return HIP_LIB


def build_emu(force=False, sanitize=False):
    """Compile the same sources with g++ on top of tests/emu (tests only)."""
    emu_src = os.path.join(EMU_DIR, 'rl_emu.cpp')
    emu_hdr = os.path.join(EMU_DIR, 'rl_emu.h')
    lib = EMU_LIB.replace('.so', '_asan.so') if sanitize else EMU_LIB
    if not force and not _stale(lib, SOURCES + HEADERS + [emu_src, emu_hdr, __file__]):
        return lib
    cmd = ['g++', '-O1' if sanitize else '-O2', '-g', '-std=c++17', '-fPIC',
           '-shared', '-DRL_EMU', '-ffp-contract=off', '-I', EMU_DIR, '-pthread',
           '-Wall', '-Wno-unused-function', '-Wno-unknown-pragmas', '-o', lib]
    if sanitize:
        cmd += ['-fsanitize=address,undefined', '-fno-omit-frame-pointer']
    for s in SOURCES:
        cmd += ['-x', 'c++', s]
    cmd += ['-x', 'c++', emu_src]
    _run(cmd)
    return lib


if __name__ == '__main__':
    if '--timing' in sys.argv:
        # experiment build with in-kernel phase stamps (rl_device.h: RL_TIMING)
        print(build_hip(force=True, extra=['-DRL_TIMING']))
    elif '--emu' in sys.argv:
        print(build_emu(force='--force' in sys.argv, sanitize='--asan' in sys.argv))
    else:
        print(build_hip(force='--force' in sys.argv))
