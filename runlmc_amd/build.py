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
# three translation units (rl_host.h holds what they share); runlmc_hip.hip includes all three
# (experiment builds that need one code object)
SOURCES = [os.path.join(CSRC, n) for n in ('rl_gridop.hip', 'rl_ski.hip', 'rl_solve.hip')]
UNITY = os.path.join(CSRC, 'runlmc_hip.hip')
import glob
HEADERS = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [
    os.path.join(ROOT, 'include', 'runlmc_hip.h')]
HIP_LIB = os.path.join(CSRC, 'librunlmc_hip.so')
EMU_DIR = os.path.join(ROOT, 'tests', 'emu')
EMU_LIB = os.path.join(EMU_DIR, 'librunlmc_emu.so')
OBJ_DIR = os.path.join(CSRC, 'build')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)


def _compile_all(cmds):
    """The translation units side by side (hipcc / g++ are single-threaded per file)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(cmds)) as pool:
        list(pool.map(_run, cmds))


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0)')


def build_hip(force=False, extra=(), unity=False, out=None):
    """Compile the gfx950 shared library (cross-compiles without a GPU)."""
    lib = out or HIP_LIB
    if not force and not _stale(lib, SOURCES + HEADERS + [__file__]):
        return lib
    hipcc = find_hipcc()
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
    if unity:
        _run([hipcc] + flags + ['-shared', '-o', lib] + list(extra) + [UNITY])
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + '.hip.o') for s in SOURCES]
    # (per object: a file whose source and headers are older than its object is not recompiled --
    # an edit of the solver costs one translation unit, not three)
    todo = [(s, o) for s, o in zip(SOURCES, objs)
            if force or extra or _stale(o, [s] + HEADERS + [__file__])]
    if todo:
        _compile_all([[hipcc] + flags + list(extra) + ['-c', s, '-o', o] for s, o in todo])
    _run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', lib] + objs)
    return lib


def build_emu(force=False, sanitize=False):
    """Compile the same sources with g++ on top of tests/emu (tests only)."""
    emu_src = os.path.join(EMU_DIR, 'rl_emu.cpp')
    emu_hdr = os.path.join(EMU_DIR, 'rl_emu.h')
    lib = EMU_LIB.replace('.so', '_asan.so') if sanitize else EMU_LIB
    if not force and not _stale(lib, SOURCES + HEADERS + [emu_src, emu_hdr, __file__]):
        return lib
    flags = ['-O1' if sanitize else '-O2', '-g', '-std=c++17', '-fPIC', '-DRL_EMU',
             '-ffp-contract=off', '-I', EMU_DIR, '-pthread', '-Wall', '-Wno-unused-function',
             '-Wno-unknown-pragmas']
    if sanitize:
        flags += ['-fsanitize=address,undefined', '-fno-omit-frame-pointer']
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = '.asan.o' if sanitize else '.emu.o'
    srcs = SOURCES + [emu_src]
    objs = [os.path.join(OBJ_DIR, os.path.basename(s).rsplit('.', 1)[0] + tag) for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs)
            if force or _stale(o, [s] + HEADERS + [emu_hdr, __file__])]
    if todo:
        _compile_all([['g++'] + flags + ['-c', '-x', 'c++', s, '-o', o] for s, o in todo])
    _run(['g++', '-shared', '-pthread', '-o', lib] + (['-fsanitize=address,undefined'] if sanitize else [])
         + objs)
    return lib


if __name__ == '__main__':
    if '--timing' in sys.argv:
        # experiment build with in-kernel phase stamps (rl_device.h: RL_TIMING): one code object
        print(build_hip(force=True, extra=['-DRL_TIMING'], unity=True))
    elif '--emu' in sys.argv:
        print(build_emu(force='--force' in sys.argv, sanitize='--asan' in sys.argv))
    else:
        print(build_hip(force='--force' in sys.argv))
