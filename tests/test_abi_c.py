"""The C ABI from plain C: examples/abi_demo.c is compiled with gcc against
librunlmc_hip.so (no Python, no torch in the process) and run on the GPU; it
checks the grid operator, the SKI operator and a MINRES solve against a dense
product of its own."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_abi_demo_from_c(tmp_path):
    from runlmc_amd.build import CSRC, HIP_LIB
    assert os.path.exists(HIP_LIB), 'librunlmc_hip.so is not built'
    gcc = shutil.which('gcc')
    assert gcc, 'gcc not found'
    exe = str(tmp_path / 'abi_demo')
    rocm = '/opt/rocm'
    cmd = [gcc, '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I%s/include' % rocm,
           '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'abi_demo.c'),
           '-o', exe, '-L' + CSRC, '-lrunlmc_hip', '-L%s/lib' % rocm, '-lamdhip64', '-lm',
           '-Wl,-rpath,' + CSRC, '-Wl,-rpath,%s/lib' % rocm]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'abi_demo ok' in out.stdout
    assert 'hip-gfx950' in out.stdout
