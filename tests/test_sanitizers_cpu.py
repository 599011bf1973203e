"""
The host side of libraoteh_hip.so (schedule builder, lane-program simulation, the three
kernel source generators, argument checking) under AddressSanitizer + UBSan: `make -C
raoteh_amd/csrc debug-test` builds libraoteh_hip_debug.so (host code instrumented, device
code as usual; GPU sanitizers are not available on the pool) and runs tests/test_host_cpu.py
against it (sanitizer debug target).
"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'),
                    reason='hipcc not available')
def test_host_code_under_address_and_ub_sanitizers():
    if os.environ.get('RAOTEH_HIP_LIB', '').endswith('libraoteh_hip_debug.so'):
        pytest.skip('already running against the sanitizer build')
    env = dict(os.environ)
    env.pop('RAOTEH_HIP_LIB', None)
    proc = subprocess.run(['make', '-C', os.path.join(ROOT, 'raoteh_amd', 'csrc'), '-j4',
                           'debug-test'], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=1500)
    out = proc.stdout.decode()
    assert proc.returncode == 0, out[-4000:]
    assert 'passed' in out and 'ERROR: AddressSanitizer' not in out and \
        'runtime error' not in out, out[-4000:]
