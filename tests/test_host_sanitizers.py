"""The host shim of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.

`make -C demuxalot_amd/csrc asan` builds demuxalot_amd/libdemux_host_asan.so: dmx_api.cpp + pack_host.cpp, with every
GPU-side symbol replaced at link time by csrc/host_stubs.cpp (device memory is malloc memory, the kernels' launchers do
nothing).  tests/sanitizer_driver.py then runs, in a subprocess that preloads the sanitizer runtime, a fuzz of
dmx_pack_calls_host against the oracle, a fuzz of dmx_exchange_slices against its definition, whole runs of the context
API on random problems (single, and with host-staged collectives of 2 .. 5 ranks, in every exchange mode) and the
error contract that replaces the reference's asserts (demux.py:78,81,98,135,160,317,359,374).  Any sanitizer report
aborts the subprocess.  Never runs on the GPU box (not a `gpu` test) and is never loaded by the product."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    compiler = shutil.which(os.environ.get('ASAN_CXX', 'g++'))
    if compiler is None:
        return None
    path = subprocess.run([compiler, f'-print-file-name={name}'], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.parametrize('seed', [0, 1])
def test_host_shim_under_address_and_ub_sanitizers(seed):
    asan = _runtime('libasan.so')
    if asan is None:
        pytest.skip('no g++ / libasan on this host')
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'demuxalot_amd', 'csrc'), 'asan'])
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    env = dict(os.environ, LD_PRELOAD=asan, PYTHONPATH=ROOT,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:allocator_may_return_null=1',  # leak checking would see the interpreter's own
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'sanitizer_driver.py'), str(seed), '150'],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0 and 'sanitizer driver ok' in done.stdout, (done.stdout[-1500:], done.stderr[-4000:])
    assert 'ERROR: AddressSanitizer' not in done.stderr and 'runtime error:' not in done.stderr, done.stderr[-4000:]


def test_pack_and_slices_fuzz_on_the_shipped_library(oracle):
    """The same two fuzzes (hypothesis-driven seeds) against the host-only entry points of the SHIPPED libdemux_hip.so -
    dmx_pack_calls_host and dmx_exchange_slices need no GPU - so that what the sanitizer build checks for memory safety
    is checked here for its results: variant matching + de-duplication bit-exact against the oracle's restatement of
    demux.py:276-300, 332-365; variant slices against their definition."""
    import numpy as np
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from demuxalot_amd import _lib
    from tests import sanitizer_driver as driver
    lib = _lib.load()

    @settings(max_examples=120, deadline=None)
    @given(st.integers(0, 2 ** 32 - 1))
    def run(seed):
        rng = np.random.default_rng(seed)
        driver.fuzz_pack(lib, oracle, rng, 1)
        driver.fuzz_slices(lib, rng, 2)

    run()
