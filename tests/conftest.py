import os
import sys

import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite pins the BIT-EXACT E-step unless a test selects a mode itself: most GPU tests assert bitwise equality with
# the reference's captured outputs / the oracle, which is the contract of DMX_ESTEP_EXACT.  The library's default mode
# (guarded: contract proven per barcode, demuxalot_amd/device.py: DEFAULT_ESTEP_MODE) is what tests/test_gpu_guarded.py,
# the guarded tests of tests/test_gpu_configs.py, the bench runs of tests/test_gpu_ranks_on_one_gpu.py and
# __graft_entry__.smoke() run.
os.environ.setdefault('DEMUXALOT_AMD_ESTEP', 'exact')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure). Builds oracle/libnpsimd.so on first use."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    from oracle import demux_oracle
    demux_oracle.load_npsimd()
    return demux_oracle
