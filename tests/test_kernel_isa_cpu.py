"""The 64-lane E-step kernels are scalar-fed: a wavefront owns one barcode, so its call records (row offset, keep, floor) are
wave-uniform and reach the gathers and the packed arithmetic through SCALAR loads.  The compiler only emits scalar loads for
memory it can prove unwritten ahead of the load: twice in round 5 an innocent-looking change - a store at the top of the kernel
(a timestamp; the queue of a guarded E-step that runs direct), a base pointer chosen at run time - turned every record load
into a vector load + v_readfirstlane, and the kernels ran 1.4x - 2x slower with bit-identical results: nothing a parity test
sees.  This test compiles csrc/kernels.hip to gfx950 assembly (no GPU needed) and counts."""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'demuxalot_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

# kernel (mangled) -> (fewest scalar loads, most vector global loads, most v_readfirstlane / v_readlane)
KERNELS = {
    '_ZN3dmx14k_estep_directILi64ELi1ELb0ELi8ELb0EEEvNS_9EstepArgsE': (20, 8, 4),    # exact, 64 genotypes (the redo of every guarded E-step)
    '_ZN3dmx14k_estep_directILi64ELi1ELb0ELi8ELb1EEEvNS_9EstepArgsE': (38, 8, 6),    # tolerance arithmetic, one barcode per wavefront
    '_ZN3dmx14k_estep_directILi64ELi2ELb0ELi4ELb0EEEvNS_9EstepArgsE': (17, 12, 4),   # exact, 65 - 128 genotypes
    '_ZN3dmx14k_estep_directILi64ELi1ELb1ELi8ELb0EEEvNS_9EstepArgsE': (20, 9, 4),    # exact, doublets up to 64 options
}


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not installed')
def test_scalar_fed_kernels_keep_their_scalar_loads(tmp_path):
    asm = tmp_path / 'kernels.s'
    subprocess.check_call([HIPCC, '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'),
                           '-I/opt/rocm/include', '--offload-arch=gfx950', '-fhip-fp32-correctly-rounded-divide-sqrt', '-fno-gpu-rdc',
                           '-S', '--cuda-device-only', os.path.join(CSRC, 'kernels.hip'), '-o', str(asm)], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    for name, (min_scalar, max_vector, max_readlane) in KERNELS.items():
        start = text.index(name + ':')
        body = text[start:text.index('.Lfunc_end', start)]
        ops = collections.Counter(line.split()[0] for line in (l.strip() for l in body.splitlines())
                                  if line and not line.startswith((';', '.')) and not re.match(r'^\S+:$', line))
        scalar = sum(n for op, n in ops.items() if op.startswith('s_load'))
        vector = sum(n for op, n in ops.items() if op.startswith('global_load'))
        readlane = ops['v_readfirstlane_b32'] + ops['v_readlane_b32']
        assert scalar >= min_scalar and vector <= max_vector and readlane <= max_readlane, \
            f'{name}: {scalar} scalar loads (>= {min_scalar}), {vector} vector global loads (<= {max_vector}), {readlane} lane reads (<= {max_readlane}): ' \
            'the record stream is no longer fetched through the scalar cache (a store or a selected pointer ahead of the loads?)'
