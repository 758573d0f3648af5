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
    text = _compile(tmp_path)
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


def _compile(tmp_path):
    asm = tmp_path / 'kernels.s'
    if not asm.exists():
        subprocess.check_call([HIPCC, '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'),
                               '-I/opt/rocm/include', '--offload-arch=gfx950', '-fhip-fp32-correctly-rounded-divide-sqrt', '-fno-gpu-rdc',
                               '-S', '--cuda-device-only', os.path.join(CSRC, 'kernels.hip'), '-o', str(asm)], stderr=subprocess.DEVNULL)
    return asm.read_text()


def _main_loop(text, name):
    """The longest innermost loop of a kernel: (its lines, the lines between its label and its first vector memory instruction)."""
    start = text.index(name + ':')
    body = text[start:text.index('.Lfunc_end', start)].splitlines()
    best = None
    for i, line in enumerate(body):
        if 'Inner Loop Header' not in line:
            continue
        label = line.split(':')[0].strip()
        back = [j for j, other in enumerate(body) if re.search(r's_c?branch\w*\s+' + re.escape(label) + r'\b', other)]
        if back and (best is None or back[-1] - i > len(best)):
            best = body[i:back[-1] + 1]
    assert best is not None, f'{name}: no loop found'
    head = []
    for line in best[1:]:
        if re.match(r'\s+(buffer_load|global_load)', line):
            break
        head.append(line)
    return best, head


# Software-pipelined walks of the tile-major E-step: gathers of several batches in flight across the loop's back edge.
#   kernel -> (fewest `row_newbcast` DPP operands in the loop, most LDS reads in it per trip)
PIPELINED = {
    # (round 6: one broadcast of r per call feeding two v_fma_mix_f32 instead of a DPP operand in each of two additions: 64 per trip, was 96)
    '_ZN3dmx20k_estep_tiled_coarseILi2EEEvNS_9EstepArgsE': (64, 8),   # coarse pass, 33 .. 64 genotypes: records through DPP only (LDS: the slot's sums)
    '_ZN3dmx20k_estep_tiled_coarseILi4EEEvNS_9EstepArgsE': (64, 16),  # ... 17 .. 32 genotypes
    '_ZN3dmx13k_estep_tiledILi1ELb1ELb0EEEvNS_9EstepArgsE': (0, 64),  # fine pass (records through LDS)
}


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not installed')
def test_pipelined_walks_do_not_drain_at_the_loop_head(tmp_path):
    """A `break` after every step of an unrolled pipeline loop makes the compiler unify the exits into a block that also carries the
    back edge; the wait-count analysis then sees the loop's head reached from states in which a register's load was the last one
    issued and puts `s_waitcnt vmcnt(0)` there: the pipeline drained once per trip, same results.  The walks have ONE exit and a
    peeled remainder; this test keeps them that way, and the coarse pass's records on the DPP path (no LDS, no lane reads per field)."""
    text = _compile(tmp_path)
    for name, (min_dpp, max_lds_reads) in PIPELINED.items():
        loop, head = _main_loop(text, name)
        assert not any('vmcnt(0)' in line for line in head), f'{name}: the loop waits for every load at its head:\n' + '\n'.join(head[:6])
        drains = sum('vmcnt(0)' in line for line in loop)
        assert drains == 0, f'{name}: {drains} x s_waitcnt vmcnt(0) inside the pipelined loop'
        dpp = sum('row_newbcast' in line for line in loop)
        if 'tiled_coarse' in name:   # the binary16 operands are converted inside the additions (v_fma_mix_f32): no conversion instruction of its own
            assert not any('v_cvt_f32_f16' in line for line in loop), f'{name}: v_cvt_f32_f16 in the pipelined loop'
            assert sum('v_fma_mix_f32' in line for line in loop) >= 64, name
        lds_reads = sum(bool(re.match(r'\s+ds_read', line)) for line in loop)
        assert dpp >= min_dpp and lds_reads <= max_lds_reads, f'{name}: {dpp} DPP row broadcasts (>= {min_dpp}), {lds_reads} LDS reads (<= {max_lds_reads}) per trip'
