"""CompressedSNPCalls: the input wire format of the hot path (mirror of the container at
demuxalot/snp_counter.py:77-139; the BAM scanner that fills it is out of scope).

Two growable structured arrays whose first n_* entries are valid:
  molecules  (compressed_cb i32, compressed_ub i32, p_group_misaligned f32)
  snp_calls  (molecule_index i32, snp_position i32, base_index u8, p_base_wrong f32)
Demultiplexer reads only `[:n]` slices, so objects produced by the reference's count_snps can
be passed in unchanged (duck typing)."""
from typing import List

import numpy as np

from .utils import compress_base

MOLECULE_DTYPE = np.dtype([('compressed_cb', 'int32'), ('compressed_ub', 'int32'), ('p_group_misaligned', 'float32')])
SNP_CALL_DTYPE = np.dtype([('molecule_index', 'int32'), ('snp_position', 'int32'), ('base_index', 'uint8'),
                           ('p_base_wrong', 'float32')])


def _grown(array):
    return np.concatenate([array, array], axis=0)


class CompressedSNPCalls:
    def __init__(self, start_snps_size=1024, start_molecule_size=128):
        self.n_molecules = 0
        self.molecules = np.zeros(start_molecule_size, dtype=MOLECULE_DTYPE)
        self.molecules[:] = (-1, -1, -1.)
        self.n_snp_calls = 0
        self.snp_calls = np.zeros(start_snps_size, dtype=SNP_CALL_DTYPE)
        self.snp_calls[:] = (-1, -1, 255, -1.)

    def add_calls_from_read_group(self, compressed_cb, compressed_ub, p_group_misaligned, snps):
        while len(snps) + self.n_snp_calls > len(self.snp_calls):
            self.snp_calls = _grown(self.snp_calls)
        if self.n_molecules == len(self.molecules):
            self.molecules = _grown(self.molecules)
        m = self.n_molecules
        self.molecules[m] = (compressed_cb, compressed_ub, p_group_misaligned)
        self.n_molecules += 1
        for position, base, p_wrong in snps:
            self.snp_calls[self.n_snp_calls] = (m, position, compress_base(base), p_wrong)
            self.n_snp_calls += 1

    def minimize_memory_footprint(self):
        self.snp_calls = self.snp_calls[:self.n_snp_calls].copy()
        self.molecules = self.molecules[:self.n_molecules].copy()
        assert np.all(self.molecules['p_group_misaligned'] != -1)
        assert np.all(self.snp_calls['p_base_wrong'] != -1)

    @staticmethod
    def from_arrays(compressed_cb, snp_calls_molecule_index, snp_position, base_index, p_base_wrong,
                    compressed_ub=None, p_group_misaligned=0.01) -> 'CompressedSNPCalls':
        """Builds a container from plain arrays (used by tests, fixtures and the synthetic generator)."""
        out = CompressedSNPCalls(start_snps_size=1, start_molecule_size=1)
        out.molecules = np.zeros(len(compressed_cb), dtype=MOLECULE_DTYPE)
        out.molecules['compressed_cb'] = compressed_cb
        out.molecules['compressed_ub'] = np.arange(len(compressed_cb)) if compressed_ub is None else compressed_ub
        out.molecules['p_group_misaligned'] = p_group_misaligned
        out.snp_calls = np.zeros(len(snp_position), dtype=SNP_CALL_DTYPE)
        out.snp_calls['molecule_index'] = snp_calls_molecule_index
        out.snp_calls['snp_position'] = snp_position
        out.snp_calls['base_index'] = base_index
        out.snp_calls['p_base_wrong'] = p_base_wrong
        out.n_molecules, out.n_snp_calls = len(out.molecules), len(out.snp_calls)
        return out

    @staticmethod
    def concatenate(snp_calls_list: List['CompressedSNPCalls']) -> 'CompressedSNPCalls':
        """Joins containers of one chromosome, re-basing molecule indices."""
        shift = 0
        calls, molecules = [], []
        for part in snp_calls_list:
            c = part.snp_calls[:part.n_snp_calls].copy()
            c['molecule_index'] += shift
            calls.append(c)
            molecules.append(part.molecules[:part.n_molecules])
            shift += part.n_molecules
        out = CompressedSNPCalls(start_snps_size=1, start_molecule_size=1)
        out.molecules = np.concatenate(molecules)
        out.snp_calls = np.concatenate(calls)
        out.n_molecules, out.n_snp_calls = len(out.molecules), len(out.snp_calls)
        return out
