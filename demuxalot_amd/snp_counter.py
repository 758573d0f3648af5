"""CompressedSNPCalls: the input wire format of the hot path (mirror of the container at
demuxalot/snp_counter.py:77-139; the BAM scanner that fills it is out of scope).

Two growable structured arrays whose first n_* entries are valid:
  molecules  (compressed_cb i32, compressed_ub i32, p_group_misaligned f32)
  snp_calls  (molecule_index i32, snp_position i32, base_index u8, p_base_wrong f32)
Demultiplexer reads only `[:n]` slices, so objects produced by the reference's count_snps can
be passed in unchanged (duck typing); this class only holds the arrays (from_arrays builds one from
plain columns) -- filling, growing and joining containers is the BAM scanner's business."""
import numpy as np

MOLECULE_DTYPE = np.dtype([('compressed_cb', 'int32'), ('compressed_ub', 'int32'), ('p_group_misaligned', 'float32')])
SNP_CALL_DTYPE = np.dtype([('molecule_index', 'int32'), ('snp_position', 'int32'), ('base_index', 'uint8'),
                           ('p_base_wrong', 'float32')])


class CompressedSNPCalls:
    def __init__(self, start_snps_size=1024, start_molecule_size=128):
        self.n_molecules = 0
        self.molecules = np.zeros(start_molecule_size, dtype=MOLECULE_DTYPE)
        self.molecules[:] = (-1, -1, -1.)
        self.n_snp_calls = 0
        self.snp_calls = np.zeros(start_snps_size, dtype=SNP_CALL_DTYPE)
        self.snp_calls[:] = (-1, -1, 255, -1.)

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
