"""Helpers turning the golden .npz fixtures (tests/golden/) back into inputs."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL = [f'f3_small_{i}.npz' for i in range(5)]
SYNTH = ['f2_synthetic_g4.npz', 'f1_synthetic_default.npz', 'f6_shipped_example.npz']


def load(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def oracle_calls(fx):
    """list of per-chromosome dicts in the reference dict's order (oracle convention)."""
    out = []
    for i, chrom in enumerate(fx['chroms']):
        out.append(dict(chrom=str(chrom), mol_cb=fx[f'c{i}_mol_cb'], call_mol=fx[f'c{i}_call_mol'],
                        call_pos=fx[f'c{i}_call_pos'], call_base=fx[f'c{i}_call_base'], call_p=fx[f'c{i}_call_p']))
    return out


def oracle_geno(fx, betas=None):
    return dict(var_chrom=[str(c) for c in fx['var_chrom']], var_pos=fx['var_pos'], var_base=fx['var_base'],
                var_row=fx['var_row'], betas=fx['betas'] if betas is None else betas,
                default_prior=float(fx['default_prior']))


def product_inputs(fx, betas=None):
    """(calls dict, ProbabilisticGenotypes, BarcodeHandler) built with demuxalot_amd's own classes,
    the way a user of the reference would hold them."""
    from demuxalot_amd import BarcodeHandler, CompressedSNPCalls, ProbabilisticGenotypes
    calls = {}
    for i, chrom in enumerate(fx['chroms']):
        calls[str(chrom)] = CompressedSNPCalls.from_arrays(
            fx[f'c{i}_mol_cb'], fx[f'c{i}_call_mol'], fx[f'c{i}_call_pos'], fx[f'c{i}_call_base'], fx[f'c{i}_call_p'],
            compressed_ub=fx[f'c{i}_mol_ub'], p_group_misaligned=fx[f'c{i}_mol_pmis'])
    genotypes = ProbabilisticGenotypes([str(s) for s in fx['genotype_names']], default_prior=float(fx['default_prior']))
    genotypes.var2varid = {(str(c), int(p), 'ACGTN'[int(b)]): int(r)
                           for c, p, b, r in zip(fx['var_chrom'], fx['var_pos'], fx['var_base'], fx['var_row'])}
    genotypes.variant_betas = np.array(fx['betas'] if betas is None else betas, dtype=np.float32)
    handler = BarcodeHandler([str(b) for b in fx['barcodes']])
    assert handler.ordered_barcodes == [str(b) for b in fx['barcodes']]
    return calls, genotypes, handler


def assert_bitwise(a, b, what=''):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    view = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    same = np.ascontiguousarray(a).view(view) == np.ascontiguousarray(b).view(view)
    if not same.all():
        bad = np.argwhere(~same.reshape(a.shape))
        raise AssertionError(f'{what}: {len(bad)} of {a.size} differ; first at {bad[0]}: '
                             f'{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}; max|d|={np.nanmax(np.abs(a - b))}')
