"""Helpers turning the golden .npz fixtures (tests/golden/) back into inputs."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL = [f'f3_small_{i}.npz' for i in range(5)]
SYNTH = ['f2_synthetic_g4.npz', 'f1_synthetic_default.npz']


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
