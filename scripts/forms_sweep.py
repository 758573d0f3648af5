"""Randomised bitwise comparison of the E-step forms (GPU box): python scripts/forms_sweep.py [n_trials] [first_seed]
Every trial: a random problem (genotypes, doublets or not, barcodes, heavy-tailed and empty rows), its importer-style
table (few distinct values per row) and the table after one M-step (all distinct); on both the E-step is run as
direct form, dictionary form (forced), packed form (forced) and packed form with the long rows on 64 lanes - logits and posteriors must be bit-identical across
forms and equal to the numpy oracle's logits, and so must the M-step that reads what each form's epilogue left."""
import os
import sys
import time
import numpy as np
os.environ.setdefault('DEMUXALOT_AMD_ESTEP', 'exact')  # bitwise comparisons: the exact mode (the library's default is the guarded one)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
import fixture_io as fio
from oracle import demux_oracle as oracle
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext

oracle.load_npsimd()
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
seen = {}
for trial in range(first, first + n_trials):
    rng = np.random.default_rng(90000 + trial)
    doublets = bool(rng.random() < 0.6)
    G = int(rng.integers(2, 41)) if doublets else int(rng.choice([2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 200, 256]))
    B = int(rng.choice([37, 200, 701, 2500]))
    S = int(rng.integers(60, 900))
    cpb = int(rng.choice([8, 40, 120, 300]))
    dp = float(rng.choice([0.1, 0.35])) if doublets else 0.0
    p = synth.generate(B, S, G, calls_per_barcode=min(cpb, S), doublets=doublets, seed=7000 + trial)
    variant, cb, e = p.variant_id, p.compressed_cb, p.p_base_wrong
    if rng.random() < 0.5:  # some barcodes lose all their calls
        drop = np.isin(cb, rng.choice(B, size=max(1, B // 20), replace=False))
        variant, cb, e = variant[~drop], cb[~drop], e[~drop]
    pen = Demultiplexer._doublet_penalties(G, dp)
    what = f'trial {trial}: G={G} K={len(pen)} B={B} S={S} cpb={cpb} dp={dp} N={len(cb)}'
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(B, p.n_variants, G, variant, cb, e, p.v2snp)
        ctx.set_betas(p.prior_betas(add_data_prior=False))
        ctx.set_addition(None)
        for stage in ('importer table', 'after one M-step'):
            table = ctx.probs_from_betas(0.01)
            want = oracle.barcode_logits(variant, cb, e, table, B, dp, log_impl='npsimd')
            results = {}
            for form, (dmode, pmode) in {'direct': ('never', 'never'), 'dict': ('always', 'never'), 'packed': ('never', 'always'), 'split': ('never', 'split')}.items():
                ctx.set_estep_dictionary(dmode)
                ctx.set_estep_packing(pmode)
                logits, probs = ctx.estep(pen, with_doublets=doublets)
                ran = ctx.estep_form()[0]
                seen[ran] = seen.get(ran, 0) + 1
                results[form] = (logits, probs, ctx.mstep(2.), ran)
            for form in ('dict', 'packed', 'split'):
                for i, name in enumerate(('logits', 'posteriors', 'additions')):
                    fio.assert_bitwise(results[form][i], results['direct'][i], f'{what} [{stage}] {name}: {results[form][3]} vs direct')
            fio.assert_bitwise(results['direct'][0], want, f'{what} [{stage}] logits vs oracle')
            ctx.set_addition(results['direct'][2])
        print('ok', what, {k: v[3] for k, v in results.items()}, flush=True)
    finally:
        ctx.close()
print(f'{n_trials} trials bit-identical across forms and to the oracle in {time.time() - t0:.0f} s; E-steps by form: {seen}')
