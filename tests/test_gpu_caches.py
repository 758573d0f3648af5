"""What the front-end keeps between calls (demuxalot_amd/demux.py: _pack_on_device, _cached_variant_keys, _barcode_index):
the packed problem resident on the shared context, the key arrays of var2varid on the genotypes object, the pandas Index on
the barcode handler.  predict -> learn -> predict on the same inputs (examples/2-with-detection-of-new-SNPs.ipynb cells
12 / 16-18, tests/test_synthetic.py:184-190 of the reference) must pack once and still return the reference's bits; every
change of the inputs must be seen."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu


def test_predict_learn_predict_packs_once_and_every_change_of_the_inputs_repacks(monkeypatch):
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext, get_context
    fx = fio.load('f1_synthetic_default.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    packs = []
    original = DeviceContext.pack_staged_and_set_problem
    monkeypatch.setattr(DeviceContext, 'pack_staged_and_set_problem', lambda self, *a, **k: (packs.append(1), original(self, *a, **k))[1])
    get_context()._resident_key = None
    dp, clip = float(fx['predict0_dp']), float(fx['predict0_clip'])

    def predict():
        logits_df, probs_df = Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
        fio.assert_bitwise(logits_df.values, fx['predict0_logits'], 'predict logits')
        fio.assert_bitwise(probs_df.values, fx['predict0_probs'], 'predict posteriors')
        assert probs_df.index.name == 'BARCODE' and list(probs_df.index) == list(handler.ordered_barcodes)
        return probs_df

    def learn():
        kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']), doublet_prior=float(fx['em0_dp']))
        learnt, last = Demultiplexer.learn_genotypes(calls, genotypes, handler, **kwargs)
        fio.assert_bitwise(last.values, fx[f'em0_it{kwargs["n_iterations"] - 1}_probs'], 'learn posteriors')
        fio.assert_bitwise(learnt.variant_betas, fx['em0_learnt_betas'], 'learnt betas')
        assert last.index.name is None

    first = predict()
    assert len(packs) == 1
    learn()
    again = predict()
    learn()
    assert len(packs) == 1, 'the same containers and genotypes were packed again'
    assert first.index is not again.index  # every frame its own Index object (callers rename them)
    # another E-step on the shared context through another entry point replaces the resident problem: the next call packs
    Demultiplexer._compute_probs_from_betas(genotypes.get_snp_ids_for_variants(), genotypes.get_betas(), 0.01)
    predict()
    assert len(packs) == 2
    # a container edited in place (same arrays, same lengths)
    name = next(iter(calls))
    saved = calls[name].snp_calls['p_base_wrong'][:calls[name].n_snp_calls].copy()
    calls[name].snp_calls['p_base_wrong'][:calls[name].n_snp_calls] *= 0.5
    _, changed = Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
    assert len(packs) == 3 and not np.array_equal(changed.values, fx['predict0_probs'])
    calls[name].snp_calls['p_base_wrong'][:calls[name].n_snp_calls] = saved
    predict()
    assert len(packs) == 4
    # ONE record in the middle of a container edited in place (advisor, round 5: a sampled checksum does not see it): the key is
    # the content hash of every record
    n_calls = calls[name].n_snp_calls
    middle = n_calls // 2 + 1
    kept = calls[name].snp_calls['p_base_wrong'][middle]
    calls[name].snp_calls['p_base_wrong'][middle] = np.float32(0.25) if kept != np.float32(0.25) else np.float32(0.125)
    Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
    assert len(packs) == 5, 'a single edited record went unnoticed'
    calls[name].snp_calls['p_base_wrong'][middle] = kept
    kept_cb = calls[name].molecules['compressed_cb'][calls[name].n_molecules // 2]
    calls[name].molecules['compressed_cb'][calls[name].n_molecules // 2] = (kept_cb + 1) % handler.n_barcodes
    Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
    assert len(packs) == 6, 'a single edited molecule record went unnoticed'
    calls[name].molecules['compressed_cb'][calls[name].n_molecules // 2] = kept_cb
    predict()
    assert len(packs) == 7
    # the same records in OTHER arrays (copies): content-addressed, no identity in the key - not packed again
    import copy
    copies = {chrom: copy.deepcopy(c) for chrom, c in calls.items()}
    logits_c, probs_c = Demultiplexer.predict_posteriors(copies, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
    fio.assert_bitwise(probs_c.values, fx['predict0_probs'], 'posteriors on copies of the containers')
    assert len(packs) == 7
    # the switches: invalidate_resident(), DEMUXALOT_AMD_RESIDENT=0
    from demuxalot_amd import invalidate_resident
    invalidate_resident()
    predict()
    assert len(packs) == 8
    monkeypatch.setenv('DEMUXALOT_AMD_RESIDENT', '0')
    predict()
    predict()
    assert len(packs) == 10
    monkeypatch.delenv('DEMUXALOT_AMD_RESIDENT')
    predict()   # (the calls above left no key behind)
    predict()
    assert len(packs) == 11
    packs[:] = packs[:4]
    # another var2varid object with the same content: the keys are derived again, the result is the same
    genotypes.var2varid = dict(genotypes.var2varid)
    predict()
    assert len(packs) == 5
    # a genotypes object grown by one variant (rows of variant_betas beyond n_variants exist: capacity 32768)
    bigger = genotypes.clone()
    bigger.variant_betas = np.vstack([bigger.variant_betas[:bigger.n_variants], np.ones((1, bigger.n_genotypes), dtype=np.float32)])
    bigger.var2varid[('chr_new', 12345, 'A')] = bigger.n_variants
    logits_b, probs_b = Demultiplexer.predict_posteriors(calls, bigger, handler, p_genotype_clip=clip, doublet_prior=dp)
    assert len(packs) == 6 and probs_b.shape == fx['predict0_probs'].shape


def test_variant_keys_follow_the_mapping():
    """The key arrays kept on the genotypes object against a fresh walk, through reassignment and growth of var2varid."""
    from demuxalot_amd.demux import _cached_variant_keys, _variant_keys
    fx = fio.load('f2_synthetic_g4.npz')
    _calls, genotypes, _handler = fio.product_inputs(fx)

    def check():
        _fp, v2snp, keys, chrom_index = _cached_variant_keys(genotypes)
        want_keys, want_index = _variant_keys(genotypes)
        assert np.array_equal(v2snp, genotypes.get_snp_ids_for_variants()) and chrom_index == want_index
        for got, want in zip(keys, want_keys):
            assert np.array_equal(got, want)
        return keys

    first = check()
    assert check()[0] is first[0]  # kept
    items = list(genotypes.var2varid.items())
    genotypes.var2varid = dict(items[::-1])  # another object, another insertion order: other chromosome / SNP numbering
    second = check()
    assert second[0] is not first[0]
    n = genotypes.n_variants
    genotypes.var2varid[('zz', 7, 'C')] = n
    genotypes.variant_betas = np.vstack([genotypes.variant_betas[:n], np.ones((1, genotypes.n_genotypes), dtype=np.float32)])
    third = check()
    assert len(third[1]) == n + 1
