"""Deterministic synthetic workloads for benchmarks and large-size tests.

The reference ships no generator at these scales (its only one writes a BAM through pysam,
tests/test_synthetic.py:106-145), so this one emits the hot path's inputs directly:

  packed form  - the columns of the reference's `barcode_calls` (variant_id, compressed_cb,
                 p_base_wrong; variant-major like np.unique leaves them, demux.py:276-300),
                 `variant_index2snp_index`, and a genotype beta table shaped like
                 ProbabilisticGenotypes.add_vcf output (prior_strength 100, genotypes.py:147-164);
  object form  - for small sizes, CompressedSNPCalls / ProbabilisticGenotypes / BarcodeHandler
                 so that the full Python entry points can be exercised.

Generator: numpy PCG64(seed).  S SNPs with exactly two variants each (V = 2S: rows 2s = ref,
2s+1 = alt); alt-allele frequency ~ Beta(.5,.5) clipped to [.02,.98]; dosage ~ Binomial(2, f);
10 % of (SNP, donor) entries "not provided" (filled with 0.1 x mean of the provided ones);
calls per barcode ~ LogNormal(mean `calls_per_barcode`, sigma .6) clipped to [16, 4000]; SNP
popularity ~ LogNormal(0, 1.5); p_base_wrong = 10^(-q/10), q ~ U{10..40}, 15 % of calls carry
the product of two such values; duplicates of a (barcode, variant) pair are merged by float32
product as the reference's repack does.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class SyntheticProblem:
    n_barcodes: int
    n_snps: int
    n_genotypes: int
    v2snp: np.ndarray          # int32[V]
    raw_betas: np.ndarray      # float32[V, G]  (ProbabilisticGenotypes.variant_betas[:V])
    variant_id: np.ndarray     # int32[N]  variant-major (variant, then barcode)
    compressed_cb: np.ndarray  # int32[N]
    p_base_wrong: np.ndarray   # float32[N]
    truth: np.ndarray          # int32[B, 2] donors of each barcode (equal for singlets)
    default_prior: float = 1.0

    @property
    def n_variants(self):
        return len(self.v2snp)

    @property
    def n_calls(self):
        return len(self.variant_id)

    def prior_betas(self, add_data_prior=True):
        """Regularised betas exactly as pack_calls computes them (demux.py:367-388), with the number
        of molecules per variant taken as the number of unique calls (the generator emits one
        molecule per call before merging duplicates; merged ones count once)."""
        betas = self.raw_betas

        def share(per_variant, reg):
            per_snp = np.bincount(self.v2snp, weights=per_variant)[self.v2snp]
            return per_variant / (per_snp + reg)

        scale = 1.
        if add_data_prior:
            scale = scale + share(np.bincount(self.variant_id, minlength=self.n_variants), 100.)
        scale = scale + share(betas.sum(axis=1), 100.)
        return betas + (scale[:, None] * self.default_prior).astype(np.float32)

    def subset_barcodes(self, lo, hi):
        """Calls of barcodes [lo, hi) with barcode indices re-based to 0 (same order)."""
        keep = (self.compressed_cb >= lo) & (self.compressed_cb < hi)
        return self.variant_id[keep], (self.compressed_cb[keep] - lo).astype(np.int32), self.p_base_wrong[keep]


def make_genotype_betas(rng, n_snps, n_genotypes, sibling_pairs=False):
    f = np.clip(rng.beta(0.5, 0.5, size=n_snps), 0.02, 0.98)
    dosage = rng.binomial(2, f[:, None], size=(n_snps, n_genotypes)).astype(np.int8)
    if sibling_pairs:  # donors 2j and 2j + 1 carry the same genotype at half of the SNPs (related donors: hard to tell apart)
        same = rng.random((n_snps, n_genotypes // 2)) < 0.5
        odd = dosage[:, 1:2 * (n_genotypes // 2):2]
        odd[same] = dosage[:, 0:2 * (n_genotypes // 2):2][same]
    betas = np.empty((2 * n_snps, n_genotypes), dtype=np.float32)
    betas[0::2] = 50.0 * (2 - dosage)
    betas[1::2] = 50.0 * dosage
    missing = rng.random((n_snps, n_genotypes)) < 0.10
    missing[:, 0] = False  # keep at least one provided donor per SNP
    provided = (~missing).astype(np.float32)
    n_provided = provided.sum(axis=1)
    for half in (0, 1):
        rows = betas[half::2]
        mean_provided = (rows * provided).sum(axis=1) / n_provided
        rows[missing] = (0.1 * mean_provided)[:, None].repeat(n_genotypes, axis=1)[missing]
    return betas, dosage


def _generate_calls(rng, B, S, G, dosage, weights, calls_per_barcode, doublets, variant_major):
    """Barcodes and their calls for a given genotype table: (truth int32[B, 2], variant, cb, p_base_wrong), the unique
    (barcode, variant) calls variant-major (barcodes ascending inside a variant) or barcode-major."""
    V = 2 * S
    truth = np.empty((B, 2), dtype=np.int32)
    truth[:, 0] = rng.integers(0, G, size=B)
    truth[:, 1] = truth[:, 0]
    if doublets:
        pairs = rng.random(B) < 0.10
        truth[pairs, 1] = rng.integers(0, G, size=int(pairs.sum()))

    sigma = 0.6
    n_b = np.clip(np.rint(rng.lognormal(np.log(calls_per_barcode) - sigma ** 2 / 2, sigma, size=B)), 16, 4000)
    n_b = np.minimum(n_b, S).astype(np.int64)
    total = int(n_b.sum())
    cb = np.repeat(np.arange(B, dtype=np.int32), n_b)

    cum = np.cumsum(weights)
    cum /= cum[-1]
    snp = np.searchsorted(cum, rng.random(total, dtype=np.float32).astype(np.float64)).astype(np.int32)
    np.minimum(snp, S - 1, out=snp)

    donor = np.where(rng.random(total, dtype=np.float32) < 0.5, truth[cb, 0], truth[cb, 1])
    q = rng.integers(10, 41, size=total)
    e = (10.0 ** (-q / 10.0)).astype(np.float32)
    twice = rng.random(total, dtype=np.float32) < 0.15
    q2 = rng.integers(10, 41, size=int(twice.sum()))
    e[twice] = e[twice] * (10.0 ** (-q2 / 10.0)).astype(np.float32)
    alt = rng.random(total, dtype=np.float32) < (dosage[snp, donor] * np.float32(0.5))
    flip = rng.random(total, dtype=np.float32) < e
    alt ^= flip
    variant = (2 * snp + alt.astype(np.int32)).astype(np.int32)
    del snp, donor, q, twice, alt, flip

    # merge duplicates of (barcode, variant): float32 product in generation order
    key = cb.astype(np.int64) * V + variant
    order = np.argsort(key, kind='stable')
    key = key[order]
    first = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    p = np.multiply.reduceat(e[order], first).astype(np.float32)
    u_cb = (key[first] // V).astype(np.int32)
    u_variant = (key[first] % V).astype(np.int32)
    del key, order, e, cb, variant

    if variant_major:
        order = np.argsort(u_variant, kind='stable')  # barcodes stay ascending inside a variant
        u_variant, u_cb, p = u_variant[order], u_cb[order], p[order]
    return truth, np.ascontiguousarray(u_variant), np.ascontiguousarray(u_cb), np.ascontiguousarray(p)


def generate(n_barcodes, n_snps, n_genotypes, calls_per_barcode=400, doublets=False, seed=1234,
             variant_major=True, seed_calls=None, sibling_pairs=False) -> SyntheticProblem:
    """`seed` fixes the genotype table and SNP popularity; `seed_calls` (default: seed) fixes the
    barcodes and their calls, so that several shards can share one genotype table.  `sibling_pairs`: donors 2j / 2j + 1
    share their genotype at half of the SNPs (with few calls per barcode: the workload the guarded E-step can prove least of)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    B, S, G = int(n_barcodes), int(n_snps), int(n_genotypes)
    V = 2 * S
    betas, dosage = make_genotype_betas(rng, S, G, sibling_pairs)
    weights = rng.lognormal(0.0, 1.5, size=S)
    if seed_calls is not None and seed_calls != seed:
        rng = np.random.Generator(np.random.PCG64(seed_calls))
    v2snp = (np.arange(V, dtype=np.int32) // 2).astype(np.int32)
    truth, u_variant, u_cb, p = _generate_calls(rng, B, S, G, dosage, weights, calls_per_barcode, doublets, variant_major)
    return SyntheticProblem(B, S, G, v2snp, betas, u_variant, u_cb, p, truth)


def generate_sharded(n_barcodes, n_snps, n_genotypes, n_shards=8, calls_per_barcode=400, doublets=False, seed=1234,
                     threads=None) -> SyntheticProblem:
    """The same experiment model at sizes where one pass of generate() takes minutes and tens of gigabytes
    (BASELINE.json configs[4]: 1M barcodes x 650k SNPs x 128 genotypes, ~4e8 calls): ONE genotype table (seed), the
    barcodes cut into n_shards consecutive ranges whose calls are drawn independently (generator seed x 1000 + shard) by
    a pool of threads (numpy's generators, sorts and searches release the interpreter lock).  Deterministic for a given
    (seed, n_shards), whatever the number of threads.  The calls come shard after shard, variant-major inside a shard
    with ascending barcodes: for every variant the barcodes ascend over the whole array (the reference's bincount order,
    demux.py:113-118), for every barcode the variants ascend (demux.py:261) - as in generate()'s output."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.Generator(np.random.PCG64(seed))
    B, S, G = int(n_barcodes), int(n_snps), int(n_genotypes)
    V = 2 * S
    betas, dosage = make_genotype_betas(rng, S, G)
    weights = rng.lognormal(0.0, 1.5, size=S)
    v2snp = (np.arange(V, dtype=np.int32) // 2).astype(np.int32)
    bounds = [B * k // n_shards for k in range(n_shards + 1)]

    def shard(k):
        rng_k = np.random.Generator(np.random.PCG64(seed * 1000 + k))
        truth, v, cb, p = _generate_calls(rng_k, bounds[k + 1] - bounds[k], S, G, dosage, weights, calls_per_barcode, doublets, True)
        cb += np.int32(bounds[k])
        return truth, v, cb, p

    workers = max(1, min(n_shards, threads if threads else (os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        parts = list(pool.map(shard, range(n_shards)))
    truth = np.concatenate([t for t, _, _, _ in parts])
    sizes = [len(v) for _, v, _, _ in parts]
    variant, cb, p = np.empty(sum(sizes), np.int32), np.empty(sum(sizes), np.int32), np.empty(sum(sizes), np.float32)
    at = 0
    for k in range(n_shards):  # shard by shard, each freed as soon as it is copied
        _t, v_k, cb_k, p_k = parts[k]
        parts[k] = None
        variant[at:at + len(v_k)], cb[at:at + len(v_k)], p[at:at + len(v_k)] = v_k, cb_k, p_k
        at += len(v_k)
    return SyntheticProblem(B, S, G, v2snp, betas, variant, cb, p, truth)


def as_objects(problem: SyntheticProblem, n_chromosomes=3):
    """Object form (CompressedSNPCalls per chromosome, ProbabilisticGenotypes, BarcodeHandler) of a
    small problem: one molecule per call, SNP s placed on chromosome s % n_chromosomes."""
    from . import BarcodeHandler, CompressedSNPCalls, ProbabilisticGenotypes
    names = [f'Donor{g + 1:03d}' for g in range(problem.n_genotypes)]
    genotypes = ProbabilisticGenotypes(names, default_prior=problem.default_prior)
    chrom_of_snp = np.arange(problem.n_snps) % n_chromosomes
    pos_of_snp = 100 + 37 * (np.arange(problem.n_snps) // n_chromosomes)
    base_pairs = [('A', 'C'), ('C', 'T'), ('G', 'A'), ('T', 'G')]
    var2varid = {}
    for v in range(problem.n_variants):
        s = v // 2
        var2varid[(f'chr{chrom_of_snp[s] + 1}', int(pos_of_snp[s]), base_pairs[s % 4][v % 2])] = v
    genotypes.var2varid = var2varid
    genotypes.variant_betas = problem.raw_betas.copy()
    width = len(str(problem.n_barcodes))
    barcodes = [f'BC{b:0{width}d}-1' for b in range(problem.n_barcodes)]  # already sorted
    handler = BarcodeHandler(barcodes)
    calls = {}
    snp_of_call = problem.variant_id // 2
    base_index = np.asarray([['ACGT'.index(a), 'ACGT'.index(b)] for a, b in base_pairs], dtype=np.uint8)
    for c in range(n_chromosomes):
        sel = np.flatnonzero(chrom_of_snp[snp_of_call] == c)
        calls[f'chr{c + 1}'] = CompressedSNPCalls.from_arrays(
            compressed_cb=problem.compressed_cb[sel], snp_calls_molecule_index=np.arange(len(sel)),
            snp_position=pos_of_snp[snp_of_call[sel]],
            base_index=base_index[snp_of_call[sel] % 4, problem.variant_id[sel] % 2],
            p_base_wrong=problem.p_base_wrong[sel])
    return calls, genotypes, handler
