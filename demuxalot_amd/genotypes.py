"""ProbabilisticGenotypes: the Dirichlet-beta store the EM reads and returns (mirror of the core
of demuxalot/genotypes.py:18-78, 301-361).  No math lives here; the table is float32[capacity, G]
with a dict (chrom, pos, base) -> row.  `var2varid`, `variant_betas`, `genotype_names` and
`default_prior` stay plain writable attributes because callers (and the reference's tests,
tests/test_synthetic.py:101-102, 182, 213) assign them directly.

The bead-array / assignment-table importers of the reference (genotypes.py:170-265) are out of
scope.  Kept because they make the package usable without pysam/htslib: the parquet round-trip
(save_betas / add_prior_betas, genotypes.py:267-299, 336-358), which is the checkpoint format of
learnt genotypes, and `add_vcf` with a plain-text VCF parser that follows the reference's import
rules (genotypes.py:112-168)."""
from collections import defaultdict
from copy import deepcopy
from typing import Dict, List, Tuple
from warnings import warn

import numpy as np


_INITIAL_ROWS = 32768  # capacity the reference starts with (genotypes.py:33); doubled on demand


def variant_columns(var2varid):
    """The (chrom, pos, base) -> row dict as columns, in the dict's iteration order: (rows int64, chromosome codes int64,
    chromosome names in order of first appearance, positions int64, bases as a list of str), or None when a position
    does not fit 32 unsigned bits (callers then walk the dict).  Passes in C (map / fromiter / pandas.factorize) instead
    of a Python loop body per variant; zip(*keys) is NOT one of them (200k-argument call: slower than the loop)."""
    import pandas as pd
    from operator import itemgetter
    n = len(var2varid)
    rows = np.fromiter(var2varid.values(), dtype=np.int64, count=n)
    if n == 0:
        return rows, np.zeros(0, np.int64), [], np.zeros(0, np.int64), []
    keys = list(var2varid)
    try:
        pos = np.fromiter(map(itemgetter(1), keys), dtype=np.int64, count=n)
    except (TypeError, ValueError, OverflowError):
        return None
    if pos.min() < 0 or pos.max() >= 2 ** 32:
        return None
    chroms = np.empty(n, dtype=object)
    chroms[:] = list(map(itemgetter(0), keys))
    codes, names = pd.factorize(chroms)  # codes in order of first appearance
    return rows, codes.astype(np.int64), list(names), pos, list(map(itemgetter(2), keys))


def snp_ids_from_columns(columns):
    """get_snp_ids_for_variants from variant_columns(var2varid): SNPs numbered in the order they first appear."""
    import pandas as pd
    rows, chrom_codes, _names, pos, _bases = columns
    snp_of_key = pd.factorize((chrom_codes << 32) | pos)[0].astype(np.int32)
    out = np.full(len(rows), -1, dtype=np.int32)
    assert len(rows) == 0 or (rows.min() >= 0 and rows.max() < len(rows)), 'var2varid rows must enumerate 0..n_variants-1'
    out[rows] = snp_of_key
    assert (out >= 0).all(), 'var2varid rows must enumerate 0..n_variants-1'
    return out


class ProbabilisticGenotypes:
    """Dirichlet-beta table float32[capacity, G] + (chrom, pos, base) -> row.  Only the first n_variants rows are
    meaningful; genotype names must come sorted and unique (they are the posterior columns)."""

    def __init__(self, genotype_names: List[str], default_prior=1.):
        names = list(genotype_names)
        assert names == sorted(names), 'please order genotype names'
        assert len(set(names)) == len(names), f'Duplicates in genotypes: {genotype_names}'
        self.genotype_names: List[str] = names
        self.default_prior: float = default_prior
        self.var2varid: Dict[Tuple, int] = {}
        self.variant_betas: np.ndarray = np.zeros((_INITIAL_ROWS, len(names)), dtype=np.float32)

    def __repr__(self):
        n_contigs = len({key[0] for key in self.var2varid})
        return f'<ProbabilisticGenotypes: {self.n_variants} variants on {n_contigs} contigs, genotypes {self.genotype_names}>'

    @property
    def n_genotypes(self):
        return len(self.genotype_names)

    @property
    def n_variants(self) -> int:
        return len(self.var2varid)

    def get_betas(self) -> np.ndarray:
        """Read-only view of the rows in use (callers must not write through it: genotypes.py:50-54)."""
        used = self.variant_betas[:self.n_variants]
        used.setflags(write=False)
        return used

    def get_snp_ids_for_variants(self) -> np.ndarray:
        """SNP (= chromosome, position) of every variant row, numbered in the order the SNPs first appear in
        var2varid (genotypes.py:56-66; the E-step never looks at the numbers, the P-step only groups by them)."""
        if not self.var2varid:
            return np.zeros(0, dtype=np.int32)
        columns = variant_columns(self.var2varid)
        if columns is not None:
            return snp_ids_from_columns(columns)
        rows = np.fromiter(self.var2varid.values(), dtype=np.int64, count=len(self.var2varid))
        numbering, snp_of_key = {}, np.empty(len(rows), dtype=np.int32)
        for i, (chrom, pos, _base) in enumerate(self.var2varid):
            snp_of_key[i] = numbering.setdefault((chrom, pos), len(numbering))
        out = np.full(len(rows), -1, dtype=np.int32)
        assert len(rows) == 0 or (rows.min() >= 0 and rows.max() < len(rows)), 'var2varid rows must enumerate 0..n_variants-1'
        out[rows] = snp_of_key
        assert (out >= 0).all(), 'var2varid rows must enumerate 0..n_variants-1'
        return out

    def get_variant_id(self, chrom, pos, base):
        """Row of a variant, allocated on first use."""
        row = self.var2varid.get((chrom, pos, base))
        if row is None:
            row = len(self.var2varid)
            self.var2varid[(chrom, pos, base)] = row
            self.extend_variants(0)
            self.invalidate()
        return row

    def invalidate(self):
        """Drops the key arrays the front-end keeps on this object between calls (demux.py: _cached_variant_keys).  Every method
        here that changes var2varid calls it; call it yourself after editing var2varid by hand."""
        self.__dict__.pop('_amd_variant_keys', None)

    def extend_variants(self, n_samples=1):
        """Makes room for n_samples more rows than are registered (the table doubles, as the reference's does)."""
        needed = self.n_variants + n_samples
        capacity = len(self.variant_betas)
        if needed <= capacity:
            return
        while capacity < needed:
            capacity *= 2
        grown = np.zeros((capacity, self.variant_betas.shape[1]), dtype=self.variant_betas.dtype)
        grown[:len(self.variant_betas)] = self.variant_betas
        self.variant_betas = grown

    # ---- VCF import without htslib -----------------------------------------------------------
    def _check_imported_genotypes(self, imported_genotypes, allow_duplicates=False) -> Dict[str, int]:
        """Which of the imported sample names are ours (genotypes.py:80-110); returns name -> column."""
        seen, duplicates = set(), []
        for name in imported_genotypes:
            if name in seen and name not in duplicates:
                duplicates.append(name)
            seen.add(name)
        if duplicates:
            if allow_duplicates:
                warn(f'Duplicate genotypes found will be imported: {duplicates}')
            else:
                raise RuntimeError(f'Duplicate genotypes found in imported data: {duplicates}')
        ours = set(self.genotype_names)
        common = seen & ours
        if not common:
            raise RuntimeError(f'No genotypes to import, expected {ours}, got {seen}')
        if seen - ours:
            warn(f'Genotypes will not be imported: {seen - ours}')
        if ours - seen:
            print(f'Some of genotypes are not provided during import: {ours - seen}')
        return {name: self.genotype_names.index(name) for name in common}

    def add_vcf(self, vcf_file_name, prior_strength: float = 100.):
        """Adds the calls of a (plain-text or gzipped) VCF, following the reference's rules
        (genotypes.py:112-168): only records whose alleles are all single A/C/G/T bases and distinct;
        each donor's called alleles share `prior_strength` (a missing allele of a diploid call leaves
        its half unassigned); records with fewer than two genotyped donors are skipped (their variant
        rows stay allocated, as in the reference); donors without a call at a kept record receive
        0.1 x the mean of the genotyped donors. Positions are stored 0-based."""
        import gzip
        opener = gzip.open if str(vcf_file_name).endswith('.gz') else open
        samples, donor2column = None, None
        n_records = n_skipped = 0
        n_before = self.n_variants
        with opener(vcf_file_name, 'rt') as handle:
            for line in handle:
                if line.startswith('##') or not line.strip():
                    continue
                fields = line.rstrip('\n').split('\t')
                if line.startswith('#'):
                    samples = fields[9:]
                    continue
                assert samples is not None, 'VCF header line (#CHROM ...) is missing'
                n_records += 1
                chrom, pos1 = fields[0], int(fields[1])
                alleles = [fields[3]] + ([] if fields[4] in ('.', '') else fields[4].split(','))
                if any(len(a) != 1 for a in alleles):
                    print('skipping non-snp, alleles = ', tuple(alleles), chrom, pos1)
                    continue
                if donor2column is None:
                    donor2column = self._check_imported_genotypes(imported_genotypes=list(samples))
                if len(set(alleles)) != len(alleles) or any(a not in 'ACGT' for a in alleles):
                    n_skipped += 1
                    continue
                rows = [self.get_variant_id(chrom, pos1 - 1, a) for a in alleles]
                keys = fields[8].split(':')
                gt_slot = keys.index('GT')
                contribution = np.zeros([len(rows), self.n_genotypes], dtype='float32')
                for name, entry in zip(samples, fields[9:]):
                    if name not in donor2column:
                        continue
                    parts = entry.split(':')
                    gt = parts[gt_slot] if gt_slot < len(parts) else '.'
                    called = gt.replace('|', '/').split('/')
                    for token in called:
                        if token not in ('.', ''):
                            contribution[int(token), donor2column[name]] += prior_strength / len(called)
                not_provided = contribution.sum(axis=0) == 0
                if np.sum(~not_provided) < 2:
                    n_skipped += 1
                    continue
                contribution[:, not_provided] = contribution[:, ~not_provided].mean(axis=1, keepdims=True) * 0.1
                self.variant_betas[rows] += contribution
        if n_skipped > 0:
            print('skipped', n_skipped, 'SNVs')
        print(f'Parsed {n_records} SNPs, got {self.n_variants - n_before} novel variants')

    def get_chromosome2positions(self):
        """chromosome -> sorted unique SNP positions (what count_snps of the reference is handed)."""
        positions = defaultdict(set)
        for chrom, pos, _base in self.var2varid:
            positions[chrom].add(pos)
        if not positions:
            warn('Genotypes are empty. Did you forget to add vcf/betas?')
        return {chrom: np.array(sorted(found), dtype=int) for chrom, found in positions.items()}

    def get_snp_positions_set(self) -> set:
        return {key[:2] for key in self.var2varid}

    def _with_betas(self, external_betas: np.ndarray, _take=False) -> 'ProbabilisticGenotypes':
        """A copy of this object whose table is exactly `external_betas` ([n_variants, G] float32, non-negative):
        how learn_genotypes hands back the learnt genotypes (genotypes.py:327-334).  `_take` (the front-end's own freshly
        downloaded table: nobody else holds it): the array itself becomes the table instead of a copy of it."""
        assert external_betas.shape == (self.n_variants, self.n_genotypes)
        assert external_betas.dtype == self.variant_betas.dtype
        assert external_betas.size == 0 or external_betas.min() >= 0
        out = self._clone(with_betas=False)
        if _take and external_betas.flags.c_contiguous and external_betas.flags.owndata:
            external_betas.flags.writeable = True
            out.variant_betas = external_betas
        else:
            out.variant_betas = np.array(external_betas, copy=True)
        return out

    def clone(self):
        """Independent copy (genotypes.py:360-361 deep-copies)."""
        return self._clone(with_betas=True)

    def _clone(self, with_betas):
        # The keys of var2varid are tuples of str / int: immutable, so sharing them between the copies cannot be
        # observed, while deep-copying 10^5..10^6 of them dominated learn_genotypes end to end.
        out = object.__new__(type(self))
        for name, value in self.__dict__.items():
            if name == 'var2varid':
                out.var2varid = dict(value)
            elif name == 'variant_betas':
                out.variant_betas = value.copy() if with_betas else None
            elif name == '_amd_variant_keys':
                continue  # (below)
            else:
                setattr(out, name, deepcopy(value))
        # the key arrays the front-end keeps on the object (demux.py: _cached_variant_keys) describe the copy's mapping as
        # well - same entries, same order -: handed over under the copy's fingerprint, so that predict_posteriors on learnt
        # genotypes does not walk var2varid again
        cached = self.__dict__.get('_amd_variant_keys')
        if cached is not None and cached[0][0][0] == id(self.var2varid) and cached[0][0][1] == len(self.var2varid):
            fingerprint = ((id(out.var2varid),) + tuple(cached[0][0][1:]),) + tuple(cached[0][1:])
            out._amd_variant_keys = (fingerprint,) + tuple(cached[1:])
        return out

    # ---- checkpoint format of learnt genotypes (parquet) ---------------------------------
    def as_pandas_dataframe(self):
        import pandas as pd
        keys = sorted(self.var2varid.items())
        rows = np.asarray([row for _key, row in keys], dtype=np.int64)
        index = pd.MultiIndex.from_frame(pd.DataFrame({
            'CHROM': [k[0] for k, _ in keys], 'POS': [k[1] for k, _ in keys], 'BASE': [k[2] for k, _ in keys]}))
        return pd.DataFrame(data=self.variant_betas[:self.n_variants][rows], index=index, columns=self.genotype_names)

    def save_betas(self, path_or_buf):
        self.as_pandas_dataframe().to_parquet(path_or_buf)

    def add_prior_betas(self, prior_filename, *, prior_strength: float = 1.):
        import pandas as pd
        prior = pd.read_parquet(prior_filename) * prior_strength
        print('Provided prior information about genotypes:', [*prior.columns])
        missing = [g for g in self.genotype_names if g not in prior.columns]
        if missing:
            print(f'No information for genotypes: {missing}')
        frame = prior.index.to_frame()
        rows = []
        for key in zip(frame['CHROM'], frame['POS'], frame['BASE']):
            if key not in self.var2varid:
                self.extend_variants(1)
                self.var2varid[key] = self.n_variants
                self.invalidate()
            rows.append(self.var2varid[key])
        for g, name in enumerate(self.genotype_names):
            if name in prior.columns:
                np.add.at(self.variant_betas[:, g], rows, prior[name])
