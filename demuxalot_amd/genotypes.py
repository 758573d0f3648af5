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


class ProbabilisticGenotypes:
    def __init__(self, genotype_names: List[str], default_prior=1.):
        self.var2varid: Dict[Tuple, int] = {}
        self.genotype_names: List[str] = list(genotype_names)
        assert (np.sort(self.genotype_names) == self.genotype_names).all(), 'please order genotype names'
        assert len(set(genotype_names)) == len(genotype_names), f'Duplicates in genotypes: {genotype_names}'
        self.variant_betas: np.ndarray = np.zeros([32768, self.n_genotypes], 'float32')
        self.default_prior: float = default_prior

    def __repr__(self):
        contigs = {chrom for chrom, _, _ in self.var2varid}
        return (f'<Genotypes with {self.n_variants} variants on {len(contigs)} contigs ("chromosomes") '
                f'and {self.n_genotypes} genotypes: \n{self.genotype_names}')

    @property
    def n_genotypes(self):
        return len(self.genotype_names)

    @property
    def n_variants(self) -> int:
        return len(self.var2varid)

    def get_betas(self) -> np.ndarray:
        view = self.variant_betas[:self.n_variants]
        view.flags.writeable = False
        return view

    def get_snp_ids_for_variants(self) -> np.ndarray:
        """SNP id of every variant row; ids are handed out in first-seen order of var2varid."""
        ids = {}
        out = np.full(self.n_variants, -1, dtype='int32')
        for (chrom, pos, _base), row in self.var2varid.items():
            out[row] = ids.setdefault((chrom, pos), len(ids))
        assert np.all(out >= 0)
        assert np.all(out < self.n_variants)
        return out

    def get_variant_id(self, chrom, pos, base):
        key = chrom, pos, base
        if key not in self.var2varid:
            self.var2varid[key] = self.n_variants
            self.extend_variants(1)
        return self.var2varid[key]

    def extend_variants(self, n_samples=1):
        while n_samples + self.n_variants > len(self.variant_betas):
            self.variant_betas = np.concatenate([self.variant_betas, np.zeros_like(self.variant_betas)], axis=0)

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
        by_chrom = defaultdict(list)
        for chrom, pos, _base in self.var2varid:
            by_chrom[chrom].append(pos)
        if len(by_chrom) == 0:
            warn('Genotypes are empty. Did you forget to add vcf/betas?')
        return {chrom: np.unique(np.asarray(p, dtype=int)) for chrom, p in by_chrom.items()}

    def get_snp_positions_set(self) -> set:
        return {(chrom, pos) for chrom, pos, _base in self.var2varid}

    def _with_betas(self, external_betas: np.ndarray) -> 'ProbabilisticGenotypes':
        """Copy of the genotypes carrying new beta weights."""
        assert external_betas.shape == (self.n_variants, self.n_genotypes)
        assert external_betas.dtype == self.variant_betas.dtype
        assert np.min(external_betas) >= 0
        out = self._clone(with_betas=False)
        out.variant_betas = external_betas.copy()
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
            else:
                setattr(out, name, deepcopy(value))
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
            rows.append(self.var2varid[key])
        for g, name in enumerate(self.genotype_names):
            if name in prior.columns:
                np.add.at(self.variant_betas[:, g], rows, prior[name])
