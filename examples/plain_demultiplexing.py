"""
Simple demultiplexing with known genotypes -- the reference's examples/1-plain_demultiplexing.py with the
EM hot path on an MI355X.

With the reference installed the only change to a user script is the import of Demultiplexer:

    from demuxalot import BarcodeHandler, ProbabilisticGenotypes, count_snps      # reference front-end (needs pysam)
    from demuxalot_amd import Demultiplexer                                      # GPU hot path

    genotypes = ProbabilisticGenotypes(genotype_names=['Donor01', 'Donor02', 'Donor03', 'Donor04'])
    genotypes.add_vcf('test_genotypes.vcf')
    barcode_handler = BarcodeHandler.from_file('test_barcodes.csv')
    snps = count_snps(bamfile_location='test_bamfile.bam', chromosome2positions=genotypes.get_chromosome2positions(),
                      barcode_handler=barcode_handler)
    learnt_genotypes, posterior_probabilities = Demultiplexer.learn_genotypes(
        snps, genotypes=genotypes, barcode_handler=barcode_handler, doublet_prior=0.25)

This script runs the same steps without pysam: genotypes and barcodes are read by this package
(ProbabilisticGenotypes.add_vcf has a plain-text VCF reader), and the BAM scan -- which stays the reference's job --
is replaced by the calls that the reference's count_snps produced for its shipped example BAM (tests/golden/f6).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from demuxalot_amd import BarcodeHandler, Demultiplexer, ProbabilisticGenotypes  # noqa: E402
from tests import fixture_io as fio  # noqa: E402

genotypes = ProbabilisticGenotypes(genotype_names=['Donor01', 'Donor02', 'Donor03', 'Donor04'])
genotypes.add_vcf(os.path.join(fio.GOLDEN, 'example_genotypes.vcf'))
print(f'Loaded genotypes: {genotypes}')

barcode_handler = BarcodeHandler.from_file(os.path.join(fio.GOLDEN, 'example_barcodes.csv'))
print(f'Loaded barcodes: {barcode_handler}')

snps, _genotypes_from_fixture, _handler = fio.product_inputs(fio.load('f6_shipped_example.npz'))
print('Collected SNPs: ')
for chromosome, snps_in_chromosome in snps.items():
    print(f'Chromosome {chromosome}, {snps_in_chromosome.n_snp_calls} calls in {snps_in_chromosome.n_molecules} mols')

learnt_genotypes, posterior_probabilities = Demultiplexer.learn_genotypes(
    snps, genotypes=genotypes, barcode_handler=barcode_handler, doublet_prior=0.25)

print('Result:')
print(posterior_probabilities.round(3))
assigned = posterior_probabilities[posterior_probabilities.max(axis=1) > 0.9].idxmax(axis=1)
print(assigned.value_counts())
