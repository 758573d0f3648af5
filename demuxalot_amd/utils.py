"""Small host-side containers the hot path reads (mirror of the parts of demuxalot/utils.py
that Demultiplexer touches: BarcodeHandler.n_barcodes / ordered_barcodes, utils.py:39-66;
base compression, utils.py:24-32).  BAM-facing behaviour is out of scope for this package;
`get_barcode_index` is kept so that the reference's BAM scanner can use this class as is."""
from collections import Counter
from pathlib import Path

BASE_INDEX = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4}


def compress_base(base: str) -> int:
    return BASE_INDEX[base]


def decompress_base(base_index: int) -> str:
    return 'ACGTN'[base_index]


class BarcodeHandler:
    """Maps barcode strings (optionally paired with an RG tag) to dense integers.
    The *sorted* barcode order defines the row order of every output (utils.py:60-61)."""

    def __init__(self, barcodes, RG_tags=None, tag='CB'):
        assert not isinstance(barcodes, (str, Path)), 'construct by passing list of possible barcodes'
        items = list(barcodes)
        self.use_rg = RG_tags is not None
        if self.use_rg:
            rgs = list(RG_tags)
            assert len(items) == len(rgs), 'RG tags should be the same length as barcodes'
            items = list(zip(items, rgs))
        assert len(set(items)) == len(items), 'all passed barcodes should be unique'
        self.ordered_barcodes = sorted(items)
        self.barcode2index = {bc: i for i, bc in enumerate(self.ordered_barcodes)}
        self.tag = tag

    @property
    def n_barcodes(self):
        return len(self.barcode2index)

    def get_barcode_index(self, read):
        """None when the read's barcode is not white-listed, else its dense index."""
        if not read.has_tag(self.tag):
            return None
        key = read.get_tag(self.tag)
        if self.use_rg:
            key = (key, read.get_tag('RG'))
        return self.barcode2index.get(key)

    @staticmethod
    def from_file(barcodes_filename, **kwargs):
        import pandas as pd
        barcodes = pd.read_csv(barcodes_filename, header=None)[0].values.astype('str')
        return BarcodeHandler(barcodes, **kwargs)

    def filter_to_rg_value(self, rg_value):
        assert self.use_rg
        out = BarcodeHandler(self.barcode2index, tag=self.tag)
        out.barcode2index = {(bc if rg == rg_value else i): i for (bc, rg), i in self.barcode2index.items()}
        out.ordered_barcodes = list(out.barcode2index)
        out.use_rg = False
        return out

    def __repr__(self):
        if not self.use_rg:
            return f'<BarcodeHandler with {self.n_barcodes} barcodes>'
        stats = Counter(rg for _bc, rg in self.barcode2index)
        return f'<BarcodeHandler with {self.n_barcodes} barcodes. Number of barcodes for RG codes: {stats}>'
