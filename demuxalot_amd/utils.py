"""What the hot path reads of the reference's BarcodeHandler (demuxalot/utils.py:60-66): the sorted barcode
list, which defines the row order of every output, and the number of barcodes.  Everything BAM-facing (read
tags, RG handling) stays with the reference's class, whose instances Demultiplexer accepts unchanged."""


class BarcodeHandler:
    def __init__(self, barcodes):
        assert not isinstance(barcodes, (str, bytes)), 'pass the list of barcodes, not a file name'
        self.ordered_barcodes = sorted(barcodes)
        self.barcode2index = {barcode: row for row, barcode in enumerate(self.ordered_barcodes)}
        assert len(self.barcode2index) == len(self.ordered_barcodes), 'all passed barcodes should be unique'

    @property
    def n_barcodes(self):
        return len(self.ordered_barcodes)

    @classmethod
    def from_file(cls, path):
        """One barcode per line (the format of the reference's example_data/test_barcodes.csv)."""
        with open(path) as lines:
            return cls([line.strip() for line in lines if line.strip()])
