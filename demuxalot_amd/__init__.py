"""demuxalot_amd: MI355X-native implementation of demuxalot's Demultiplexer EM hot path.

Drop-in names of the reference package (demuxalot/__init__.py:3-7) that belong to the hot path:
"""
__version__ = '0.1.0'

from .utils import BarcodeHandler
from .snp_counter import CompressedSNPCalls
from .genotypes import ProbabilisticGenotypes
from .demux import Demultiplexer, DevicePosteriors, invalidate_resident

__all__ = ['BarcodeHandler', 'CompressedSNPCalls', 'ProbabilisticGenotypes', 'Demultiplexer', 'DevicePosteriors', 'invalidate_resident']
