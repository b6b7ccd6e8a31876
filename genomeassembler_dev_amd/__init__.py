"""genomeassembler_dev_amd — MI355X (gfx950) implementation of the de Bruijn graph construction/traversal and k-meric
breakage-probability scoring hot path of SahakyanLab/GenomeAssembler_dev (lib/DeNovoAssembler.cpp +
lib/BreakageScorer.cpp), behind the reference's own entry points.  Compute is hand-written HIP in libgasm.so, reached
through the C ABI in include/gasm.h; there is no CPU fallback."""
from . import qtable, seqio, solutions, synth  # noqa: F401
from ._lib import Context, GasmError, default_context  # noqa: F401
from .api import (ContigMatrix, Scaffolds, assemble_contigs, assemble_contigs_velvet, calc_breakscore, coverage_percent, get_contigs,  # noqa: F401
                  get_contigs_from_reads, get_kmers_from_reads, levenshtein, unpack_kmers)
from .batch import SegmentBatch  # noqa: F401
