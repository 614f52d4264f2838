"""fqcomp28_amd -- MI355X-native block entropy coder for fqcomp28-compatible streams.

The product is fqcomp28_amd/libfqgpu.so (HIP kernels behind the C ABI of
include/fqgpu.h).  This package is the thin ctypes binding the tests and bench.py
drive it through; the C++ drop-in surface for the reference is
fqcomp28_amd/csrc/workspace.hpp.  There is no CPU fallback: without the built
extension or without a GPU every call raises.
"""
from .binding import (  # noqa: F401
    FqgpuError, REC_DTYPE, SEQ_FT_DTYPE, QUAL_FT_DTYPE, lib, lib_path, build,
    device_count, bound_seq, bound_qual, parse_fastq, synth_fastq, freq_tables,
    tables_from_counts, Context, DBlock, F_WRITE_BACK_N, F_DECODE_INDEX, memcompress, memdecompress, pinned_empty,
)
