"""rapmap_amd -- MI355X-native drop-in for the `rapmap quasimap` hot path (RapMap v0.6.0).

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/qmap_mi355.h), api.py (host mirror
of the reference's call surface), synth.py (synthetic transcriptome / reads)."""
from .api import (QuasiIndex, QuasiMapper, QmError, QmOpts, default_opts, build_index, pack_reads,  # noqa: F401
                  FastxReader, ReadBatch, MappedStream, reserve_stream_memory, SamWriter, sam_header_text, sam_records_text,
                  HIT_DTYPE, INTERVAL_DTYPE, LIB_PATH, ABI_SYMBOLS)
