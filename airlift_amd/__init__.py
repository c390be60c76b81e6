"""airlift_amd -- MI355X-native re-alignment path for AirLift (drop-in for the aligner calls in
src/0-align_reads.sh, src/0-align_singletons.sh and src/3-align_gaps/align_gaps.sh of the reference).

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C-ABI of include/airlift.h);
this package is a thin ctypes mirror of that C-ABI for tests and bench.py.  There is no CPU path:
importing works anywhere, but creating a mapping context without a GPU raises."""
from .capi import (AirliftError, Index, Context, MapOpt, IdxOpt, Reg, lib_path, load, read_fastx, build)  # noqa: F401
