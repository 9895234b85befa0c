"""well_duplicates_amd - MI355X-native well-duplicate scanner (hot path of
EdinburghGenomics/well_duplicates' count_well_duplicates.py).

Host side stays Python with the reference's CLI and targets-file format; the compare +
tally path runs as hand-written HIP kernels for gfx950 behind the C ABI declared in
include/welldup.h (libwelldup.so, loaded with ctypes - see _lib.py).
"""
__version__ = "0.1.0"
