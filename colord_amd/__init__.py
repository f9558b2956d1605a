"""colord_amd — MI355X-native long-read compression hot path (CoLoRd-compatible).

The compute path lives in ``colord_amd/csrc`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/colord_hip.h``).  The Python in this package is plumbing only: ctypes bindings, FASTQ/archive
host I/O used by the tests and the benchmark, and the synthetic-data generator.
"""
__version__ = "0.1.0"
