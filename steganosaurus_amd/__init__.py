"""steganosaurus_amd -- MI355X-native hot path of the TurtleFFT steganography tool.

Only what the path needs lives here:
  csrc/                 hand-written HIP kernels (gfx950), the C ABI, the host walk, the C++ CLI
  libturtlefft_hip.so   the built product library (include/turtlefft_hip.h)
  binding.py            ctypes plumbing for tests / bench (no CPU fallback)
  synth.py              synthetic covers and secrets used by tests, bench and golden generation
"""
from .binding import BIN_DTYPE, Context, TfftError, Walk, load, make_bins, bins_to_triples, walk_jitter, bins_sort  # noqa: F401

__all__ = ["BIN_DTYPE", "Context", "TfftError", "Walk", "load", "make_bins", "bins_to_triples", "walk_jitter", "bins_sort"]
