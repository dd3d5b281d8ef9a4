"""MI355X-native CU depth/mode decision engine (the hot path behind HM's TEncCu::compressCtu).

Layout:
  csrc/fcu_engine.h      the CTU engine (device code, one wavefront per chain)
  csrc/fcu_kernels.hip   gfx950 kernel entry + C ABI (include/fcu.h) -> libfcu.so
  engine.py              ctypes binding + `TEncCu`-shaped host class
  synth.py               synthetic YUV generators
  sequence.py            picture-level driver: fast-decision schedule, slices as chains, deblocking, .yuv I/O
  lowdelay.py            lowdelay_P driver: P pictures referencing the previous filtered reconstruction
"""
from . import lowdelay, sequence, sharding, synth  # noqa: F401
from .engine import CuEngine, FrameParams, FcuError, lib_path, load_lib  # noqa: F401
