"""recboard_amd -- MI355X-native embedding-and-scoring engine behind the FreeRec/RecBoard model API.

Only the hot path of BASELINE.json `north_star` lives here:
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/recengine.h)
  lib.py           ctypes binding of librecengine.so (fails loudly when the library is missing)
  ops.py           torch-tensor front end of the C ABI (device memory + streams only)
"""
__version__ = "0.1.0"
