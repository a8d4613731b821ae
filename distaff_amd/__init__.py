"""distaff_amd -- MI355X-native STARK prover backend for Distaff (GuildOfWeavers/distaff v0.5.1).

The product is ``libdistaff_hip.so`` (hand-written HIP kernels for gfx950 behind the C-ABI of ``include/distaff_hip.h``);
this package is its Python binding plus the multi-GPU orchestration.  There is no CPU fallback.
"""
from .lib import (Calibration, Comm, Context, DistaffError, DST_ERR_AIR, DST_ERR_ARG, DST_ERR_COMM, DST_ERR_HIP, DST_ERR_STATE, DST_OK, EXPORTS, LIB_PATH, arr_to_ints, blake3, fibonacci_trace, ints_to_arr, load,  # noqa: F401
                  prng_vector, prove_sharded_local, query_positions, use_test_hooks, use_product, library_path, load_hooks, PRODUCT_LIB, HOOKS_LIB)
