"""panogrf_amd — MI355X (gfx950) native implementation of the PanoGRF inference hot path.

The compute lives in `libpanogrf_hip.so` (hand-written HIP, C ABI in include/panogrf_hip.h); this
package is the thin Python host that mirrors the reference's renderer interface
(network/renderer.py) on top of it.  There is NO CPU / PyTorch fallback: if the library is
missing or no GPU is present the ops raise.
"""
from ._lib import lib, PgrfError, library_path  # noqa: F401

__all__ = ["lib", "PgrfError", "library_path"]
