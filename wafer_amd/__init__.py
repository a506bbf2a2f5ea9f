"""wafer_amd -- MI355X-native engine for Wafer's grid::evolve hot path.

The product is the C-ABI library (include/wafer_hip.h, wafer_amd/csrc); this
package is the thin host-side mirror used by tests, bench.py and the slab
driver.  There is no CPU fallback: importing `engine` without the built
library, or creating a context without a GPU, raises.
"""
from .engine import (Context, Params, WaferError, POTENTIALS, INITIAL_CONDITIONS,  # noqa: F401
                     load_library, library_path)
