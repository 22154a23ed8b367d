"""oracle/raster_oracle64.py -- TEST INFRASTRUCTURE ONLY.

The fp64 build of the C restatement (oracle/raster_ref.c with -DED3REF_FP64 -> libraster_ref64.so) behind the same
functions as oracle/raster_oracle.py: this module executes that file's text with fp64 array / argument types.  Used by
tests/test_oracle_pins_cpu.py for (1) central differences of the oracle's forward against its hand-derived backward and
(2) the 1e-9-level cross-check with the independent autograd restatement (oracle/torch_raster.py)."""
import os

_PRECISION = "f64"
_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_oracle.py")
exec(compile(open(_src).read(), _src, "exec"), globals())
