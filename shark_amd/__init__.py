"""shark_amd -- MI355X (gfx950) implementation of shark's k-mer classification
hot path.  The product is libsharkhip.so (HIP kernels behind the C ABI in
include/shark_hip.h) and the `shark` CLI built from shark_amd/csrc; this
package only carries the ctypes plumbing used by tests and bench.py."""
from .capi import SharkHip, SharkHipError, load, LIB_PATH, EXPORTS  # noqa: F401
