"""densecap_amd -- MI355X (gfx950) native dense-captioning inference path.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/densecap.h),
`_lib` (ctypes binding), `ops` (per-op wrappers named after the reference's modules) and
`model.DenseCapModel` (host mirror of densecap/DenseCapModel.lua's test-time API).
"""
from .model import DenseCapModel  # noqa: F401
from .ops import Context  # noqa: F401
