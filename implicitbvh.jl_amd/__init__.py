"""MI355X-native implicit-BVH engine: host-side mirror of ImplicitBVH.jl's hot-path API
(BVH / traverse / traverse_rays / BVHOptions) over the libibvh C ABI (include/ibvh.h).

The compute path is hand-written HIP for gfx950 in csrc/; this package only allocates device
buffers (torch tensors), fills descriptors and calls the C entry points.  There is no CPU
fallback: importing `api` objects works anywhere, but every operation raises if libibvh.so is
missing or no GPU is present.
"""
from . import abi  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import __all__  # noqa: F401
