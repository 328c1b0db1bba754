"""ctunet_amd -- MI355X-native 3D U-Net path behind the ctunet model-class API.

The kernels live in libctunet_hip.so (C ABI: include/ctunet_hip.h); this package is the host
side that mirrors ``ctunet.pytorch.models`` / ``ctunet.utilities`` / ``ctunet.pytorch.ProblemHandler``
for the hot path only.  Importing it does not need a GPU; running a model does.
"""
from . import _lib  # noqa: F401
from .models import (UNet, UNet4_2IC, UNet4b1i3o, UNet4b2i3o, UNet5b2i3o, UNetDO, UNetSP, UNetSPSmall,  # noqa: F401
                     recAE_v2_fixed)

__all__ = ["UNet", "UNet4b2i3o", "UNet5b2i3o", "UNet4b1i3o", "UNetSP", "UNetSPSmall", "UNetDO", "recAE_v2_fixed",
           "UNet4_2IC"]
