"""liodom_amd — MI355X-native LiODOM hot path (edge extraction, edge-to-line correspondence,
pose solve, sliding-window map) behind the C-ABI of include/liodom_hip.h.

This package is a thin ctypes binding of liodom_amd/lib/libliodom_hip.so (hand-written HIP
kernels for gfx950, liodom_amd/csrc/).  There is no CPU fallback: importing works anywhere, but
`load()` raises if the library has not been built and `Liodom(...)` raises without a GPU.
"""
from . import api  # noqa: F401
from .api import (Config, device_count, device_pci_bus_id, KernelStat, Liodom, Map, MapConfig, LiodomError, LmTrace, Params, StepInfo, build, lib_path,  # noqa: F401
                  load, make_config, make_params)
