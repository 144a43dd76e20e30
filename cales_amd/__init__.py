"""MI355X-native CaLES hot path: ctypes binding (capi), operator mirror (hotpath), namelist reader (nml), y-slab layer (decomp)."""
import os as _os

# several processes with GPU buffers (RCCL over xGMI): the host driver supports dmabuf IPC only; must be in the environment before the HIP
# runtime starts, so it is set when the package is imported (no effect if the caller already chose a value)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
