"""spectavi_amd -- MI355X (gfx950) implementation of Spectavi's descriptor-matching
and DLT-triangulation hot path behind the reference's ctypes surface.

    from spectavi_amd import feature, mvg          # host numpy arrays in/out
    from spectavi_amd import device                # torch tensors resident in HBM
    from spectavi_amd import sharded               # one process per GPU, RCCL gather

Importing the package loads spectavi_amd/libspectavi.so (HIP); there is no CPU
fallback (see spectavi_amd/_lib.py).
"""
from spectavi_amd._lib import (SpectaviError, device_count, set_device, set_devices, set_gather_mode, set_hash_seed)  # noqa: F401

__version__ = "0.1.0"
