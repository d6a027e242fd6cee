"""Loader for libspectavi.so (gfx950 build).

Counterpart of reference spectavi/__libspectavi.py:1-9: the shared library sits
next to the package and is opened with ctypes.  There is no CPU fallback: a
missing library raises ImportError here, and a missing GPU surfaces as
SpectaviError from the first call.
"""
import ctypes as ct
import os

_PKG_DIR = os.path.dirname(os.path.realpath(__file__))
lib_path = os.path.join(_PKG_DIR, "libspectavi.so")

if not os.path.exists(lib_path):
    raise ImportError(
        "libspectavi.so (HIP/gfx950) is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C spectavi_amd/csrc`.  spectavi_amd has no CPU fallback.")



def _preload_torch_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (+ libhsa-runtime64) under torch/lib; libspectavi.so is linked against the system ROCm.
    If libspectavi.so initialises the system runtime first and torch is imported later, torch
    brings up a second runtime and reports "No HIP GPUs are available".  When torch is
    installed but not yet imported, load ITS runtime first so that both sides share it (the
    loader resolves libspectavi's libamdhip64.so.7 dependency to the copy already in the
    process).  SPECTAVI_NO_TORCH_PRELOAD=1 disables this."""
    import sys
    if "torch" in sys.modules or os.environ.get("SPECTAVI_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ct.CDLL(path, mode=ct.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the system runtime


_preload_torch_hip_runtime()
clib = ct.cdll.LoadLibrary(lib_path)

SPV_OK, SPV_ERR_INVALID, SPV_ERR_HIP, SPV_ERR_NOMEM, SPV_ERR_INTERNAL = 0, 1, 2, 3, 4

clib.spv_last_status.restype = ct.c_int
clib.spv_last_status.argtypes = []
clib.spv_last_error.restype = ct.c_char_p
clib.spv_last_error.argtypes = []
clib.spv_version.restype = ct.c_char_p
clib.spv_version.argtypes = []
clib.spv_device_count.restype = ct.c_int
clib.spv_device_count.argtypes = []
clib.spv_set_device.restype = ct.c_int
clib.spv_set_device.argtypes = [ct.c_int]
clib.spv_set_hash_seed.restype = None
clib.spv_set_hash_seed.argtypes = [ct.c_uint32, ct.c_int]


class SpectaviError(RuntimeError):
    """Raised when a libspectavi entry point reports a non-zero status."""

    def __init__(self, status, message):
        super().__init__("libspectavi status %d: %s" % (status, message))
        self.status = status


def check(status=None):
    """Raise SpectaviError if `status` (or the thread's last status) is non-zero."""
    if status is None:
        status = clib.spv_last_status()
    if status != SPV_OK:
        raise SpectaviError(status, (clib.spv_last_error() or b"").decode("utf-8", "replace"))


def device_count():
    return int(clib.spv_device_count())


def set_device(device):
    check(clib.spv_set_device(int(device)))


def set_devices(devices):
    """Shard the host-array entry points (feature.*, mvg.*) over these GPUs of the node."""
    devices = [int(d) for d in devices]
    arr = (ct.c_int * len(devices))(*devices)
    clib.spv_set_devices.restype = ct.c_int
    clib.spv_set_devices.argtypes = [ct.POINTER(ct.c_int), ct.c_int]
    check(clib.spv_set_devices(arr, len(devices)))


def set_hash_seed(seed=None):
    """Fix the hyperplane seed of nn_cascading_hash (None: back to std::random_device)."""
    if seed is None:
        clib.spv_set_hash_seed(0, 0)
    else:
        clib.spv_set_hash_seed(int(seed) & 0xFFFFFFFF, 1)


GATHER_AUTO, GATHER_DIRECT, GATHER_RCCL, GATHER_PEERCOPY = -1, 0, 1, 2


def set_gather_mode(mode):
    """How multi-device host-array calls collect their shards: "rccl" (ncclGather of 16-byte records
    on the first listed GPU, inside libspectavi.so), "copy" (the same records and root layout moved by peer copies, no RCCL), "direct" (each shard copied to its slice), or
    "auto" (SPECTAVI_GATHER if set, else RCCL when more than one distinct device is listed)."""
    mode = {"auto": GATHER_AUTO, "direct": GATHER_DIRECT, "rccl": GATHER_RCCL, "copy": GATHER_PEERCOPY}.get(mode, mode)
    clib.spv_set_gather_mode.restype = ct.c_int
    clib.spv_set_gather_mode.argtypes = [ct.c_int]
    check(clib.spv_set_gather_mode(int(mode)))
