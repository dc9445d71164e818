"""ctypes binding of libasp_hip.so (the C ABI declared in include/asp.h).

There is no CPU fallback anywhere in this package: if the shared object is
missing and cannot be built, or no GPU is visible when a compute entry point is
called, an :class:`AspError` is raised.
"""
from __future__ import annotations

import atexit
import ctypes
import os
import sys
import weakref
from typing import Optional

import numpy as np

from . import build as _build

c_void_p = ctypes.c_void_p
c_u64 = ctypes.c_uint64
c_u32 = ctypes.c_uint32
c_i32 = ctypes.c_int32
c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double


class AspError(RuntimeError):
    """A libasp_hip call failed; ``code`` is the asp_status."""

    def __init__(self, code: int, message: str):
        super().__init__("libasp_hip error %d: %s" % (code, message))
        self.code = code


class SaInfo(ctypes.Structure):
    """Mirror of ``asp_sa_info`` (include/asp.h)."""

    _fields_ = [
        ("num_spins", c_u64),
        ("nnz_offdiag", c_u64),
        ("ell_entries", c_u64),
        ("num_colors", c_u32),
        ("num_blocks", c_u32),
        ("max_degree", c_u32),
        ("energy_scale_exp", c_i32),
        ("diag_sum", c_double),
        ("beta0_auto", c_double),
        ("beta1_auto", c_double),
    ]


SA_BATCH_SHUFFLED = 1  # asp.h: ASP_SA_BATCH_SHUFFLED


class SaBatchItem(ctypes.Structure):
    """Mirror of ``asp_sa_batch_item`` (include/asp.h)."""

    _fields_ = [
        ("plan", c_void_p),
        ("seed", c_u64),
        ("betas", c_void_p),
        ("num_sweeps", c_u32),
        ("repetitions", c_u32),
        ("replica_offset", c_u32),
        ("flags", c_u32),
        ("out_x", c_void_p),
        ("out_e", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/asp.h declares
SIGNATURES = {
    "asp_last_error": (ctypes.c_char_p, []),
    "asp_last_error_code": (c_int, []),
    "asp_clear_error": (None, []),
    "asp_device_count": (c_int, []),
    "asp_device_touched": (c_int, []),
    "asp_set_device": (c_int, [c_int]),
    "asp_get_device": (c_int, []),
    "asp_version": (ctypes.c_char_p, []),
    "asp_shutdown": (c_int, []),
    "build_matrix": (c_u64, [c_u64] + [c_void_p] * 11),
    "extract_signs": (None, [c_u64, c_void_p, c_void_p]),
    "asp_build_create": (c_void_p, [c_u64, c_u64]),
    "asp_build_upload": (c_int, [c_void_p] * 8),
    "asp_build_run": (c_int, [c_void_p, ctypes.POINTER(c_u64)]),
    "asp_build_last_ms": (c_float, [c_void_p]),
    "asp_build_download": (c_int, [c_void_p] * 5),
    "asp_build_destroy": (None, [c_void_p]),
    "asp_ising_elements": (c_int, [c_u64, c_void_p, c_void_p, c_u64] + [c_void_p] * 7),
    "asp_ising_elements_last_ms": (c_float, []),
    "asp_operator_create": (c_int, [ctypes.c_uint32, ctypes.c_uint32, c_void_p, c_void_p, c_void_p,
                                    ctypes.POINTER(c_void_p)]),
    "asp_operator_destroy": (None, [c_void_p]),
    "asp_operator_set_symmetry": (c_int, [c_void_p, c_u32, c_void_p, c_i32]),
    "asp_operator_state_info": (c_int, [c_void_p, c_u64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "asp_plain_basis_create": (c_int, [c_void_p, c_i32, ctypes.POINTER(c_void_p)]),
    "asp_plain_basis_destroy": (None, [c_void_p]),
    "asp_plain_basis_dimension": (c_u64, [c_void_p]),
    "asp_plain_basis_states": (c_int, [c_void_p, c_void_p]),
    "asp_plain_matvec": (c_int, [c_void_p, c_void_p, c_void_p]),
    "asp_table_create": (c_int, [c_u64, c_void_p, ctypes.POINTER(c_void_p)]),
    "asp_table_destroy": (None, [c_void_p]),
    "asp_table_index": (c_int, [c_void_p, c_u64, c_void_p, c_void_p]),
    "asp_sector_enumerate": (c_int, [c_void_p, c_i32, c_u64, c_void_p, c_void_p, ctypes.POINTER(c_u64)]),
    "asp_sector_width": (ctypes.c_uint32, [c_void_p]),
    "asp_sector_rows": (c_int, [c_void_p, c_u64, c_void_p, c_void_p, c_u32, c_void_p, c_void_p, c_void_p]),
    "asp_sector_matvec": (c_int, [c_u64, c_u32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "asp_operator_unique_targets": (c_int, [c_void_p]),
    "asp_operator_max_connections": (ctypes.c_uint32, [c_void_p]),
    "asp_operator_apply": (c_int, [c_void_p, c_u64, c_void_p, c_u64, c_void_p, c_void_p, c_void_p,
                                   ctypes.POINTER(c_u64)]),
    "asp_operator_ising": (c_int, [c_void_p, c_u64, c_void_p, c_void_p, c_u64, c_void_p, c_void_p,
                                   c_void_p, ctypes.POINTER(c_u64)]),
    "asp_operator_ising_csr": (c_int, [c_void_p, c_u64, c_void_p, c_void_p, c_u64, c_void_p, c_void_p,
                                       c_void_p, ctypes.POINTER(c_u64)]),
    "asp_operator_extend": (c_int, [c_void_p, c_u64, c_void_p, c_u64, c_void_p,
                                    ctypes.POINTER(c_u64)]),
    "asp_operator_last_ms": (c_float, []),
    "asp_sparsify_component": (c_int, [c_u64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       ctypes.c_double, c_u64, c_void_p, ctypes.POINTER(c_u64),
                                       c_u64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_u64)]),
    "asp_sparsify_last_ms": (c_float, []),
    "asp_sa_plan_create": (c_void_p, [c_u64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "asp_sa_plan_destroy": (None, [c_void_p]),
    "asp_sa_plan_info": (c_int, [c_void_p, ctypes.POINTER(SaInfo)]),
    "asp_sa_layout_host": (c_int, [c_u64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   ctypes.POINTER(SaInfo), c_void_p, c_void_p]),
    "asp_sa_shuffled_order_host": (c_int, [c_u64, c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_u32,
                                           c_void_p, c_void_p, ctypes.POINTER(c_u32)]),
    "asp_sa_set_launch": (c_int, [c_void_p, c_int, c_int]),
    "asp_sa_set_packed": (c_int, [c_void_p, c_int]),
    "asp_sa_set_wide": (c_int, [c_void_p, c_int]),
    "asp_sa_set_team": (c_int, [c_void_p, c_int]),
    "asp_sa_last_layout": (c_int, [c_void_p]),
    "asp_sa_set_field_cache": (c_int, [c_void_p, c_int]),
    "asp_sa_anneal": (c_int, [c_void_p, c_u64, c_void_p, c_u32, c_u32, c_u32, c_void_p, c_void_p,
                              c_void_p]),
    "asp_sa_anneal_trace": (c_int, [c_void_p, c_u64, c_void_p, c_u32, c_u32, c_u32, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "asp_sa_anneal_shuffled": (c_int, [c_void_p, c_u64, c_void_p, c_u32, c_u32, c_u32, c_void_p, c_void_p,
                                       c_void_p]),
    "asp_sa_set_shuffled_launch": (c_int, [c_void_p, c_int, c_int]),
    "asp_sa_set_shuffled_teams": (c_int, [c_void_p, c_int]),
    "asp_sa_last_shuffled": (c_int, [c_void_p, ctypes.POINTER(c_u32), ctypes.POINTER(c_float)]),
    "asp_sa_team_watchdog_trips": (c_int, [c_void_p, ctypes.POINTER(c_u32), ctypes.POINTER(c_u64)]),
    "asp_sa_last_shuffled_blocks": (c_int, [c_void_p, ctypes.POINTER(c_u32), ctypes.POINTER(c_u32)]),
    "asp_sa_last_shuffled_fill": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "asp_sa_anneal_batch": (c_int, [ctypes.POINTER(SaBatchItem), c_u32]),
    "asp_sa_batch_last_ms": (c_float, []),
    "asp_sa_greedy": (c_int, [c_void_p, c_u32, c_void_p, c_void_p, ctypes.POINTER(c_u32)]),
    "asp_sa_greedy_tree_host": (c_int, [c_u64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "asp_sa_last_sweep_ms": (c_float, [c_void_p]),
    "asp_sa_last_total_ms": (c_float, [c_void_p]),
    "asp_sa_last_stats": (c_int, [c_void_p, c_u32, c_void_p, c_void_p]),
    "asp_sa_last_launch": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int)]),
    "asp_sa_energy": (c_int, [c_void_p, c_u32, c_void_p, c_void_p]),
}

_lib: Optional[ctypes.CDLL] = None
_closed = False  # shutdown() ran: device handles are gone, destructors must not touch HIP
_live = weakref.WeakSet()  # objects owning device handles; each has release()


def track(owner) -> None:
    """Register an object that owns a libasp_hip handle: :func:`shutdown` calls its
    ``release()`` before the interpreter starts tearing modules down."""
    _live.add(owner)


def closed() -> bool:
    return _closed or sys.is_finalizing()


def shutdown() -> None:
    """Destroy every live handle, then ``asp_shutdown``.  Registered with ``atexit`` (which runs
    before module globals are cleared and long before the C runtime's exit handlers), so that no
    ``__del__`` reaches into HIP during interpreter teardown — where the HIP runtime or a
    profiler attached to it may already be finalised — and nothing of this library is left for
    the runtime's own exit handler to clean up."""
    global _closed
    if _lib is None or _closed:
        return
    for owner in list(_live):
        try:
            owner.release()
        except Exception:
            pass
    _closed = True
    try:
        _lib.asp_shutdown()
    except Exception:
        pass


def library_path() -> str:
    return _build.LIB_PATH


def _preload_torch_hip() -> None:
    """One HIP runtime per process.  PyTorch ships its own ``libamdhip64.so``; libasp_hip.so is
    linked against the system one (same SONAME).  Whichever is loaded first serves both — and
    torch cannot initialise its device on the system runtime (``torch.cuda.is_available()`` turns
    False) when this library came first.  So torch's copy is mapped before libasp_hip.so whenever
    torch is installed, without importing torch: the order of imports then does not matter, and
    device pointers of torch tensors (sector_ed.py, distributed.py) belong to the runtime this
    library runs on."""
    import importlib.util

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> ctypes.CDLL:
    """Load (building first if the sources changed and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    # Eight hardware queues instead of the runtime's default of four (read when HIP initialises the
    # device, so only a process whose GPU is still untouched can ask): a batched shuffled anneal runs
    # up to four classes of sweep kernels beside two streams of order kernels, and streams that
    # share a hardware queue block one another while one of them waits for an event (the sampled-
    # cluster pipeline in the shuffled order: 14.8 -> 12.1 s per round of 32 clusters).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _preload_torch_hip()
    path = _build.LIB_PATH
    try:
        path = _build.build()
    except Exception as exc:  # no hipcc / read-only tree: use the shipped .so if any
        if not os.path.exists(path):
            raise AspError(-2, "libasp_hip.so is missing and could not be built: %s" % exc)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        if _build._TAG and not hasattr(lib, name):
            continue  # development A/B against an older tagged build (tools/ab_tags.sh)
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    atexit.register(shutdown)
    return lib


_wait_policy_applied = False


def _apply_wait_policy() -> None:
    """``$ASP_HIP_WAIT`` = ``spin`` | ``yield`` | ``block``: how a host thread waits for the device
    (hipSetDeviceFlags, once, when the process first asks for the GPU — never earlier: it
    initialises the HIP runtime, and a process that has done so must not fork).  HIP's default spins
    on a core: the right thing for one thread, but sixteen pipeline threads waiting that way take
    the cores away from the host stages of the other clusters."""
    global _wait_policy_applied
    if _wait_policy_applied:
        return
    _wait_policy_applied = True
    mode = os.environ.get("ASP_HIP_WAIT", "").strip().lower()
    flag = {"spin": 1, "yield": 2, "block": 4}.get(mode)
    if flag is None:
        return
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipSetDeviceFlags(ctypes.c_uint(flag))  # (an error — device in use already — is not fatal)
    except OSError:
        pass


def last_error() -> str:
    return load().asp_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise AspError(rc, last_error())


def check_recorded() -> None:
    """For the reference-signature symbols, which cannot return a status."""
    lib = load()
    code = lib.asp_last_error_code()
    if code != 0:
        raise AspError(code, last_error())


def device_count() -> int:
    n = load().asp_device_count()
    if n < 0:
        raise AspError(n, last_error())
    return n


def gpu_touched() -> bool:
    """True once this process has made a HIP call through the library."""
    return bool(load().asp_device_touched())


def require_gpu() -> None:
    if device_count() <= 0:
        raise AspError(-1, "no HIP device visible; this package has no CPU fallback")
    _apply_wait_policy()


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_void_p)
