"""ctypes binding of libgolemhip.so (include/golemflavor_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded, `lib()` raises
`GolemHipUnavailable`; if it loads but no gfx950 device is present, `gf_model_create` returns
GF_ERR_NO_DEVICE and `check()` raises `GolemHipError`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOLEMHIP_LIB") or os.path.join(HERE, "libgolemhip.so")   # override: kernel A/B experiments

GF_ABI_VERSION = 5
GF_MAX_DIM = 16
GF_MAX_BINS = 64
GF_COMM_ID_BYTES = 128
GF_IPC_HANDLE_BYTES = 64

GF_OK, GF_ERR_INVALID_ARG, GF_ERR_NO_DEVICE, GF_ERR_HIP, GF_ERR_ALLOC, GF_ERR_COMM, GF_ERR_UNSUPPORTED, GF_ERR_QUEUE_OVERFLOW = range(8)
GF_ST_OK, GF_ST_OUT_OF_PRIOR, GF_ST_NON_UNITARY, GF_ST_NAN = range(4)
GF_MODE_PRIOR_ONLY, GF_MODE_SM_GAUSS, GF_MODE_BSM_GAUSS = range(3)
GF_LAYOUT_AOS, GF_LAYOUT_SOA = range(2)


class GolemHipUnavailable(RuntimeError):
    """libgolemhip.so could not be loaded (not built, or ROCm runtime missing)."""


class GolemHipError(RuntimeError):
    """A C-ABI call returned a non-zero gf_error."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class GfModelDesc(C.Structure):
    """struct gf_model_desc, field for field."""
    _fields_ = [
        ("abi_version", C.c_int32), ("ndim", C.c_int32), ("mode", C.c_int32), ("texture", C.c_int32),
        ("dimension", C.c_int32), ("nbins", C.c_int32),
        ("idx_sm", C.c_int32 * 4), ("idx_mass", C.c_int32 * 2), ("idx_src", C.c_int32 * 2),
        ("idx_scale", C.c_int32), ("idx_mm", C.c_int32 * 4), ("idx_gamma", C.c_int32),
        ("idx_src_x", C.c_int32), ("reserved0", C.c_int32),
        ("prior_kind", C.c_int32 * GF_MAX_DIM),
        ("lo", C.c_double * GF_MAX_DIM), ("hi", C.c_double * GF_MAX_DIM),
        ("loc", C.c_double * GF_MAX_DIM), ("sigma", C.c_double * GF_MAX_DIM),
        ("log_mass", C.c_double * GF_MAX_DIM),
        ("sm_fixed", C.c_double * 4), ("mass_fixed", C.c_double * 2), ("source_ratio", C.c_double * 3),
        ("scale_fixed", C.c_double), ("mm_fixed", C.c_double * 4), ("gamma_fixed", C.c_double),
        ("bestfit_fr", C.c_double * 3), ("smearing", C.c_double), ("offset", C.c_double),
        ("flat_llh", C.c_double), ("bin_edges", C.c_double * (GF_MAX_BINS + 1)),
    ]


_vp, _dp, _ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)

# name -> (restype, argtypes): every symbol include/golemflavor_hip.h declares
SIGNATURES = {
    "gf_abi_version": (C.c_int, []),
    "gf_strerror": (C.c_char_p, [C.c_int]),
    "gf_last_hip_error": (C.c_char_p, []),
    "gf_sizeof_model_desc": (C.c_size_t, []),
    "gf_diagnostic_overrides": (C.c_int, [C.c_char_p, C.c_size_t]),
    "gf_device_trim": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "gf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gf_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "gf_model_create": (C.c_int, [C.POINTER(GfModelDesc), C.c_int, C.POINTER(_vp)]),
    "gf_model_destroy": (None, [_vp]),
    "gf_model_ndim": (C.c_int, [_vp]),
    "gf_lnprob_batch": (C.c_int, [_vp, _dp, C.c_int64, _dp, _dp, _ip]),
    "gf_lnprob_cube_batch": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, _ip, _dp, _dp, _dp, _ip]),
    "gf_propagate_batch": (C.c_int, [_vp, _dp, C.c_int64, _dp, _ip]),
    "gf_haar_draw": (C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int64, _dp, _dp]),
    "gf_device_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "gf_device_free": (C.c_int, [_vp, _vp]),
    "gf_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gf_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gf_lnprob_batch_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, _vp]),
    "gf_propagate_batch_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp]),
    "gf_haar_draw_device": (C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int64, _vp, _vp]),
    "gf_flavor_histogram_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp]),
    "gf_flavor_histogram": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, C.POINTER(C.c_uint64)]),
    "gf_model_sync": (C.c_int, [_vp]),
    "gf_event_create": (C.c_int, [C.POINTER(_vp)]),
    "gf_event_destroy": (C.c_int, [_vp]),
    "gf_event_record": (C.c_int, [_vp, _vp]),
    "gf_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "gf_sampler_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint64, C.c_double, C.POINTER(_vp)]),
    "gf_sampler_create_multi": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_uint64, C.c_double, C.POINTER(_vp)]),
    "gf_sampler_destroy": (None, [_vp]),
    "gf_sampler_set_state": (C.c_int, [_vp, _dp]),
    "gf_sampler_run": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int]),
    "gf_sampler_sync": (C.c_int, [_vp]),
    "gf_sampler_run_to_host": (C.c_int, [_vp, C.c_int64, C.c_int, _dp, _dp, _dp]),
    "gf_sampler_reset": (C.c_int, [_vp]),
    "gf_sampler_nstored": (C.c_int64, [_vp]),
    "gf_sampler_iterations": (C.c_int64, [_vp]),
    "gf_sampler_get_state": (C.c_int, [_vp, _dp, _dp]),
    "gf_sampler_get_chain": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gf_sampler_walker_mean": (C.c_int, [_vp, _dp]),
    "gf_sampler_postprocess": (C.c_int, [_vp, _dp, _ip, C.c_int, C.POINTER(C.c_uint64)]),
    "gf_sampler_postprocess_with": (C.c_int, [_vp, C.POINTER(_vp), _dp, _ip, C.c_int, C.POINTER(C.c_uint64)]),
    "gf_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "gf_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gf_comm_destroy": (None, [_vp]),
    "gf_comm_broadcast": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "gf_comm_allgather": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gf_comm_gather": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int]),
    "gf_ipc_export": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "gf_ipc_gather": (C.c_int, [C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, _vp, _vp, C.c_size_t]),
    "gf_comm_barrier": (C.c_int, [_vp]),
    "gf_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gf_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "gf_device_release": (C.c_int, [C.c_int, _vp]),
    "gf_comm_last_error": (C.c_char_p, []),
    "gf_comm_library_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "gf_host_prepare": (C.c_int, [_vp, C.c_size_t]),
    "gf_host_prepare_n": (C.c_int, [_vp, C.c_size_t, C.c_int]),
    "gf_host_register": (C.c_int, [_vp, C.c_size_t]),
    "gf_host_unregister": (C.c_int, [_vp]),
    "gf_sampler_set_stream_ids": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "gf_sampler_get_chain_device": (C.c_int, [_vp, _vp, _vp]),
    "gf_sampler_postprocess_device": (C.c_int, [_vp, C.POINTER(_vp), _vp, _vp]),
    "gf_sampler_postprocess_rows_device": (C.c_int, [_vp, C.POINTER(_vp), _vp]),
    "gf_sampler_postprocess_rows": (C.c_int, [_vp, C.POINTER(_vp), _dp]),
}

_lib = None


def lib():
    """Load libgolemhip.so once; raise GolemHipUnavailable loudly if that is impossible."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GolemHipUnavailable(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C golemflavor_amd/csrc`). golemflavor_amd has no CPU fallback." % LIB_PATH)
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise GolemHipUnavailable("cannot load %s: %s" % (LIB_PATH, exc)) from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.gf_abi_version() != GF_ABI_VERSION:
            raise GolemHipUnavailable("libgolemhip ABI %d != binding %d" % (L.gf_abi_version(), GF_ABI_VERSION))
        if L.gf_sizeof_model_desc() != C.sizeof(GfModelDesc):
            raise GolemHipUnavailable("gf_model_desc layout mismatch: C %d bytes, ctypes %d bytes"
                                      % (L.gf_sizeof_model_desc(), C.sizeof(GfModelDesc)))
        _lib = L
    return _lib


def check(code, what=""):
    if code != GF_OK:
        L = lib()
        msg = L.gf_strerror(code).decode()
        detail = L.gf_last_hip_error().decode()
        if code == GF_ERR_COMM and not detail:
            detail = L.gf_comm_last_error().decode()
        raise GolemHipError(code, "%s failed: %s%s" % (what or "libgolemhip call", msg,
                                                     (" [" + detail + "]") if detail else ""))


def diagnostic_overrides():
    """The GF_* environment overrides honoured in this process ('' = none): the library's own list, plus the host side's A/B
    switches of the grid scans (GF_SCAN_*), which change how a scan is scheduled, never what it computes."""
    buf = C.create_string_buffer(1100)
    check(lib().gf_diagnostic_overrides(buf, 1100), "gf_diagnostic_overrides")
    host = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("GF_SCAN_"))
    return (buf.value.decode() + " " + host).strip()


def device_trim(device=0):
    """Release the device memory the library caches between uses (idle unitarity workspaces, pooled constant blocks);
    returns the bytes handed back."""
    n = C.c_size_t(0)
    check(lib().gf_device_trim(int(device), C.byref(n)), "gf_device_trim")
    return int(n.value)


def device_count():
    n = C.c_int(0)
    check(lib().gf_device_count(C.byref(n)), "gf_device_count")
    return n.value
