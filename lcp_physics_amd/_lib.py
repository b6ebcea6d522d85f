"""ctypes binding of liblcp_hip.so (the C ABI in include/lcp_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, the
caller gets an exception.  PyTorch is only used for device memory and streams.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LCP_HIP_LIB", os.path.join(_HERE, "csrc", "liblcp_hip.so"))   # override: A/B builds

COMPUTE_F32 = 0
COMPUTE_F64 = 1
HINT_ALL_CONTACT = 0x100
PATH_GENERIC = 0x200
IO_F64 = 0x400
PATH_CONTACT_SPACE = 0x2000
PATH_PRIMAL = 0x4000
PATH_QUAD = 0x8000
PATH_SOLO = 0x10000
HINT_PINNED = 0x20000
BWD_ADJOINT = 0x40000

ST_SINGULAR_Q = 1
ST_SINGULAR_S11 = 2
ST_SINGULAR_T = 4
ST_NAN = 8
ST_TRUNCATED = 16

_ERRORS = {-1: "LCP_E_BADARG", -2: "LCP_E_TOOLARGE", -3: "LCP_E_LAUNCH"}

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int

# symbol -> (restype, argtypes); must list every function include/lcp_hip.h declares
SIGNATURES = {
    "lcp_version": (_c.c_char_p, []),
    "lcp_workspace_bytes": (_c.c_size_t, [_I, _I, _I, _I, _I]),
    "lcp_pdipm_forward_f32": (_I, [_I] * 4 + [_P] * 7 + [_c.c_double, _I, _I, _I] + [_P] * 4 + [_P, _P, _P, _P]),
    "lcp_pdipm_forward_f64": (_I, [_I] * 4 + [_P] * 7 + [_c.c_double, _I, _I] + [_P] * 4 + [_P, _P, _P, _P]),
    "lcp_pdipm_backward_f32": (_I, [_I] * 4 + [_P] * 3 + [_I] + [_P] * 7 + [_P, _P]),
    "lcp_pdipm_backward_f64": (_I, [_I] * 4 + [_P] * 3 + [_P] * 7 + [_P, _P]),
    "lcp_assemble_contacts_f32": (_I, [_I] * 4 + [_P] * 11 + [_c.c_float] + [_P] * 7 + [_P]),
    "lcp_step_fused_f32": (_I, [_I] * 4 + [_P] * 12 + [_c.c_float, _c.c_double, _I, _I, _I] + [_P] * 5
                           + [_P, _P, _P, _P]),
    "lcp_step_backward_f32": (_I, [_I] * 4 + [_P] * 11 + [_c.c_float, _P, _I] + [_P] * 8 + [_P, _P]),
    "lcp_step_backward_je_f32": (_I, [_I] * 4 + [_P] * 11 + [_c.c_float, _P, _I] + [_P] * 9 + [_P, _P]),
    "lcp_step_has_backward": (_I, [_I, _I, _I, _I]),
    "lcp_post_stabilization_has_backward": (_I, [_I, _I, _I, _I]),
    "lcp_post_stabilization_backward_f32": (_I, [_I] * 4 + [_P] * 10 + [_I] + [_P] * 7 + [_P, _P]),
    "lcp_solve_dynamics_f32": (_I, [_I] * 4 + [_P] * 12 + [_c.c_float, _c.c_double, _I, _I, _I] + [_P] * 4
                               + [_P, _P, _P, _P]),
    "lcp_post_stabilization_f32": (_I, [_I] * 4 + [_P] * 10 + [_c.c_double, _I, _I, _I, _P, _P, _c.c_double, _P]
                                   + [_P, _P, _P, _P, _P]),
    "lcp_move_find_contacts_f64": (_I, [_I] * 3 + [_P] * 7 + [_c.c_double, _c.c_double, _I, _I, _c.c_double,
                                                          _c.c_double] + [_P] * 12 + [_P]),
    "lcp_joint_jacobian_f64": (_I, [_I] * 4 + [_P] * 8 + [_c.c_double, _c.c_double, _P, _P]),
    "lcp_contact_frame_backward_f64": (_I, [_I] * 3 + [_P] * 6 + [_c.c_double] + [_P] * 5 + [_P]),
    "lcp_joint_jacobian_backward_f64": (_I, [_I] * 4 + [_P] * 6 + [_P, _P, _P]),
    "lcp_state_update_backward_f64": (_I, [_I] * 3 + [_P] * 5 + [_c.c_double] + [_P] * 3 + [_P]),
    "lcp_debug_set_trace": (None, [_P]),
    "lcp_debug_set_path": (None, [_I]),
    "lcp_set_backward_adjoint": (None, [_I]),
}

_lib = None


def load():
    """Load the HIP library or raise - never degrade silently."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liblcp_hip.so not found at %s - build it with `make -C lcp_physics_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _ERRORS.get(rc, "unknown"), rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu_tensor(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("%s must live on a GPU (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


_PATH_BITS = {"auto": 0, "wave64": 0, "generic": PATH_GENERIC, "big": PATH_CONTACT_SPACE, "primal": PATH_PRIMAL,
              "quad": PATH_QUAD, "solo": PATH_SOLO}     # quad / solo: four scenes / one scene per wavefront at every batch size
_tls = threading.local()          # the default is per host thread, like the library's own debugging aids


def set_path(path):
    """A/B aid: 'auto' | 'generic' | 'big' (contact-space kernels where the default is a body-space one) | 'primal' (one wave per
    scene at every size) - the calling thread's default.  Every forward wrapper ORs `path_bits()` into the `compute` word of its
    call and RECORDS the word in the handle it returns (`LCPSolution.compute`, `out["compute"]`); the backward wrappers pass the
    recorded word on, so a backward picks its forward's kernel family on whatever thread autograd runs it (PyTorch runs the
    backward of a CUDA op on a device worker thread, where no default was ever set)."""
    if path not in _PATH_BITS:
        raise ValueError("unknown kernel path %r" % (path,))
    _tls.path = path


class thread_path:
    """The fp64-I/O entry points carry no `compute` word: for the duration of ONE call the calling thread's library default
    (lcp_debug_set_path) is set from the path bits recorded with the op, and reset afterwards."""
    _CODE = {0: 0, PATH_GENERIC: 1, PATH_CONTACT_SPACE: 3, PATH_PRIMAL: 4}          # (quad / solo do not apply to the dense fp64 entries)

    def __init__(self, word):
        self.code = self._CODE[word & (PATH_GENERIC | PATH_CONTACT_SPACE | PATH_PRIMAL)]

    def __enter__(self):
        if self.code:
            load().lcp_debug_set_path(self.code)

    def __exit__(self, *a):
        if self.code:
            load().lcp_debug_set_path(0)


def path_bits(path="auto"):
    """LCP_PATH_* bits of a per-call `path` argument ('auto' = the module default set by `set_path`)."""
    return _PATH_BITS[getattr(_tls, "path", "auto") if path == "auto" else path]


def workspace_bytes(B, nz, m, e, compute):
    return int(load().lcp_workspace_bytes(B, nz, m, e, compute))
