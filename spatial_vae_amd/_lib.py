"""ctypes binding of include/svae.h.  Loading fails loudly: there is no fallback path."""
import ctypes
import os

# torch ships its own libamdhip64.so.7; it must be in the process first so that this library's
# NEEDED entry of the same soname binds to THAT runtime (two HIP runtimes in one process do not
# share a device context: launches then fail with "no ROCm-capable device").
import torch  # noqa: F401

from .build import library_path

MAX_HIDDEN = 7
ACT = {"tanh": 0, "leakyrelu": 1, "relu": 2, "sigmoid": 3}
FLAG_RESID, FLAG_BILINEAR, FLAG_SOFTPLUS = 1, 2, 4

ABI_VERSION = 2


# every function include/svae.h declares; build() and tests/test_host_cpu.py re-derive this list from the header and check it
# against the library's exports (the header is documentation: importing the package must not need it)
EXPORTS = ("svae_abi_version", "svae_adam_step", "svae_bce_loglik", "svae_colsum", "svae_ctf_filter",
           "svae_ctf_filter_workspace_bytes", "svae_decoder_backward", "svae_decoder_forward",
           "svae_decoder_forward_bce", "svae_elbo_head_backward", "svae_elbo_head_forward", "svae_gaussian_loglik",
           "svae_gaussian_workspace_bytes", "svae_gemm_mode_get", "svae_gemm_mode_set", "svae_last_error",
           "svae_latent_backward", "svae_latent_forward", "svae_linear_backward", "svae_linear_forward",
           "svae_path_counts", "svae_path_name", "svae_profile_enable", "svae_profile_kind_name",
           "svae_profile_read", "svae_rotate_bicubic", "svae_saved_bytes", "svae_workspace_bytes")


def declared_in_header(header=None):
    """The function names include/svae.h declares (used by build() and the tests, never at import)."""
    import re
    header = header or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "svae.h")
    with open(header) as f:
        return tuple(sorted(set(re.findall(r"\b(svae_[a-z0-9_]+)\s*\(", f.read()))))


PROF_KINDS = 20


class Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "N", "H", "L", "Zd", "C", "in_dim", "act", "flags")]


class Params(ctypes.Structure):
    _fields_ = [("coord_w", ctypes.c_void_p), ("coord_b", ctypes.c_void_p), ("latent_w", ctypes.c_void_p),
                ("bilinear_w", ctypes.c_void_p), ("hidden_w", ctypes.c_void_p * MAX_HIDDEN),
                ("hidden_b", ctypes.c_void_p * MAX_HIDDEN), ("out_w", ctypes.c_void_p), ("out_b", ctypes.c_void_p)]


class Pose(ctypes.Structure):
    _fields_ = [("coords", ctypes.c_void_p), ("grid", ctypes.c_void_p), ("theta", ctypes.c_void_p),
                ("dx", ctypes.c_void_p)]


class LatentDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("inf_dim", ctypes.c_int32), ("rotate", ctypes.c_int32),
                ("translate", ctypes.c_int32), ("mu_penalty", ctypes.c_int32), ("dx_scale", ctypes.c_float),
                ("z_scale", ctypes.c_float), ("theta_prior", ctypes.c_float)]


class PoseGrads(ctypes.Structure):
    _fields_ = [("dcoords", ctypes.c_void_p), ("dtheta", ctypes.c_void_p), ("ddx", ctypes.c_void_p)]


_lib = None


def lib():
    """The loaded library.  Raises if it was never built: the product path has no other backend."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("spatial_vae_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or spatial_vae_amd.build()); there is no fallback implementation" % path)
    L = ctypes.CDLL(path)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32
    L.svae_abi_version.restype = ctypes.c_int
    L.svae_last_error.restype = ctypes.c_char_p
    L.svae_saved_bytes.restype = sz
    L.svae_saved_bytes.argtypes = [ctypes.POINTER(Desc)]
    L.svae_workspace_bytes.restype = sz
    L.svae_workspace_bytes.argtypes = [ctypes.POINTER(Desc)]
    L.svae_decoder_forward.restype = ctypes.c_int
    L.svae_decoder_forward.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Params), ctypes.POINTER(Pose), vp,
                                       vp, vp, vp, vp, sz, vp]
    L.svae_decoder_forward_bce.restype = ctypes.c_int
    L.svae_decoder_forward_bce.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Params), ctypes.POINTER(Pose), vp,
                                           vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.svae_decoder_backward.restype = ctypes.c_int
    L.svae_decoder_backward.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(Params), ctypes.POINTER(Pose), vp,
                                        vp, vp, vp, vp, ctypes.POINTER(Params), vp, ctypes.POINTER(PoseGrads),
                                        vp, sz, vp]
    L.svae_bce_loglik.restype = ctypes.c_int
    L.svae_bce_loglik.argtypes = [i32, i32, vp, vp, vp, vp, vp]
    L.svae_gaussian_workspace_bytes.restype = sz
    L.svae_gaussian_workspace_bytes.argtypes = [i32, i32]
    L.svae_gaussian_loglik.restype = ctypes.c_int
    L.svae_gaussian_loglik.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, sz, vp]
    L.svae_latent_forward.restype = ctypes.c_int
    L.svae_latent_forward.argtypes = [ctypes.POINTER(LatentDesc), vp, vp, vp, vp, vp, vp, vp]
    L.svae_latent_backward.restype = ctypes.c_int
    L.svae_latent_backward.argtypes = [ctypes.POINTER(LatentDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    L.svae_elbo_head_forward.restype = ctypes.c_int
    L.svae_elbo_head_forward.argtypes = [vp, vp, i32, vp, vp]
    L.svae_elbo_head_backward.restype = ctypes.c_int
    L.svae_elbo_head_backward.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.svae_colsum.restype = ctypes.c_int
    L.svae_colsum.argtypes = [vp, i32, i32, vp, vp]
    L.svae_linear_forward.restype = ctypes.c_int
    L.svae_linear_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.svae_linear_backward.restype = ctypes.c_int
    L.svae_linear_backward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.svae_adam_step.restype = ctypes.c_int
    L.svae_adam_step.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_int64, i32, vp]
    L.svae_rotate_bicubic.restype = ctypes.c_int
    L.svae_rotate_bicubic.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.svae_ctf_filter.restype = ctypes.c_int
    L.svae_ctf_filter_workspace_bytes.restype = sz
    L.svae_ctf_filter_workspace_bytes.argtypes = [i32, i32, i32]
    L.svae_ctf_filter.argtypes = [vp, vp, i32, i32, i32, ctypes.c_double, vp, sz, vp]
    L.svae_gemm_mode_set.restype = ctypes.c_int
    L.svae_gemm_mode_set.argtypes = [ctypes.c_int]
    L.svae_gemm_mode_get.restype = ctypes.c_int
    L.svae_profile_enable.restype = ctypes.c_int
    L.svae_profile_enable.argtypes = [ctypes.c_int]
    L.svae_profile_read.restype = ctypes.c_int
    L.svae_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    L.svae_profile_kind_name.restype = ctypes.c_char_p
    L.svae_profile_kind_name.argtypes = [ctypes.c_int]
    L.svae_path_counts.restype = ctypes.c_int
    L.svae_path_counts.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.svae_path_name.restype = ctypes.c_char_p
    L.svae_path_name.argtypes = [ctypes.c_int]
    if L.svae_abi_version() != ABI_VERSION:
        raise RuntimeError("spatial_vae_amd: %s has ABI version %d, this binding needs %d -- rebuild it"
                           % (path, L.svae_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("svae: %s (code %d)" % (lib().svae_last_error().decode(), rc))


def set_gemm_mode(name):
    """'fp32' (fp32 MFMA) or 'fp16x3' (split-operand f16 MFMA, fp32-accurate).  Call before any decoder call of the process:
    buffer sizes depend on it."""
    check(lib().svae_gemm_mode_set({"fp32": 0, "fp16x3": 1}[name]))


def gemm_mode():
    return ("fp32", "fp16x3")[lib().svae_gemm_mode_get()]


def profile_enable(level):
    """0 = off, 1 = the three MFMA GEMM kernels only (cheap enough for a timed region), 2 = every kernel."""
    check(lib().svae_profile_enable(int(level)))


def profile_read():
    """{kernel kind: (total ms, launches)} since the last read; synchronises the recorded events."""
    L = lib()
    ms = (ctypes.c_double * PROF_KINDS)()
    cnt = (ctypes.c_int64 * PROF_KINDS)()
    check(L.svae_profile_read(ms, cnt))
    return {L.svae_profile_kind_name(k).decode(): (ms[k], cnt[k]) for k in range(PROF_KINDS) if cnt[k]}


PATH_KINDS = 16


def path_counts(reset=False):
    """{kernel family: launches} dispatched by this process (svae_path_counts): tells a run of the fp16x3 split kernels, the
    rank-1 output-layer backward etc. from a fallback to the plain fp32 kernels."""
    L = lib()
    arr = (ctypes.c_int64 * PATH_KINDS)()
    check(L.svae_path_counts(arr, 1 if reset else 0))
    return {L.svae_path_name(i).decode(): int(arr[i]) for i in range(PATH_KINDS) if L.svae_path_name(i)}
