"""Loader of the C-ABI HIP library (include/ed3dgs.h).  There is no CPU fallback: if the library is missing the
import of any product module fails loudly."""
import ctypes as C
import os
import subprocess

_CSRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc"))
# ED3DGS_LIB_PATH: another build of the same library (A/B experiments of compile-time options: tools/ab_build.sh)
LIB_PATH = os.environ.get("ED3DGS_LIB_PATH") or os.path.join(_CSRC, "libed3dgs_hip.so")
_lib = None

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class DeformCfg(C.Structure):
    """ed3dgs_deform_cfg of include/ed3dgs.h (field order must match)."""
    _fields_ = [("P", C.c_int), ("W", C.c_int), ("D", C.c_int), ("E", C.c_int), ("TD", C.c_int), ("n_sh", C.c_int),
                ("max_embeddings", C.c_int), ("num_offsets", C.c_int), ("use_stage", C.c_int * 2),
                ("n_rows", C.c_int * 2), ("no_ds", C.c_int), ("no_dr", C.c_int), ("no_do", C.c_int),
                ("no_dc", C.c_int), ("coef", C.c_float), ("coef_c", C.c_float), ("coef_o", C.c_float),
                ("coef_s", C.c_float), ("time", C.c_float), ("cam_no", C.c_int)]


class StateView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "rec", "rec_coord", "depths", "cov3D", "clamped", "tiles_touched", "point_offsets", "point_list_keys",
        "point_list", "ranges", "n_contrib", "accum_coord", "accum_depth", "normal_length", "depth_order")]


def build(verbose=False):
    """Compile every HIP source for gfx950 into libed3dgs_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"ed3dgs_amd: HIP library {LIB_PATH} is missing -- run __graft_entry__.build() (or `make -C {_CSRC}`). "
            "There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    L.ed3dgs_last_error.restype = C.c_char_p
    L.ed3dgs_abi_version.restype = C.c_int
    for n in ("ed3dgs_geometry_bytes", "ed3dgs_image_bytes", "ed3dgs_binning_bytes", "ed3dgs_backward_workspace_bytes",
              "ed3dgs_deform_param_count", "ed3dgs_deform_workspace_bytes", "ed3dgs_filter3d_workspace_bytes",
              "ed3dgs_knn_workspace_bytes", "ed3dgs_integrate_point_bytes", "ed3dgs_integrate_workspace_bytes"):
        getattr(L, n).restype = C.c_size_t
    for n in ("ed3dgs_rasterize_forward", "ed3dgs_rasterize_backward", "ed3dgs_mark_visible", "ed3dgs_state_view_get",
              "ed3dgs_deform_forward", "ed3dgs_deform_backward", "ed3dgs_deform_forward_activated", "ed3dgs_deform_backward_activated", "ed3dgs_profile_begin", "ed3dgs_profile_end", "ed3dgs_profile_begin_slots", "ed3dgs_profile_end_slots", "ed3dgs_activations_forward",
              "ed3dgs_activations_backward", "ed3dgs_compute_3d_filter", "ed3dgs_knn_mean_dist2", "ed3dgs_knn_neighbours", "ed3dgs_integrate", "ed3dgs_image_stats", "ed3dgs_profile_tile_backward_counts",
              "ed3dgs_set_option", "ed3dgs_get_option", "ed3dgs_binning_path", "ed3dgs_profile_tile_counts",
              "ed3dgs_measure_mfma_ceiling"):
        getattr(L, n).restype = C.c_int
    L.ed3dgs_set_option.argtypes = [C.c_char_p, C.c_int]
    L.ed3dgs_get_option.argtypes = [C.c_char_p]
    _lib = L
    return L


EXPORTS = (
    "ed3dgs_last_error", "ed3dgs_abi_version", "ed3dgs_geometry_bytes", "ed3dgs_image_bytes", "ed3dgs_binning_bytes",
    "ed3dgs_backward_workspace_bytes", "ed3dgs_rasterize_forward", "ed3dgs_rasterize_backward", "ed3dgs_mark_visible",
    "ed3dgs_state_view_get", "ed3dgs_deform_param_count", "ed3dgs_deform_workspace_bytes", "ed3dgs_deform_forward",
    "ed3dgs_deform_backward", "ed3dgs_deform_forward_activated", "ed3dgs_deform_backward_activated", "ed3dgs_profile_begin", "ed3dgs_profile_end", "ed3dgs_profile_begin_slots", "ed3dgs_profile_end_slots", "ed3dgs_activations_forward",
    "ed3dgs_activations_backward", "ed3dgs_filter3d_workspace_bytes", "ed3dgs_compute_3d_filter",
    "ed3dgs_knn_workspace_bytes", "ed3dgs_knn_mean_dist2", "ed3dgs_knn_neighbours",
    "ed3dgs_integrate_point_bytes", "ed3dgs_integrate_workspace_bytes", "ed3dgs_integrate", "ed3dgs_image_stats", "ed3dgs_profile_tile_backward_counts",
    "ed3dgs_set_option", "ed3dgs_get_option", "ed3dgs_binning_path", "ed3dgs_profile_tile_counts", "ed3dgs_measure_mfma_ceiling")


def set_option(name, value):
    """Set a process-wide switch of the library (include/ed3dgs.h: ed3dgs_set_option); returns the previous value."""
    old = lib().ed3dgs_set_option(name.encode(), int(value))
    if old < 0 and last_error().startswith("ed3dgs_set_option"):
        raise KeyError(last_error())
    return old


def get_option(name):
    return lib().ed3dgs_get_option(name.encode())


class options:
    """Context manager: switches set on entry, restored on exit (tests, bench A/B passes)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def last_error():
    return lib().ed3dgs_last_error().decode()


def raw_stream(device):
    """The current HIP stream of `device` as a ctypes pointer.  torch.cuda.current_stream() builds a Stream object through four
    Python layers (17 us per call, six calls per training step); the raw handle is one C call."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:   # a torch without the private accessor
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
