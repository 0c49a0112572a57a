"""ctypes binding of libd3m_raster.so (include/d3m_raster.h).  There is no fallback: if the HIP
library is missing or a call fails, this module raises."""
import ctypes
import os

import torch

from .build import LOAD_PATH as LIB_PATH

_c_f32p = ctypes.c_void_p      # device pointers travel as integers
_I, _F, _SZ, _L = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long
_P = ctypes.c_void_p


class D3MCamera(ctypes.Structure):
    _fields_ = [("mode", _I), ("perspective", _I), ("tan_half_width", _F), ("orig_size", _F),
                ("rot", _P), ("eye_or_t", _P), ("K", _P), ("dist", _P),
                ("rot_batch", _I), ("eye_batch", _I), ("K_batch", _I), ("dist_batch", _I)]


class D3MBasis(ctypes.Structure):
    _fields_ = [("eye", _P), ("at_or_direction", _P), ("up", _P), ("eye_batch", _I), ("at_batch", _I), ("up_batch", _I),
                ("is_look_at", _I)]


class D3MVertexTarget(ctypes.Structure):
    _fields_ = [("grad_vertices", _P), ("tri", _P), ("num_vertices", _I), ("num_tri", _I), ("tri_batch", _I),
                ("fill_back", _I)]


class D3MFitTargets(ctypes.Structure):
    _fields_ = [("rgb_target", _P), ("depth_target", _P), ("alpha_target", _P), ("mask", _P), ("scratch", _P),
                ("loss", _P), ("grad_rgb_map", _P), ("grad_alpha_map", _P), ("grad_depth_map", _P), ("grad_loss", _P),
                ("mask_sum", _P), ("edge_grad", _P), ("edge_dot", _P), ("edge_nz_lo_inv", _P), ("edge_nz_hi1", _P), ("flags", _I)]


class D3MG2SBlock(ctypes.Structure):
    """d3m_g2s_block (include/d3m_raster.h, section D), field for field."""
    _fields_ = ([("batch_size", _I), ("height", _I), ("width", _I), ("image_size", _I), ("anti_aliasing", _I), ("flip", _I),
                 ("inv_K", _P), ("inv_K_batch", _I), ("K", _P), ("K_batch", _I),
                 ("rot_center_depth", _F), ("depth_min", _F), ("depth_max", _F), ("near", _F), ("far", _F),
                 ("camera", ctypes.POINTER(D3MCamera)), ("view", _P), ("view_components", _I)] +
                [(n, _P) for n in ("rot", "trans", "depth", "albedo", "light_a", "light_b", "light_d", "target", "extra_mask",
                                   "normal", "diffuse_shading", "texture", "recon_depth", "recon_im", "recon_im_mask",
                                   "losses")] +
                [("lam_smooth", _F), ("with_smooth", _I)] +
                [(n, _P) for n in ("screen_vertices", "zbuffer", "scratch",
                                   "grad_recon_im", "grad_l1", "grad_l1_flip", "grad_smooth", "grad_total",
                                   "grad_texture", "grad_tri", "grad_depth_map", "grad_normal", "grad_depth_mesh",
                                   "grad_depth", "grad_albedo", "grad_light_a", "grad_light_b", "grad_light_d", "grad_rot",
                                   "grad_trans", "grad_view")])


CAMERA_NONE, CAMERA_LOOK_AT, CAMERA_LOOK, CAMERA_PROJECTION = 0, 1, 2, 3
PRECLEARED = 1            # D3M_PRECLEARED: the caller has zeroed what the operator would clear (include/d3m_raster.h, "Clears")
FIT_FINISH_DEFERRED = 2   # D3M_FIT_FINISH_DEFERRED: the fused objective's finish is left to d3m_backward_textures_lit
FIT_POOLED = 4            # D3M_FIT_POOLED: records of an objective on the 2x2-pooled images (anti-aliasing)
GRAD_OF_OUTPUT_IMAGE = 32   # D3M_GRAD_OF_OUTPUT_IMAGE (d3m_backward_depth_map_mesh)
FRONT_RANGES = 10         # clears d3m_lit_front takes

_SIGNATURES = {
    "d3m_version": (ctypes.c_char_p, []),
    "d3m_last_hip_error": (_I, []),
    "d3m_zero_ranges": (_I, [ctypes.POINTER(_P), ctypes.POINTER(_SZ), _I, _P]),
    "d3m_error_string": (ctypes.c_char_p, [_I]),
    "d3m_timing_enable": (None, [_I]),
    "d3m_timing_collect": (_I, [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_I), ctypes.POINTER(_F), _I]),
    "d3m_forward_workspace_bytes": (_SZ, [_I, _I, _I]),
    "d3m_forward_workspace_min_bytes": (_SZ, [_I, _I, _I]),
    "d3m_set_coverage_form": (_I, [_I]),
    "d3m_get_coverage_form": (_I, []),
    "d3m_set_deterministic": (_I, [_I]),
    "d3m_get_deterministic": (_I, []),
    "d3m_vertex_gather": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "d3m_face_light_backward_gather": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_forward_coverage_form": (_I, [_I, _I, _I]),
    "d3m_forward_big_batch": (_I, [_I, _I, _I]),
    "d3m_forward_face_index_map": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _I, _P, _SZ, _P]),
    "d3m_forward_face_index_map_mesh": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _SZ, _P, _SZ, _I, _P]),
    "d3m_forward_face_index_map_mesh_modes": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _SZ, _P, _SZ,
                                                   _P, _P, _P, _I, _P]),
    "d3m_forward_clear_bytes": (_SZ, [_I, _I, _I, _I, _SZ]),
    "d3m_forward_texture_sampling": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "d3m_backward_pixel_map_workspace_bytes": (_SZ, [_I, _I, _I]),
    "d3m_backward_pixel_map_workspace_min_bytes": (_SZ, [_I, _I, _I]),
    "d3m_backward_depth_map_mesh": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "d3m_backward_pixel_map": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _I, _P, _SZ, _P, _P, _P, _SZ, _P, _P]),
    "d3m_edge_plan_bytes": (_SZ, [_I, _I, _I]),
    "d3m_edge_plan_min_bytes": (_SZ, [_I, _I, _I]),
    "d3m_edge_plan_extents_offset": (_SZ, [_I, _I, _I, ctypes.POINTER(_SZ)]),
    "d3m_edge_plan": (_I, [_P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "d3m_edge_plan_clear_bytes": (_SZ, [_I, _I, _I]),
    "d3m_visibility_bytes": (_SZ, [_I, _I]),
    "d3m_visibility": (_I, [_P, _P, _SZ, _I, _I, _I, _P]),
    "d3m_backward_faces_workspace_bytes": (_SZ, [_I, _I]),
    "d3m_backward_textures": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "d3m_backward_depth_map": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "d3m_camera_basis": (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _I, _P]),
    "d3m_camera_forward": (_I, [_P, _I, ctypes.POINTER(D3MCamera), _P, _I, _I, _P]),
    "d3m_camera_backward": (_I, [_P, _I, ctypes.POINTER(D3MCamera), _P, _P, _I, _I, _P]),
    "d3m_camera_backward_add": (_I, [_P, _I, ctypes.POINTER(D3MCamera), _P, _P, _I, _I, _P]),
    "d3m_gather_faces": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "d3m_scatter_face_grads": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "d3m_lighting_forward": (_I, [_P, _P, _P, _F, _F, _P, _P, _P, _L, _I, _P]),
    "d3m_lighting_backward": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _L, _I, _P]),
    "d3m_view_transform": (_I, [_P, _I, _P, _P, _I, _P]),
    "d3m_view_transform_backward": (_I, [_P, _I, _P, _P, _P, _I, _P]),
    "d3m_grid_warp": (_I, [_P, _P, _I, _P, _P, _F, _P, _I, _P, _P, _I, _I, _I, _P]),
    "d3m_grid_warp_backward": (_I, [_P, _P, _I, _P, _P, _F, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_depth_normals": (_I, [_P, _P, _I, _P, _I, _I, _I, _P]),
    "d3m_depth_normals_backward": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "d3m_textures_from_im": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "d3m_textures_from_im_backward": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "d3m_uv_unwrap": (_I, [_P] * 11 + [_I] * 8 + [_P]),
    "d3m_uv_unwrap_backward": (_I, [_P] * 12 + [_I] * 8 + [_P]),
    "d3m_face_light": (_I, [_P, _I, _P, _I, _P, _F, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "d3m_face_light_backward": (_I, [_P, _I, _P, _I, _P, _P, _F, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "d3m_forward_texture_sampling_lit": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "d3m_backward_textures_lit_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "d3m_render_lit_epilogue": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I,
                                     _P, _P]),
    "d3m_render_fit_scratch_floats": (_SZ, [_I, _I]),
    "d3m_render_fit_scratch_clear_range": (_SZ, [_I, _I, ctypes.POINTER(_SZ)]),
    "d3m_fit_finish": (_I, [ctypes.POINTER(D3MFitTargets), _I, _I, _P]),
    "d3m_fit_loss_records": (_I, [_P, _P, _P, _P, ctypes.POINTER(D3MFitTargets), _I, _I, _P]),
    "d3m_backward_textures_lit": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _SZ,
                                       _P, _P, _P, _I, _P]),
    "d3m_backward_textures_lit_clear_ranges": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, ctypes.POINTER(_P),
                                                    ctypes.POINTER(_SZ)]),
    "d3m_lit_back": (_I, [_P, _I, ctypes.POINTER(D3MCamera), _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _F, _F, _P, _P, _P, _P]),
    "d3m_lit_front": (_I, [_P, _I, ctypes.POINTER(D3MCamera), ctypes.POINTER(D3MBasis), _P, _I, _I, _P, _I, _I, _I, _P, _I,
                           _F, _F, _P, _P, _P, ctypes.POINTER(_P), ctypes.POINTER(_SZ), _I, _P]),
    "d3m_output_epilogue": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_output_epilogue_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_output_epilogue_backward_records": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_photometric_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "d3m_sum_squared_error": (_I, [_P, _P, _P, _P, _P, _L, _P]),
    "d3m_smooth_loss_forward": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "d3m_smooth_loss_backward": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "d3m_fit_loss_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_fit_loss_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "d3m_warp_resample": (_I, [_P, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "d3m_warp_resample_partials": (_I, [_I, _I]),
    "d3m_warp_resample_backward": (_I, [_P, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "d3m_g2s_scratch_floats": (_SZ, [_I, _I, _I, _I]),
    "d3m_g2s_forward": (_I, [ctypes.POINTER(D3MG2SBlock), _P]),
    "d3m_g2s_backward": (_I, [ctypes.POINTER(D3MG2SBlock), _P]),
    "d3m_mesh_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "d3m_mesh_render_colors": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "d3m_mesh_render_texture": (_I, [_P] * 8 + [_I] * 10 + [_P, _SZ, _P]),
    "d3m_mesh_map_texture": (_I, [_P] * 6 + [_I] * 8 + [_P]),
    "d3m_mesh_vis_of_vertices": (_I, [_P] * 6 + [_I] * 4 + [_P, _SZ, _P]),
    "d3m_mesh_get_triangle_buffer": (_I, [_P] * 5 + [_I] * 4 + [_P, _SZ, _P]),
    "d3m_mesh_get_norm_direction": (_I, [_P, _P, _P, _I, _I, _P, _SZ, _P]),
    "d3m_mesh_triangle_mean": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "d3m_mesh_triangle_normals": (_I, [_P, _P, _P, _I, _I, _P]),
    "d3m_mesh_normalize": (_I, [_P, _I, _P]),
    "d3m_mesh_get_correspondence": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "d3m_load_textures": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "d3m_create_texture_image": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
}

_lib = None


def exported_symbols():
    """Every entry point include/d3m_raster.h declares."""
    return sorted(_SIGNATURES)


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m deep3dmap_amd.build`). "
                "deep3dmap_amd has no CPU / eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)     # AttributeError if the build is stale
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        L = lib()
        raise RuntimeError(f"{what} failed: {L.d3m_error_string(rc).decode()} (code {rc}, hip error "
                           f"{L.d3m_last_hip_error()})")


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  (Through the raw accessor: torch.cuda.current_stream()
    builds a Stream object per call, ~9 us each and a dozen calls per eager render step.)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def ptr(t):
    """Device pointer of a tensor, or NULL for None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def require_device(*tensors, names=None):
    """The reference's CHECK_INPUT (rasterize_cuda.cpp:66-68): CUDA + contiguous, else RuntimeError.
    Dtypes: the reference's launchers dispatch on faces' type (rasterize_cuda_kernel.cu:614, float and double) and then
    read every map with .data<scalar_t>() / .data<int32_t>(), which raises for a tensor of another type; its Python side
    only ever allocates float32 / int32 maps (rasterize.py:50-69), so scalar_t = double cannot get past the first map.
    This library instantiates scalar_t = float only and raises the same kind of error up front."""
    for i, t in enumerate(tensors):
        if t is None:
            continue
        n = names[i] if names else f"argument {i}"
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{n} must be contiguous")
        if t.is_floating_point() and t.dtype != torch.float32:
            raise RuntimeError(f"{n}: expected scalar type Float but found {t.dtype}; "
                               "libd3m_raster is built for float32 (the reference's float64 dispatch, "
                               "rasterize_cuda_kernel.cu:614, is not reachable from its own Python either)")


COVERAGE_FORMS = {"auto": -1, "binned": 0, "bidding": 1}


class coverage_form:
    """Context manager: run the forward's coverage in one form ("auto" | "binned" | "bidding", d3m_set_coverage_form)."""

    def __init__(self, form):
        self.form = COVERAGE_FORMS[form] if isinstance(form, str) else int(form)

    def __enter__(self):
        self.previous = lib().d3m_get_coverage_form()
        check(lib().d3m_set_coverage_form(self.form), "d3m_set_coverage_form")
        return self

    def __exit__(self, *exc):
        lib().d3m_set_coverage_form(self.previous)
        return False


class deterministic:
    """Context manager: the visibility list in ascending face order (d3m_set_deterministic) inside the block."""

    def __init__(self, on=True):
        self.on = int(bool(on))

    def __enter__(self):
        self.previous = lib().d3m_get_deterministic()
        check(lib().d3m_set_deterministic(self.on), "d3m_set_deterministic")
        return self

    def __exit__(self, *exc):
        lib().d3m_set_deterministic(self.previous)
        return False


ZERO_RANGES_MAX = 6       # FILL_RANGES of csrc/d3m_launch.h


def zero_(*tensors):
    """Zero-fill contiguous device tensors, up to six per launch (d3m_zero_ranges); returns them.  The kernel takes raw
    (pointer, byte count) ranges of whole 4-byte words: a tensor that is not contiguous (its data_ptr + numel would
    cover its neighbours) or whose size is not a multiple of four bytes is cleared by torch instead."""
    raw = []
    for t in tensors:
        if t is None or t.numel() == 0:
            continue
        if t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0 and t.data_ptr() % 4 == 0:
            raw.append(t)
        else:
            t.zero_()
    for at in range(0, len(raw), ZERO_RANGES_MAX):
        ts = raw[at:at + ZERO_RANGES_MAX]
        ptrs = (_P * len(ts))(*[t.data_ptr() for t in ts])
        sizes = (_SZ * len(ts))(*[t.numel() * t.element_size() for t in ts])
        check(lib().d3m_zero_ranges(ptrs, sizes, len(ts), stream_ptr()), "d3m_zero_ranges")
    return tensors


def zero_raw(ranges):
    """Zero (device pointer, bytes) ranges -- 4-byte multiples -- with as few launches as d3m_zero_ranges allows."""
    ranges = [(int(p), int(n)) for p, n in ranges if p and n]
    for at in range(0, len(ranges), ZERO_RANGES_MAX):
        part = ranges[at:at + ZERO_RANGES_MAX]
        ptrs = (_P * len(part))(*[p for p, _ in part])
        sizes = (_SZ * len(part))(*[n for _, n in part])
        check(lib().d3m_zero_ranges(ptrs, sizes, len(part), stream_ptr()), "d3m_zero_ranges")


def tensor_range(t):
    """(device pointer, bytes) of a contiguous tensor whose size is a multiple of four bytes"""
    assert t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0
    return t.data_ptr(), t.numel() * t.element_size()


def kernel_timing(enable):
    lib().d3m_timing_enable(int(bool(enable)))


def collect_kernel_times(max_entries=64):
    """{kernel name: (launches, total ms)} since timing was enabled / last collected (synchronises)."""
    names = (ctypes.c_char_p * max_entries)()
    counts = (_I * max_entries)()
    ms = (_F * max_entries)()
    n = lib().d3m_timing_collect(names, counts, ms, max_entries)
    return {names[i].decode(): (counts[i], ms[i]) for i in range(n)}
