"""ctypes binding of libd2r.so (include/d2r.h).  Fails loudly when the library is missing or
there is no gfx950 device: this package has no CPU compute path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd2r.so")
ABI_VERSION = 9          # D2R_ABI_VERSION of include/d2r.h this binding was written against

EXPORTS = [
    "d2r_abi_version", "d2r_ctx_create", "d2r_ctx_destroy", "d2r_ctx_set_stream", "d2r_ctx_synchronize",
    "d2r_last_error", "d2r_nerf_create", "d2r_nerf_destroy", "d2r_render", "d2r_nerf_eval_points",
    "d2r_set_background", "d2r_render_composite", "d2r_clip_create", "d2r_clip_destroy",
    "d2r_clip_score_frames", "d2r_clip_preprocess", "d2r_clip_embed_pixels", "d2r_render_score",
    "d2r_get_render_stats", "d2r_collect_render_stats", "d2r_ctx_set_option", "d2r_get_timing", "d2r_text_create",
    "d2r_text_destroy", "d2r_text_encode", "d2r_comm_get_unique_id", "d2r_comm_init", "d2r_comm_destroy",
    "d2r_allgather_scores", "d2r_phys_create", "d2r_phys_destroy", "d2r_phys_check", "d2r_nerf_load_ingp", "d2r_lens_undistort_view",
    "d2r_rectify_background_depth", "d2r_ingp_inspect", "d2r_render_score_host", "d2r_png_write", "d2r_png_write_batch",
    "d2r_png_read_batch", "d2r_png_size", "d2r_savetxt", "d2r_ingp_validate", "d2r_debug_gemm_fp8", "d2r_ctx_get_option", "d2r_png_write_batch_bg",
]


class D2RError(RuntimeError):
    pass


class NerfDesc(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("level_scale", C.c_void_p),
                ("level_res", C.c_void_p), ("level_size", C.c_void_p), ("level_offset", C.c_void_p),
                ("n_entries", C.c_uint32), ("grid_fp16", C.c_void_p), ("dw1_fp16", C.c_void_p),
                ("dw2_fp16", C.c_void_p), ("cw1_fp16", C.c_void_p), ("cw2_fp16", C.c_void_p),
                ("cw3_fp16", C.c_void_p), ("occupancy_bits", C.c_void_p), ("aabb_scale", C.c_uint32),
                ("render_aabb", C.c_float * 6)]


class ViewC(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("focal", C.c_float * 2),
                ("center", C.c_float * 2), ("scale", C.c_float), ("offset", C.c_float * 3),
                ("background", C.c_float * 4), ("min_transmittance", C.c_float),
                ("near_distance", C.c_float), ("lens_mode", C.c_uint32), ("lens_params", C.c_float * 4)]


class ClipDesc(C.Structure):
    _fields_ = [("image_size", C.c_uint32), ("patch_size", C.c_uint32), ("hidden_size", C.c_uint32),
                ("num_layers", C.c_uint32), ("num_heads", C.c_uint32), ("mlp_size", C.c_uint32),
                ("proj_dim", C.c_uint32)]


class TextDesc(C.Structure):
    _fields_ = [("vocab_size", C.c_uint32), ("context_length", C.c_uint32), ("hidden_size", C.c_uint32),
                ("num_layers", C.c_uint32), ("num_heads", C.c_uint32), ("mlp_size", C.c_uint32),
                ("proj_dim", C.c_uint32)]


class PhysParams(C.Structure):
    _fields_ = [("sample_res", C.c_uint32 * 6), ("init_pose", C.c_float * 16), ("table_z", C.c_float),
                ("unsup_thresh", C.c_float), ("gravity", C.c_float * 3), ("perturb", C.c_float),
                ("stability_check", C.c_int32), ("disallow_regrasp", C.c_int32), ("margin", C.c_float)]


class IngpView(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("w", C.c_uint32), ("h", C.c_uint32),
                ("lens_mode", C.c_uint32), ("lens_params", C.c_float * 4)]


class IngpInfo(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("aabb_scale", C.c_uint32),
                ("has_background", C.c_int32), ("dataset_scale", C.c_double), ("dataset_offset", C.c_double * 3),
                ("background_color", C.c_float * 4), ("n_views", C.c_uint32), ("n_views_written", C.c_uint32),
                ("n_unknown_keys", C.c_uint32), ("render_with_lens_distortion", C.c_int32)]


class FrameSink(C.Structure):
    _fields_ = [("png_dir", C.c_char_p), ("png_first_index", C.c_uint32), ("png_threads", C.c_int32), ("png_level", C.c_int32)]


class RenderStats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("rays_alive", C.c_uint64), ("samples", C.c_uint64),
                ("wave_iters", C.c_uint64), ("l0_tokens", C.c_uint64), ("l0_touched", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("march_ms", C.c_double), ("march_launches", C.c_uint64), ("raygen_ms", C.c_double),
                ("raygen_launches", C.c_uint64), ("prep_ms", C.c_double), ("prep_launches", C.c_uint64),
                ("clip_ms", C.c_double), ("clip_launches", C.c_uint64), ("sort_ms", C.c_double), ("sort_launches", C.c_uint64),
                ("vit_qkv_ms", C.c_double), ("vit_qkv_launches", C.c_uint64), ("vit_attn_ms", C.c_double), ("vit_attn_launches", C.c_uint64),
                ("vit_out_ms", C.c_double), ("vit_out_launches", C.c_uint64), ("vit_fc1_ms", C.c_double), ("vit_fc1_launches", C.c_uint64),
                ("vit_fc2_ms", C.c_double), ("vit_fc2_launches", C.c_uint64)]


COMM_ID_BYTES = 128      # D2R_COMM_ID_BYTES

_lib = None


def load() -> C.CDLL:
    """dlopen libd2r.so and declare prototypes.  Raises if the library has not been built
    (`python -c "import __graft_entry__ as g; g.build()"`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D2RError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                       "(make -C dream2real_amd/csrc); there is no CPU fallback")
    # torch wheels bundle their own libamdhip64.so.7; whichever copy of that soname is mapped first
    # serves the whole process.  Load torch's first so libd2r, torch tensors/streams and RCCL all
    # share ONE HIP runtime (with /opt/rocm's copy mapped first, torch finds no GPU afterwards).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.d2r_last_error.restype = C.c_char_p
    lib.d2r_last_error.argtypes = [C.c_void_p]
    lib.d2r_ctx_destroy.restype = None
    lib.d2r_nerf_destroy.restype = None
    lib.d2r_clip_destroy.restype = None
    lib.d2r_text_destroy.restype = None
    lib.d2r_phys_destroy.restype = None
    for name in EXPORTS:
        getattr(lib, name)          # every declared symbol must be exported
    if lib.d2r_abi_version() != ABI_VERSION:
        raise D2RError("libd2r.so ABI version mismatch")
    _lib = lib
    return lib


def ingp_inspect(data: bytes) -> str:
    """d2r_ingp_inspect: the msgpack tree of a snapshot as text, each leaf marked as read / ignored by the loader,
    plus the sizes the loader derives (host only: works without a GPU)."""
    lib = load()
    lib.d2r_ingp_inspect.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    need = C.c_size_t(0)
    check(lib.d2r_ingp_inspect(data, len(data), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    check(lib.d2r_ingp_inspect(data, len(data), buf, need.value, None))
    return buf.value.decode("utf-8", "replace")


def png_write(rgb, path: str, level: int = -1):
    """d2r_png_write: one uint8 [h,w,3] image -> an RGB PNG file (host only)."""
    a = np.ascontiguousarray(rgb, np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    check(load().d2r_png_write(ptr(a), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[0]), os.fsencode(path), C.c_int(level)))


def png_write_batch(frames, out_dir: str, first_index: int = 0, threads: int = 0, level: int = -1):
    """d2r_png_write_batch: uint8 [n,h,w,3] -> <out_dir>/cb_rgb_%04d.png on a pool of host threads (the GIL is released
    for the duration of the call)."""
    a = np.ascontiguousarray(frames, np.uint8)
    assert a.ndim == 4 and a.shape[3] == 3
    check(load().d2r_png_write_batch(ptr(a), C.c_uint32(a.shape[0]), C.c_uint32(a.shape[2]), C.c_uint32(a.shape[1]),
                                     os.fsencode(out_dir), C.c_uint32(first_index), C.c_int(threads), C.c_int(level)))


def png_write_batch_bg(frames, background, out_dir: str, first_index: int = 0, threads: int = 0):
    """d2r_png_write_batch_bg: like png_write_batch for frames that equal `background` [h,w,3] in most scanlines (those are entropy-coded
    once); same pixels in the files."""
    a = np.ascontiguousarray(frames, np.uint8)
    b = np.ascontiguousarray(background, np.uint8)
    assert a.ndim == 4 and a.shape[3] == 3 and b.shape == a.shape[1:]
    check(load().d2r_png_write_batch_bg(ptr(a), C.c_uint32(a.shape[0]), C.c_uint32(a.shape[2]), C.c_uint32(a.shape[1]), ptr(b),
                                        os.fsencode(out_dir), C.c_uint32(first_index), C.c_int(threads)))


def png_size(path: str):
    w, h = C.c_uint32(0), C.c_uint32(0)
    check(load().d2r_png_size(os.fsencode(path), C.byref(w), C.byref(h)))
    return int(w.value), int(h.value)


def png_read_batch(in_dir: str, n: int = 0, first_index: int = 0, threads: int = 0, indices=None, size=None) -> np.ndarray:
    """d2r_png_read_batch: <in_dir>/cb_rgb_%04d.png for n consecutive indices from first_index (or for the given
    `indices`) -> uint8 [n,h,w,3].  `size` = (w, h), default the first file's; a file of another size is an error."""
    idx = None if indices is None else np.ascontiguousarray(indices, np.uint32)
    if idx is not None:
        n = idx.shape[0]
    if n == 0:
        return np.empty((0, 0, 0, 3), np.uint8)
    w, h = size or png_size(os.path.join(in_dir, f"cb_rgb_{int(idx[0]) if idx is not None else first_index:04d}.png"))
    out = np.empty((n, h, w, 3), np.uint8)
    check(load().d2r_png_read_batch(os.fsencode(in_dir), ptr(idx), C.c_uint32(first_index), C.c_uint32(n), C.c_uint32(w),
                                    C.c_uint32(h), ptr(out), C.c_int(threads)))
    return out


def savetxt(path: str, array):
    """d2r_savetxt: np.savetxt(path, array) with numpy's defaults, byte for byte, formatted on the library's worker
    threads (1-D: one number per line; 2-D: one row per line; 0-D is refused like np.savetxt does)."""
    a = np.asarray(array)
    if a.ndim == 0 or a.ndim > 2:
        raise ValueError(f"Expected 1D or 2D array, got {a.ndim}D array instead")
    a = np.ascontiguousarray(a, np.float64)
    rows, cols = (a.shape[0], 1) if a.ndim == 1 else a.shape
    check(load().d2r_savetxt(os.fsencode(path), ptr(a), C.c_uint64(rows), C.c_uint64(cols), C.c_int(0)))


def ingp_validate(data: bytes) -> IngpInfo:
    """d2r_ingp_validate: raises D2RError with the loader's message (naming the key) when d2r_nerf_load_ingp would refuse
    the snapshot; returns what the loader would report about it otherwise.  Host only."""
    info = IngpInfo()
    lib = load()
    lib.d2r_ingp_validate.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    check(lib.d2r_ingp_validate(data, len(data), C.byref(info)))
    return info


def check(rc: int, ctx=None):
    if rc != 0:
        msg = load().d2r_last_error(ctx)
        raise D2RError(f"libd2r error {rc}: {msg.decode() if msg else '?'}")


def ptr(a) -> C.c_void_p:
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def view_c(v) -> ViewC:
    return ViewC(v.width, v.height, (C.c_float * 2)(*v.focal), (C.c_float * 2)(*v.center), v.scale,
                 (C.c_float * 3)(*v.offset), (C.c_float * 4)(*v.background), v.min_transmittance,
                 v.near_distance, int(v.lens_mode), (C.c_float * 4)(*[float(x) for x in v.lens_params]))
