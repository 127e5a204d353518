"""ctypes front-end of oracle/d2r_oracle.c.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by dream2real_amd/.  See the header of d2r_oracle.c for what is
restated and which reference lines each function follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libd2r_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "d2r_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


class _Nerf(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32),
                ("level_scale", C.c_void_p), ("level_res", C.c_void_p),
                ("level_size", C.c_void_p), ("level_offset", C.c_void_p),
                ("grid", C.c_void_p), ("dw1", C.c_void_p), ("dw2", C.c_void_p),
                ("cw1", C.c_void_p), ("cw2", C.c_void_p), ("cw3", C.c_void_p),
                ("occ_bits", C.c_void_p), ("aabb_scale", C.c_uint32), ("render_aabb", C.c_float * 6)]


class _View(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("focal", C.c_float * 2),
                ("center", C.c_float * 2), ("scale", C.c_float), ("offset", C.c_float * 3),
                ("background", C.c_float * 4), ("min_transmittance", C.c_float),
                ("near_distance", C.c_float), ("lens_mode", C.c_uint32), ("lens_params", C.c_float * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.d2r_oracle_num_threads.restype = C.c_int
        _lib.d2r_oracle_resample_coeffs.restype = C.c_int
    return _lib


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


class OracleNerf:
    """Keeps the numpy buffers alive next to the C struct that points into them."""

    def __init__(self, model):
        lv = model.levels
        self._keep = [np.ascontiguousarray(lv.scale, np.float32),
                      np.ascontiguousarray(lv.res, np.uint32),
                      np.ascontiguousarray(lv.size, np.uint32),
                      np.ascontiguousarray(lv.offset, np.uint32)]
        for name in ("grid", "dw1", "dw2", "cw1", "cw2", "cw3"):
            self._keep.append(np.ascontiguousarray(getattr(model, name), np.float16))
        self._keep.append(np.ascontiguousarray(model.occ_bits, np.uint8))
        ra = getattr(model, "render_aabb", None)
        self.c = _Nerf(lv.n_levels, lv.n_features, *[_ptr(a) for a in self._keep], int(getattr(model, "aabb_scale", 1)),
                       (C.c_float * 6)(*([0.0] * 6 if ra is None else [float(x) for x in ra])))
        self.n_feat = lv.n_levels * lv.n_features


def _cview(view) -> _View:
    return _View(view.width, view.height, (C.c_float * 2)(*view.focal),
                 (C.c_float * 2)(*view.center), view.scale, (C.c_float * 3)(*view.offset),
                 (C.c_float * 4)(*view.background), view.min_transmittance, view.near_distance,
                 int(getattr(view, "lens_mode", 0)), (C.c_float * 4)(*getattr(view, "lens_params", (0.0,) * 4)))


def nerf_matrix_to_ngp(m34, scale, offset) -> np.ndarray:
    m = np.ascontiguousarray(m34, np.float32).reshape(12)
    out = np.zeros(12, np.float32)
    off = np.ascontiguousarray(offset, np.float32)
    lib().d2r_oracle_nerf_matrix_to_ngp(C.c_void_p(_ptr(m)), C.c_float(scale),
                                        C.c_void_p(_ptr(off)), C.c_void_p(_ptr(out)))
    return out.reshape(3, 4)


def render(model: OracleNerf, view, cam_nerf):
    """Testbed.render(w,h,1,True) in Shade and Depth mode -> (rgba [H,W,4], depth [H,W], n_samples)."""
    cam = np.ascontiguousarray(np.asarray(cam_nerf, np.float64)[:3, :4], np.float32).reshape(12)
    H, W = view.height, view.width
    rgba = np.zeros((H, W, 4), np.float32)
    depth = np.zeros((H, W), np.float32)
    n = C.c_uint64(0)
    v = _cview(view)
    lib().d2r_oracle_render(C.byref(model.c), C.byref(v), C.c_void_p(_ptr(cam)),
                            C.c_void_p(_ptr(rgba)), C.c_void_p(_ptr(depth)), C.byref(n))
    return rgba, depth, int(n.value)


def set_arith(mode: int) -> int:
    """0: the specification (fp16 parameters, fp32 arithmetic).  1 / 2: emulation of tiny-cuda-nn's half arithmetic (half corner
    accumulation in the grid — older / newer form —, half accumulators and activations in the MLPs); see d2r_oracle.c.  Returns the
    previous mode."""
    lib().d2r_oracle_get_arith.restype = C.c_int
    old = int(lib().d2r_oracle_get_arith())
    lib().d2r_oracle_set_arith(C.c_int(int(mode)))
    return old


def round_half(x) -> np.ndarray:
    """the oracle's float -> half -> float rounding, element-wise (test hook)"""
    lib().d2r_oracle_round_half.restype = C.c_float
    a = np.asarray(x, np.float32)
    return np.array([lib().d2r_oracle_round_half(C.c_float(float(v))) for v in a.reshape(-1)], np.float32).reshape(a.shape)


def lens_distort(params, uv) -> np.ndarray:
    """[n,2] pinhole directions (u, v, 1) -> where the OpenCV lens (k1, k2, p1, p2) puts them on the sensor."""
    p = np.ascontiguousarray(params, np.float32)
    a = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    out = np.zeros_like(a)
    lib().d2r_oracle_lens_distort(C.c_void_p(_ptr(p)), C.c_void_p(_ptr(a)), C.c_uint32(a.shape[0]), C.c_void_p(_ptr(out)))
    return out


def lens_undistort(params, uv) -> np.ndarray:
    """instant-ngp's iterative undistortion of [n,2] sensor positions (the step pixel_to_ray runs on every ray)."""
    p = np.ascontiguousarray(params, np.float32)
    a = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    out = np.zeros_like(a)
    lib().d2r_oracle_lens_undistort(C.c_void_p(_ptr(p)), C.c_void_p(_ptr(a)), C.c_uint32(a.shape[0]), C.c_void_p(_ptr(out)))
    return out


def cone_lattice(t0: float, n: int) -> np.ndarray:
    """sample distances t_0..t_{n-1} of an aabb_scale-2 ray whose lattice starts at t0"""
    out = np.zeros(n, np.float32)
    lib().d2r_oracle_cone_lattice(C.c_float(t0), C.c_uint32(n), C.c_void_p(_ptr(out)))
    return out


def composite(fg_rgba, fg_depth, bg_rgba, bg_depth) -> np.ndarray:
    """reference reconstruction/combined_rendering.py:133-155 -> uint8 [H,W,3]."""
    H, W = fg_depth.shape
    a = [np.ascontiguousarray(x, np.float32) for x in (fg_rgba, fg_depth, bg_rgba, bg_depth)]
    out = np.zeros((H, W, 3), np.uint8)
    lib().d2r_oracle_composite(*[C.c_void_p(_ptr(x)) for x in a], C.c_uint32(W), C.c_uint32(H),
                               C.c_void_p(_ptr(out)))
    return out


def clip_preprocess(frame_u8, S: int, rot90: bool = True):
    """rot90 (clip_scoring.py:145) + HF CLIPImageProcessor (clip_scoring.py:177)
    -> (pixel_values [3,S,S] f32, cropped uint8 [S,S,3])."""
    f = np.ascontiguousarray(frame_u8, np.uint8)
    H, W, _ = f.shape
    pv = np.zeros((3, S, S), np.float32)
    u8 = np.zeros((S, S, 3), np.uint8)
    lib().d2r_oracle_clip_preprocess(C.c_void_p(_ptr(f)), C.c_uint32(W), C.c_uint32(H),
                                     C.c_uint32(S), C.c_int(int(rot90)), C.c_void_p(_ptr(pv)),
                                     C.c_void_p(_ptr(u8)))
    return pv, u8


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's fixed-point bicubic coefficient table: (bounds [out,2] int32, kk [out,ksize] int32)."""
    cap = out_size * 64
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros(cap, np.int32)
    ks = lib().d2r_oracle_resample_coeffs(C.c_int(in_size), C.c_int(out_size),
                                          C.c_void_p(_ptr(bounds)), C.c_void_p(_ptr(kk)), C.c_int(cap))
    return bounds, kk[:out_size * ks].reshape(out_size, ks).copy()


def encode_points(model: OracleNerf, xyz) -> np.ndarray:
    p = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros((p.shape[0], model.n_feat), np.float32)
    lib().d2r_oracle_encode_points(C.byref(model.c), C.c_void_p(_ptr(p)), C.c_uint32(p.shape[0]),
                                   C.c_void_p(_ptr(out)))
    return out


def eval_points(model: OracleNerf, xyz, dirs) -> np.ndarray:
    """[n,4] = (sigma, r, g, b) with rgb the network's (sRGB-space) prediction."""
    p = np.ascontiguousarray(xyz, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    out = np.zeros((p.shape[0], 4), np.float32)
    lib().d2r_oracle_eval_points(C.byref(model.c), C.c_void_p(_ptr(p)), C.c_void_p(_ptr(d)),
                                 C.c_uint32(p.shape[0]), C.c_void_p(_ptr(out)))
    return out


def num_threads() -> int:
    return int(lib().d2r_oracle_num_threads())
