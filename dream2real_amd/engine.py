"""Python handles over libd2r.so: Context (one per GPU), Testbed (the pyngp.Testbed surface the
path uses) and ClipScorer (the CLIPModel/CLIPProcessor surface the path uses).

Reference surfaces mirrored here:
  pyngp.Testbed  — reconstruction/combined_rendering.py:98-105,112-113,116,123-130 and
                   reconstruction/ngp_visual_model.py:24-28
  CLIPModel / CLIPProcessor — clip_scoring.py:150-151,177-181
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os

import numpy as np

from . import _lib
from .clip_model import pack_text_weights, pack_vision_weights
from .scene import LENS_OPENCV, LENS_PERSPECTIVE, NerfModel, View

Shade = "Shade"     # pyngp.Shade
Depth = "Depth"     # pyngp.Depth


class Context:
    """d2r_ctx: one per GPU; owns the HIP stream and workspaces."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.d2r_ctx_create(C.c_int(device), C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.d2r_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        _lib.check(rc, self.h)

    def set_stream(self, hip_stream_ptr: int):
        self.check(self.lib.d2r_ctx_set_stream(self.h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        self.check(self.lib.d2r_ctx_synchronize(self.h))

    def set_option(self, key: str, value: int):
        self.check(self.lib.d2r_ctx_set_option(self.h, key.encode(), C.c_int64(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        self.check(self.lib.d2r_ctx_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def render_stats(self, collect_K: int | None = None) -> dict:
        if collect_K is not None:
            self.check(self.lib.d2r_collect_render_stats(self.h, C.c_uint32(collect_K)))
        s = _lib.RenderStats()
        self.check(self.lib.d2r_get_render_stats(self.h, C.byref(s)))
        return {"rays_total": s.rays_total, "rays_alive": s.rays_alive, "samples": s.samples,
                "wave_iters": s.wave_iters, "l0_tokens": s.l0_tokens, "l0_touched": s.l0_touched}

    def timing(self) -> dict:
        """Device time per kernel family since set_option("timing", 1) (synchronises)."""
        t = _lib.Timing()
        self.check(self.lib.d2r_get_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    # --- pose-shard communicator (RCCL behind the C ABI) ------------------------
    def comm_unique_id(self) -> bytes:
        """Rank 0: the bootstrap blob every rank hands to comm_init (ncclGetUniqueId)."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(self.lib.d2r_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, id_blob: bytes | None, rank: int, world: int):
        if id_blob is not None:
            assert len(id_blob) == _lib.COMM_ID_BYTES
        self.check(self.lib.d2r_comm_init(self.h, id_blob, C.c_int(rank), C.c_int(world)))

    def comm_destroy(self):
        self.check(self.lib.d2r_comm_destroy(self.h))

    def allgather_scores(self, local_dev_ptr: int, n_local: int, global_dev_ptr: int):
        """d2r_allgather_scores: [n_local] fp32 on every rank -> [world][n_local], async on the stream."""
        self.check(self.lib.d2r_allgather_scores(self.h, C.c_void_p(local_dev_ptr), C.c_size_t(n_local),
                                                 C.c_void_p(global_dev_ptr)))

    def set_background(self, view: View, bg_rgba: np.ndarray, bg_depth: np.ndarray):
        a = np.ascontiguousarray(bg_rgba, np.float32)
        d = np.ascontiguousarray(bg_depth, np.float32)
        assert a.shape == (view.height, view.width, 4) and d.shape == (view.height, view.width)
        v = _lib.view_c(view)
        self.check(self.lib.d2r_set_background(self.h, C.byref(v), _lib.ptr(a), _lib.ptr(d)))

    def lens_undistort_view(self, view: View) -> np.ndarray:
        """[h, w, 2]: the undistorted camera-space direction (x, y; z = 1) of every pixel centre of a view with a lens
        (parity hook, d2r_lens_undistort_view)."""
        out = np.empty((view.height, view.width, 2), np.float32)
        v = _lib.view_c(view)
        self.check(self.lib.d2r_lens_undistort_view(self.h, C.byref(v), _lib.ptr(out)))
        return out

    def rectify_background_depth(self, depth, mask, width: int, height: int, return_mask: bool = False):
        """Sensor depth [sh, sw] (fp16 or fp32 metres) and movable mask [sh, sw] (bool / uint8, or None) of a render
        view -> background depth [height, width] float32: centre crop, cv2.INTER_CUBIC resize, 100 where the resized
        mask is 0 (d2r_rectify_background_depth; reference combined_rendering.py:107-110, 166-209)."""
        d = np.asarray(depth)
        d = np.ascontiguousarray(d, np.float16 if d.dtype == np.float16 else np.float32)
        sh, sw = d.shape
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
            assert m.shape == (sh, sw)
        out = np.empty((height, width), np.float32)
        mo = np.empty((height, width), np.uint8) if (return_mask and m is not None) else None
        self.check(self.lib.d2r_rectify_background_depth(
            self.h, _lib.ptr(d), C.c_int(1 if d.dtype == np.float16 else 0), _lib.ptr(m) if m is not None else None,
            C.c_uint32(sw), C.c_uint32(sh), C.c_uint32(width), C.c_uint32(height), _lib.ptr(out),
            _lib.ptr(mo) if mo is not None else None))
        return (out, mo) if return_mask else out


class _Nerf:
    """`testbed.nerf` namespace of pyngp: render_min_transmittance, and the render lens — `render_with_lens_distortion`
    (False on a fresh Testbed; reference reconstruction/train_ngp.py:70 sets it by hand, set_camera_to_training_view sets it
    for every frame the path renders) and `render_lens` (dict(mode=0 | 1, params=(k1, k2, p1, p2)))."""

    def __init__(self):
        self.render_min_transmittance = 0.01
        self.render_with_lens_distortion = False
        self.render_lens = dict(mode=LENS_PERSPECTIVE, params=(0.0, 0.0, 0.0, 0.0))


class Testbed:
    """The slice of pyngp.Testbed that Dream2Real's render path touches, backed by libd2r.

    `training_views` is the per-view metadata a snapshot carries (intrinsics at the training
    resolution, and optionally `lens=(k1, k2, p1, p2)`: the OpenCV coefficients of the transforms the
    model was trained from); `dataset_scale` / `dataset_offset` are the values nerf_matrix_to_ngp uses.
    """

    def __init__(self, ctx: Context, model: NerfModel, training_views=None, dataset_scale: float = 1.0,
                 dataset_offset=(0.0, 0.3, 0.5)):
        self.ctx = ctx
        self.model = model
        lv = model.levels
        self._keep = dict(
            scale=np.ascontiguousarray(lv.scale, np.float32), res=np.ascontiguousarray(lv.res, np.uint32),
            size=np.ascontiguousarray(lv.size, np.uint32), offset=np.ascontiguousarray(lv.offset, np.uint32),
            grid=np.ascontiguousarray(model.grid, np.float16), dw1=np.ascontiguousarray(model.dw1, np.float16),
            dw2=np.ascontiguousarray(model.dw2, np.float16), cw1=np.ascontiguousarray(model.cw1, np.float16),
            cw2=np.ascontiguousarray(model.cw2, np.float16), cw3=np.ascontiguousarray(model.cw3, np.float16),
            occ=np.ascontiguousarray(model.occ_bits, np.uint8))
        k = self._keep
        desc = _lib.NerfDesc(lv.n_levels, lv.n_features, _lib.ptr(k["scale"]), _lib.ptr(k["res"]),
                             _lib.ptr(k["size"]), _lib.ptr(k["offset"]), lv.n_entries, _lib.ptr(k["grid"]),
                             _lib.ptr(k["dw1"]), _lib.ptr(k["dw2"]), _lib.ptr(k["cw1"]), _lib.ptr(k["cw2"]),
                             _lib.ptr(k["cw3"]), _lib.ptr(k["occ"]), int(getattr(model, "aabb_scale", 1)),
                             (C.c_float * 6)(*([0.0] * 6 if getattr(model, "render_aabb", None) is None
                                               else [float(x) for x in model.render_aabb])))
        h = C.c_void_p()
        ctx.check(ctx.lib.d2r_nerf_create(ctx.h, C.byref(desc), C.byref(h)))
        self.h = h
        self._init_state(training_views, dataset_scale, dataset_offset)

    def _init_state(self, training_views, dataset_scale, dataset_offset):
        # pyngp.Testbed attributes the path reads/writes
        self.background_color = [0.0, 0.0, 0.0, 1.0]
        self.render_ground_truth = False
        self.render_mode = Shade
        self.snap_to_pixel_centers = True
        self.shall_train = False
        self.nerf = _Nerf()
        self.training_views = training_views or [dict(fx=924.66912, fy=926.49735, cx=654.51953, cy=355.18523,
                                                      w=1280, h=720)]
        self.dataset_scale = dataset_scale
        self.dataset_offset = tuple(dataset_offset)
        self._view_idx = 0
        self._cam = np.eye(4, dtype=np.float32)[:3]

    @classmethod
    def from_snapshot(cls, ctx: "Context", path: str) -> "Testbed":
        """`ngp.Testbed(ngp.TestbedMode.Nerf)` + `load_snapshot(path)` (reference
        reconstruction/ngp_visual_model.py:24-28) through the C ABI: d2r_nerf_load_ingp parses the file's bytes
        (zlib/gzip msgpack) in the library and returns the model handle plus the state a Testbed keeps beside it
        (training-view intrinsics, dataset scale/offset, saved background colour)."""
        data = open(path, "rb").read()
        buf = np.frombuffer(data, np.uint8)
        h = C.c_void_p()
        info = _lib.IngpInfo()
        cap = 4096
        views = (_lib.IngpView * cap)()
        ctx.check(ctx.lib.d2r_nerf_load_ingp(ctx.h, _lib.ptr(buf), C.c_size_t(buf.size), C.byref(h), C.byref(info), views,
                                             C.c_uint32(cap)))
        tv = [dict(fx=float(v.fx), fy=float(v.fy), cx=float(v.cx), cy=float(v.cy), w=int(v.w), h=int(v.h),
                   lens=tuple(float(x) for x in v.lens_params) if v.lens_mode == LENS_OPENCV else None)
              for v in views[: info.n_views_written]]
        tb = cls.__new__(cls)
        tb.ctx, tb.model, tb._keep, tb.h = ctx, None, {}, h
        tb._init_state(tv or None, float(info.dataset_scale), tuple(float(x) for x in info.dataset_offset))
        tb.snapshot_unknown_keys = int(info.n_unknown_keys)
        tb.nerf.render_with_lens_distortion = bool(info.render_with_lens_distortion)
        if info.n_unknown_keys:
            import warnings
            warnings.warn(f"{path}: {info.n_unknown_keys} key(s) this loader does not know (neither read, checked nor known to be "
                          "irrelevant to rendering); list them with dream2real_amd._lib.ingp_inspect(open(path, 'rb').read())")
        if info.has_background:            # a Testbed restores the colour the snapshot was saved with
            tb.background_color = [float(x) for x in info.background_color]
        return tb

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.d2r_nerf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- pyngp.Testbed surface -------------------------------------------------
    def set_camera_to_training_view(self, idx: int):
        """pyngp semantics (reference call sites reconstruction/combined_rendering.py:98,116): the view's intrinsics become
        the render intrinsics AND its lens becomes the render lens with `nerf.render_with_lens_distortion` switched on —
        the reference's frames are rendered through the OpenCV distortion its configs carry (configs/shopping_demo.json:51-56).
        Setting `nerf.render_with_lens_distortion = False` afterwards renders the same view as a pinhole."""
        self._view_idx = int(idx) % len(self.training_views)
        lens = self.training_views[self._view_idx].get("lens")
        self.nerf.render_with_lens_distortion = True
        self.nerf.render_lens = (dict(mode=LENS_OPENCV, params=tuple(float(x) for x in lens)) if lens is not None and any(lens)
                                 else dict(mode=LENS_PERSPECTIVE, params=(0.0, 0.0, 0.0, 0.0)))

    def set_nerf_camera_matrix(self, m):
        self._cam = np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4], np.float32)

    def view(self, width: int, height: int) -> View:
        tv = self.training_views[self._view_idx]
        lens = self.nerf.render_lens if self.nerf.render_with_lens_distortion else dict(mode=LENS_PERSPECTIVE, params=(0.0,) * 4)
        return View.from_training_view(width, height, tv["fx"], tv["fy"], tv["cx"], tv["cy"], tv["w"], tv["h"],
                                       scale=self.dataset_scale, offset=self.dataset_offset,
                                       background=tuple(self.background_color),
                                       min_transmittance=self.nerf.render_min_transmittance,
                                       lens_mode=int(lens["mode"]), lens_params=tuple(lens["params"]))

    def render(self, width: int, height: int, spp: int = 1, linear: bool = True) -> np.ndarray:
        """float32 [h,w,4]; Shade -> premultiplied linear RGBA, Depth -> depth in every colour
        channel (reference combined_rendering.py:127,130 read channel 0 only)."""
        if spp != 1 or not linear:
            raise NotImplementedError("the path renders with spp=1, linear=True only")
        rgba, depth = self.render_batch(self._cam[None], width, height)
        if self.render_mode == Shade:
            return rgba[0]
        out = np.zeros((height, width, 4), np.float32)
        out[..., 0] = out[..., 1] = out[..., 2] = depth[0]
        out[..., 3] = rgba[0][..., 3]
        return out

    # --- batched form ----------------------------------------------------------
    def render_batch(self, cams_nerf, width: int, height: int):
        """n cameras (each the 3x4 handed to set_nerf_camera_matrix) -> (rgba [n,h,w,4], depth [n,h,w])."""
        cams = np.ascontiguousarray(np.asarray(cams_nerf, np.float64)[:, :3, :4], np.float32)
        n = cams.shape[0]
        rgba = np.empty((n, height, width, 4), np.float32)
        depth = np.empty((n, height, width), np.float32)
        v = _lib.view_c(self.view(width, height))
        ns = C.c_uint64(0)
        self.ctx.check(self.ctx.lib.d2r_render(self.ctx.h, self.h, C.byref(v), _lib.ptr(cams), C.c_uint32(n),
                                               _lib.ptr(rgba), _lib.ptr(depth), C.byref(ns)))
        self.last_samples = int(ns.value)
        return rgba, depth

    def eval_points(self, xyz, dirs) -> np.ndarray:
        p = np.ascontiguousarray(xyz, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        out = np.empty((p.shape[0], 4), np.float32)
        self.ctx.check(self.ctx.lib.d2r_nerf_eval_points(self.ctx.h, self.h, _lib.ptr(p), _lib.ptr(d),
                                                         C.c_uint32(p.shape[0]), _lib.ptr(out)))
        return out

    def render_composite(self, view: View, obj_pose_now, cam_pose, obj_poses) -> np.ndarray:
        """K candidate poses -> uint8 [K,h,w,3] (needs Context.set_background for this view)."""
        a = np.ascontiguousarray(np.asarray(obj_pose_now, np.float64).reshape(16), np.float32)
        c = np.ascontiguousarray(np.asarray(cam_pose, np.float64).reshape(16), np.float32)
        p = np.ascontiguousarray(np.asarray(obj_poses, np.float64).reshape(-1, 16), np.float32)
        K = p.shape[0]
        frames = np.empty((K, view.height, view.width, 3), np.uint8)
        v = _lib.view_c(view)
        self.ctx.check(self.ctx.lib.d2r_render_composite(self.ctx.h, self.h, C.byref(v), _lib.ptr(a), _lib.ptr(c),
                                                         _lib.ptr(p), C.c_uint32(K), _lib.ptr(frames)))
        return frames


class ClipScorer:
    """CLIPModel + CLIPProcessor as the path uses them: images in, logits_per_image out, text
    embeddings computed once and cached by the caller."""

    def __init__(self, ctx: Context, cfg: dict, state_dict: dict):
        self.ctx, self.cfg = ctx, cfg
        blob = pack_vision_weights(state_dict, cfg)
        desc = _lib.ClipDesc(cfg["image_size"], cfg["patch_size"], cfg["hidden_size"], cfg["num_layers"],
                             cfg["num_heads"], cfg["mlp"], cfg["proj"])
        h = C.c_void_p()
        ctx.check(ctx.lib.d2r_clip_create(ctx.h, C.byref(desc), _lib.ptr(blob), C.c_size_t(blob.size), C.byref(h)))
        self.h = h
        self.logit_scale = float(np.exp(np.float32(state_dict.get("logit_scale", 4.6052))))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.d2r_clip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def score_frames(self, frames_u8, text_embeds, rot90: bool = True, return_embeds: bool = False):
        f = np.ascontiguousarray(frames_u8, np.uint8)
        n, h, w, _ = f.shape
        t = np.ascontiguousarray(text_embeds, np.float32)
        Cn = t.shape[0]
        logits = np.empty((n, Cn), np.float32)
        emb = np.empty((n, self.cfg["proj"]), np.float32) if return_embeds else None
        self.ctx.check(self.ctx.lib.d2r_clip_score_frames(
            self.ctx.h, self.h, _lib.ptr(f), C.c_uint32(n), C.c_uint32(w), C.c_uint32(h), C.c_int(int(rot90)),
            _lib.ptr(t), C.c_uint32(Cn), C.c_float(self.logit_scale), _lib.ptr(logits), _lib.ptr(emb)))
        return (logits, emb) if return_embeds else logits

    def preprocess(self, frames_u8, rot90: bool = True) -> np.ndarray:
        f = np.ascontiguousarray(frames_u8, np.uint8)
        n, h, w, _ = f.shape
        S = self.cfg["image_size"]
        pv = np.empty((n, 3, S, S), np.float32)
        self.ctx.check(self.ctx.lib.d2r_clip_preprocess(self.ctx.h, self.h, _lib.ptr(f), C.c_uint32(n), C.c_uint32(w),
                                                        C.c_uint32(h), C.c_int(int(rot90)), _lib.ptr(pv)))
        return pv

    def embed_pixels(self, pixel_values) -> np.ndarray:
        pv = np.ascontiguousarray(pixel_values, np.float32)
        n = pv.shape[0]
        emb = np.empty((n, self.cfg["proj"]), np.float32)
        self.ctx.check(self.ctx.lib.d2r_clip_embed_pixels(self.ctx.h, self.h, _lib.ptr(pv), C.c_uint32(n), _lib.ptr(emb)))
        return emb


class TextEncoder:
    """CLIP text tower on the GPU: tokenised captions -> L2-normalised text embeddings, computed
    once per task and cached by the caller (reference clip_scoring.py:177-180 redoes it per batch)."""

    def __init__(self, ctx: Context, cfg: dict, state_dict: dict):
        self.ctx, self.cfg = ctx, cfg
        blob = pack_text_weights(state_dict, cfg)
        desc = _lib.TextDesc(cfg["vocab"], cfg["ctx"], cfg["text_hidden"], cfg["text_layers"], cfg["text_heads"],
                             cfg["text_mlp"], cfg["proj"])
        h = C.c_void_p()
        ctx.check(ctx.lib.d2r_text_create(ctx.h, C.byref(desc), _lib.ptr(blob), C.c_size_t(blob.size), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.d2r_text_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, input_ids) -> np.ndarray:
        ids = np.ascontiguousarray(input_ids, np.int32)
        Cn, T = ids.shape
        out = np.empty((Cn, self.cfg["proj"]), np.float32)
        self.ctx.check(self.ctx.lib.d2r_text_encode(self.ctx.h, self.h, _lib.ptr(ids), C.c_uint32(Cn), C.c_uint32(T),
                                                    _lib.ptr(out)))
        return out


def render_score_host(ctx: Context, fg: Testbed, scorer: ClipScorer, view: View, obj_pose_now, cam_pose, obj_poses,
                      text_embeds, *, return_frames: bool = False, png_dir: str | None = None, png_first_index: int = 0,
                      png_threads: int = 0, png_level: int = -1):
    """The fused hot path for host arrays (d2r_render_score_host): K candidate poses (NGP convention) -> logits [K,C];
    the frames stay on the GPU unless `return_frames` (-> uint8 [K,h,w,3] as well) or `png_dir` (cb_rgb_%04d.png files,
    written by the library's worker threads while the GPU works on the next chunk).  Needs Context.set_background."""
    a = np.ascontiguousarray(np.asarray(obj_pose_now, np.float64).reshape(16), np.float32)
    c = np.ascontiguousarray(np.asarray(cam_pose, np.float64).reshape(16), np.float32)
    p = np.ascontiguousarray(np.asarray(obj_poses, np.float64).reshape(-1, 16), np.float32)
    t = np.ascontiguousarray(text_embeds, np.float32)
    K, Cn = p.shape[0], t.shape[0]
    logits = np.empty((K, Cn), np.float32)
    frames = np.empty((K, view.height, view.width, 3), np.uint8) if return_frames else None
    if K == 0:                                        # nothing to render: no call, no files
        return (logits, frames) if return_frames else logits
    sink = None
    if png_dir is not None:
        sink = _lib.FrameSink(os.fsencode(png_dir), png_first_index, png_threads, png_level)
    v = _lib.view_c(view)
    ctx.check(ctx.lib.d2r_render_score_host(ctx.h, fg.h, scorer.h, C.byref(v), _lib.ptr(a), _lib.ptr(c), _lib.ptr(p),
                                            C.c_uint32(K), _lib.ptr(t), C.c_uint32(Cn), C.c_float(scorer.logit_scale),
                                            _lib.ptr(logits), _lib.ptr(frames), C.byref(sink) if sink is not None else None))
    return (logits, frames) if return_frames else logits


def render_score_device(ctx: Context, fg: Testbed, scorer: ClipScorer, view: View, obj_pose_now, cam_pose,
                        obj_poses_dev_ptr: int, K: int, text_embeds, logits_dev_ptr: int, frames_out=None):
    """The fused hot path on device pointers (d2r_render_score): candidate poses already in
    HBM, logits written to HBM, asynchronous on the context's stream."""
    a = np.ascontiguousarray(np.asarray(obj_pose_now, np.float64).reshape(16), np.float32)
    c = np.ascontiguousarray(np.asarray(cam_pose, np.float64).reshape(16), np.float32)
    t = np.ascontiguousarray(text_embeds, np.float32)
    v = _lib.view_c(view)
    ctx.check(ctx.lib.d2r_render_score(ctx.h, fg.h, scorer.h, C.byref(v), _lib.ptr(a), _lib.ptr(c),
                                       C.c_void_p(obj_poses_dev_ptr), C.c_uint32(K), _lib.ptr(t),
                                       C.c_uint32(t.shape[0]), C.c_float(scorer.logit_scale),
                                       C.c_void_p(logits_dev_ptr), _lib.ptr(frames_out)))
