"""The training view's lens (round 6).  The reference renders every frame after `set_camera_to_training_view`
(reconstruction/combined_rendering.py:98,116), which makes the view's OpenCV lens (k1, k2, p1, p2 of
configs/shopping_demo.json:51-56, carried into the NeRFs' transforms by reconstruction/train_ngp.py:171-180 and
utils/accio2ngp.py:47-56) the render lens: instant-ngp undistorts every ray's camera-space direction iteratively.

CPU part: the oracle's restatement of that iteration against an independent float64 solve and its own forward model.
GPU part (`-m gpu`): the HIP ray generators / marcher against the oracle with the lens on, the rectangle cull's
conservativeness under distortion, the Testbed semantics and the snapshot loader.
"""
import dataclasses

import numpy as np
import pytest

from dream2real_amd.scene import DEMO_LENS, LENS_OPENCV, View
from oracle import host_ref, render_ref
from tests.scenes import make_scene

STRONG_LENS = (0.31, -0.22, 0.012, -0.009)       # a far stronger lens than the RealSense's: exercises the cull's margins


def _pixel_dirs(view):
    """camera-space (x, y) of every pixel centre of a pinhole view, in the float32 operation order of the ray generators"""
    W, H = view.width, view.height
    u = (np.arange(W, dtype=np.float32) + np.float32(0.5)) / np.float32(W)
    v = (np.arange(H, dtype=np.float32) + np.float32(0.5)) / np.float32(H)
    x = (u - np.float32(view.center[0])) * np.float32(W) / np.float32(view.focal[0])
    y = (v - np.float32(view.center[1])) * np.float32(H) / np.float32(view.focal[1])
    return np.stack(np.broadcast_arrays(x[None, :], y[:, None]), -1).astype(np.float32)


def _distort64(prm, uv):
    k1, k2, p1, p2 = [float(x) for x in prm]
    u, v = uv[..., 0].astype(np.float64), uv[..., 1].astype(np.float64)
    r2 = u * u + v * v
    rad = k1 * r2 + k2 * r2 * r2
    return np.stack([u + u * rad + 2 * p1 * u * v + p2 * (r2 + 2 * u * u), v + v * rad + 2 * p2 * u * v + p1 * (r2 + 2 * v * v)], -1)


@pytest.mark.parametrize("W,H", [(640, 360), (336, 336), (160, 90)])
@pytest.mark.parametrize("prm", [DEMO_LENS, STRONG_LENS])
def test_oracle_undistortion_inverts_the_lens(W, H, prm):
    view = View.from_training_view(W, H)
    g = _pixel_dirs(view).reshape(-1, 2)
    und = render_ref.lens_undistort(prm, g)
    # distort(undistort(x)) == x: float32 forward model, and an independent float64 one
    assert np.abs(render_ref.lens_distort(prm, und) - g).max() <= 1e-6
    assert np.abs(_distort64(prm, und) - g).max() <= 1e-6
    # against a float64 Newton solve with the analytic Jacobian (independent of the central-difference iteration restated in C)
    x = g.astype(np.float64).copy()
    k1, k2, p1, p2 = [float(c) for c in prm]
    for _ in range(50):
        r = _distort64(prm, x) - g
        u, v = x[:, 0], x[:, 1]
        r2 = u * u + v * v
        rad, drad = k1 * r2 + k2 * r2 * r2, 2 * k1 + 4 * k2 * r2
        a = 1 + rad + u * u * drad + 2 * p1 * v + 6 * p2 * u
        b = u * v * drad + 2 * p1 * u + 2 * p2 * v
        d = 1 + rad + v * v * drad + 2 * p2 * u + 6 * p1 * v
        det = a * d - b * b
        x -= np.stack([(d * r[:, 0] - b * r[:, 1]) / det, (a * r[:, 1] - b * r[:, 0]) / det], -1)
    assert np.abs(und - x).max() <= 5e-7
    # and it is not a no-op: the demo lens moves the frame's corner by several pixels (~3 px at 336^2)
    shift_px = np.abs(und - g).max() * view.focal[1]
    assert shift_px > (0.008 if prm is DEMO_LENS else 0.03) * H, shift_px


def test_oracle_zero_lens_and_mode_gate():
    """mode 0 ignores the coefficients; an all-zero OpenCV lens returns the pinhole direction bit for bit"""
    g = _pixel_dirs(View.from_training_view(96, 54)).reshape(-1, 2)
    np.testing.assert_array_equal(render_ref.lens_undistort((0, 0, 0, 0), g), g)
    scene = make_scene("shopping")
    W, H = 48, 27
    m = render_ref.OracleNerf(scene.fg)
    v0 = scene.view(W, H)
    from oracle.pipeline import OraclePipeline
    cam = OraclePipeline(scene, W, H).fg_camera(scene.obj_pose)
    a = render_ref.render(m, v0, cam)
    b = render_ref.render(m, dataclasses.replace(v0, lens_mode=0, lens_params=STRONG_LENS), cam)
    c = render_ref.render(m, dataclasses.replace(v0, lens_mode=LENS_OPENCV, lens_params=(0.0, 0.0, 0.0, 0.0)), cam)
    d = render_ref.render(m, dataclasses.replace(v0, lens_mode=LENS_OPENCV, lens_params=STRONG_LENS), cam)
    for x in (b, c):
        np.testing.assert_array_equal(a[0], x[0])
        np.testing.assert_array_equal(a[1], x[1])
    assert (a[1] > 0).sum() > 20 and (a[1] != d[1]).sum() > 10          # the lens does change the rays


def test_scene_fixture_and_view_carry_the_lens():
    scene = make_scene("shopping", lens=DEMO_LENS)
    v = scene.view(64, 36)
    assert v.lens_mode == LENS_OPENCV and v.lens_params == tuple(float(x) for x in DEMO_LENS)
    assert scene.training_views[0]["lens"] == DEMO_LENS
    assert make_scene("shopping").view(64, 36).lens_mode == 0
    from dream2real_amd import _lib
    c = _lib.view_c(v)
    assert c.lens_mode == 1 and abs(c.lens_params[1] - DEMO_LENS[1]) < 1e-7


# ------------------------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def gpu():
    from dream2real_amd import engine
    ctx = engine.Context(0)
    yield dict(engine=engine, ctx=ctx)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(640, 360), (336, 336), (70, 50)])
@pytest.mark.parametrize("prm", [DEMO_LENS, STRONG_LENS])
def test_gpu_undistortion_table_is_the_oracles_bit_for_bit(gpu, W, H, prm):
    view = dataclasses.replace(View.from_training_view(W, H), lens_mode=LENS_OPENCV, lens_params=prm)
    got = gpu["ctx"].lens_undistort_view(view)
    want = render_ref.lens_undistort(prm, _pixel_dirs(view).reshape(-1, 2)).reshape(H, W, 2)
    np.testing.assert_array_equal(got, want)
    # a second view (other size / coefficients) replaces the cached table
    view2 = dataclasses.replace(View.from_training_view(W // 2, H // 2), lens_mode=LENS_OPENCV, lens_params=DEMO_LENS)
    np.testing.assert_array_equal(gpu["ctx"].lens_undistort_view(view2),
                                  render_ref.lens_undistort(DEMO_LENS, _pixel_dirs(view2).reshape(-1, 2)).reshape(H // 2, W // 2, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,W,H", [("shopping", 336, 336), ("shopping", 160, 90), ("shelf", 160, 90)])
def test_lens_render_matches_oracle(gpu, kind, W, H):
    """Testbed.render (Shade + Depth) through the demo lens against the oracle: the bars of test_render_matches_oracle."""
    from oracle.pipeline import OraclePipeline
    ctx = gpu["ctx"]
    scene = make_scene(kind, lens=DEMO_LENS)
    fg, bg = scene.testbeds(ctx)
    assert fg.nerf.render_with_lens_distortion and fg.nerf.render_lens["mode"] == LENS_OPENCV
    pipe = OraclePipeline(scene, W, H)
    assert pipe.view_fg.lens_mode == LENS_OPENCV
    poses = host_ref.sample_poses_grid(scene.scene_centre, [2, 2, 1, 1, 1, 1], scene.scene_type).reshape(-1, 4, 4)[:3]
    cams = np.stack([pipe.fg_camera(p) for p in poses])
    rgba, depth = fg.render_batch(cams, W, H)
    hits = 0
    for i, p in enumerate(poses):
        orgba, odepth = pipe.fg_render(p)
        assert ((depth[i] > 0) == (odepth > 0)).all(), "hit-pixel sets differ"
        hits += int((odepth > 0).sum())
        np.testing.assert_allclose(rgba[i], orgba, rtol=0, atol=5e-3)
        np.testing.assert_allclose(depth[i], odepth, rtol=0, atol=2e-3)
    assert hits > 200
    assert abs(fg.last_samples - pipe.n_samples) <= 0.01 * pipe.n_samples
    # the background through the lens
    cam = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    brgba, bdepth = bg.render_batch(cam[None, :3], W, H)
    orgba, odepth = pipe.background()
    assert ((bdepth[0] > 0) == (odepth > 0)).all()
    np.testing.assert_allclose(brgba[0], orgba, rtol=0, atol=5e-3)
    # switching the flag off on the Testbed renders the pinhole frame of the lens-free scene, bit for bit
    fg.nerf.render_with_lens_distortion = False
    plain = make_scene(kind)
    fg0, _ = plain.testbeds(ctx)
    a, b = fg.render_batch(cams, W, H), fg0.render_batch(cams, W, H)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert (a[1] > 0).sum() > 200
    fg.nerf.render_with_lens_distortion = True
    assert (fg.render_batch(cams, W, H)[1] != a[1]).sum() > 20                      # ... and the lens frames are different ones
    for tb in (fg, bg, fg0):
        tb.close()


@pytest.mark.gpu
def test_zero_lens_is_the_pinhole_bit_for_bit(gpu):
    """an all-zero OpenCV lens goes through the table path (k_lens_table + the lookup in make_ray) and must reproduce the
    perspective frames exactly: the Newton step of a zero lens is exactly zero"""
    from oracle.pipeline import OraclePipeline
    ctx = gpu["ctx"]
    scene = make_scene("shopping")
    fg, bg = scene.testbeds(ctx)
    W, H = 160, 90
    cam = OraclePipeline(scene, W, H).fg_camera(scene.obj_pose)
    a = fg.render_batch(cam[None], W, H)
    fg.nerf.render_lens = dict(mode=LENS_OPENCV, params=(0.0, 0.0, 0.0, 0.0))
    assert fg.view(W, H).lens_mode == LENS_OPENCV
    b = fg.render_batch(cam[None], W, H)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert (a[1] > 0).sum() > 100
    fg.close(); bg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,W,H,prm", [("shopping", 640, 360, DEMO_LENS), ("shopping", 336, 336, DEMO_LENS), ("shopping", 320, 180, STRONG_LENS),
                                          ("shopping_huge", 320, 180, STRONG_LENS), ("shelf", 320, 180, STRONG_LENS)])
def test_lens_composite_matches_oracle_and_rect_cull_is_conservative(gpu, kind, W, H, prm):
    """renderer.render through the lens: uint8 frames within the 1-LSB bar of the oracle, and the rectangle cull (which has to
    bound where the lens puts the object's box) produces exactly the full-frame generator's frames, ray and sample counts —
    also for a lens several times stronger than the RealSense's, an object that fills a large part of the frame, and objects
    partly / wholly off screen."""
    from oracle.pipeline import OraclePipeline
    ctx = gpu["ctx"]
    scene = make_scene(kind, lens=prm)
    fg, bg = scene.testbeds(ctx)
    pipe = OraclePipeline(scene, W, H)
    sr = [3, 3, 1, 1, 1, 1] if kind != "shelf" else [2, 2, 2, 1, 1, 1]
    base = host_ref.sample_poses_grid(scene.scene_centre, sr, scene.scene_type).reshape(-1, 4, 4)
    extra = []
    for dx, dy, dz in [(0.18, 0.0, 0.0), (-0.22, 0.1, 0.0), (0.0, 0.16, 0.0), (0.6, 0.6, 0.0), (0.0, 0.0, 0.3), (0.0, 0.0, 0.52)]:
        p = np.array(scene.obj_pose, np.float64).reshape(4, 4).copy()
        p[:3, 3] += (dx, dy, dz)
        extra.append(p)
    poses = np.concatenate([base, np.stack(extra)]).astype(np.float32)
    cam_bg = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    brgba, bdepth = bg.render_batch(cam_bg[None, :3], W, H)
    view = fg.view(W, H)
    assert view.lens_mode == LENS_OPENCV
    ctx.set_background(view, brgba[0], bdepth[0])
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    out = {}
    for flag in (1, 0):
        ctx.set_option("raygen_rect", flag)
        frames = fg.render_composite(view, T1, cam_bg, host_ref.converter(poses))
        st = ctx.render_stats()
        out[flag] = (frames, st["rays_alive"], st["samples"])
    ctx.set_option("raygen_rect", 1)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1] > 500 and out[0][2] == out[1][2]
    # against the oracle on a few of the candidates (the oracle renders every pixel of every frame on the host)
    n_chk = 3 if W * H > 100000 else 6
    want = pipe.frames(poses[:n_chk], bg=(brgba[0], bdepth[0]))
    diff = np.abs(out[1][0][:n_chk].astype(int) - want.astype(int)).max(-1)
    # the bar of test_composited_frames_match_oracle (1 LSB, under 2 % of the pixels) stated per OBJECT pixel, because the 5x object of
    # `shopping_huge` covers a quarter of the frame: about a third of an object's pixels sit within one bf16 rounding of a quantisation step
    background = out[1][0][len(base) + 3]                     # the candidate far outside the frustum: the plain background frame
    obj = (want != background[None]).any(-1)
    assert diff.max() <= 1 and (diff > 0).sum() <= 0.5 * obj.sum() + 0.002 * diff.size, (diff.max(), (diff > 0).mean(), obj.mean())
    assert obj.sum() > 100
    fg.close(); bg.close()


@pytest.mark.gpu
def test_fused_render_score_through_the_lens(gpu):
    """d2r_render_score with the demo lens: logits within the bar of the oracle pipeline (oracle render through the lens +
    fp32 tower), identical with the background-patch / layer-0 reuse paths on and off (they rest on the candidates' rectangles)."""
    import torch
    from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
    from tests.parity_utils import OraclePipeline, logit_bar, oracle_logits, random_unit_text_embeds
    engine, ctx = gpu["engine"], gpu["ctx"]
    scene = make_scene("shopping", lens=DEMO_LENS)
    fg, bg = scene.testbeds(ctx)
    cfg = CLIP_CONFIGS["vit_tiny"]
    sd = random_clip_state_dict(cfg, seed=6, text=False)
    sc = engine.ClipScorer(ctx, cfg, sd)
    text = random_unit_text_embeds(cfg["proj"], 2)
    W, H = 160, 90
    pipe = OraclePipeline(scene, W, H)
    obg = pipe.background()
    view = fg.view(W, H)
    ctx.set_background(view, obg[0], obg[1])
    poses = host_ref.sample_poses_grid(scene.scene_centre, [5, 3, 1, 1, 1, 1], scene.scene_type)
    pn = host_ref.converter(poses.reshape(-1, 4, 4)).reshape(-1, 16)
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    TC = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    dev = torch.device("cuda:0")
    p_dev = torch.from_numpy(pn).to(dev)
    res = {}
    for reuse in (1, 0):
        ctx.set_option("prep_reuse", reuse)
        ctx.set_option("l0_reuse", reuse)
        lg = torch.zeros((len(pn), 2), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        engine.render_score_device(ctx, fg, sc, view, T1, TC, p_dev.data_ptr(), len(pn), text, lg.data_ptr(), None)
        ctx.synchronize()
        res[reuse] = lg.cpu().numpy()
    ctx.set_option("prep_reuse", 1)
    ctx.set_option("l0_reuse", 1)
    np.testing.assert_array_equal(res[0], res[1])
    olg, _ = oracle_logits(pipe.frames(poses.reshape(-1, 4, 4), bg=obg), cfg, sd, text)
    err = float(np.abs(res[1] - olg).max() / sc.logit_scale)
    assert err <= logit_bar(cfg, err, "fused d2r_render_score through the demo lens, 15 candidates")
    # the lens matters at the level of the scores: the pinhole pipeline's logits are a different set
    plain = make_scene("shopping")
    olg0, _ = oracle_logits(OraclePipeline(plain, W, H).frames(poses.reshape(-1, 4, 4)[:4]), cfg, sd, text)
    assert np.abs(olg0 - olg[:4]).max() / sc.logit_scale > 1e-4
    sc.close(); fg.close(); bg.close()


@pytest.mark.gpu
def test_snapshot_lens_reaches_the_render(gpu, tmp_path):
    """metadata[].lens of a snapshot -> Testbed.training_views -> set_camera_to_training_view -> the render lens; a snapshot
    that carries render_with_lens_distortion loads (round 5 refused it) with the flag set before any view is selected."""
    import msgpack
    import zlib
    from tests.ingp_writer import save_ingp
    from oracle.pipeline import OraclePipeline
    engine, ctx = gpu["engine"], gpu["ctx"]
    scene = make_scene("shopping", lens=DEMO_LENS)
    path = str(tmp_path / "fg_base.ingp")
    views = [dict(scene.training_views[0]), dict(scene.training_views[0], lens=None), dict(scene.training_views[0], lens=STRONG_LENS)]
    save_ingp(path, scene.fg, training_views=views)
    tb = engine.Testbed.from_snapshot(ctx, path)
    assert not tb.nerf.render_with_lens_distortion                      # a fresh Testbed: off until a training view is selected
    assert [v["lens"] is not None for v in tb.training_views] == [True, False, True]
    np.testing.assert_allclose(tb.training_views[0]["lens"], np.float32(DEMO_LENS), rtol=0, atol=1e-9)        # the C ABI carries floats
    np.testing.assert_allclose(tb.training_views[2]["lens"], np.float32(STRONG_LENS), rtol=0, atol=1e-9)
    W, H = 160, 90
    cam = OraclePipeline(scene, W, H).fg_camera(scene.obj_pose)
    fg, bg = scene.testbeds(ctx)
    plain_fg, plain_bg = make_scene("shopping").testbeds(ctx)
    pin = tb.render_batch(cam[None], W, H)                              # before set_camera_to_training_view: pinhole
    np.testing.assert_array_equal(pin[0], plain_fg.render_batch(cam[None], W, H)[0])
    tb.set_camera_to_training_view(0)
    assert tb.nerf.render_with_lens_distortion and tb.nerf.render_lens["mode"] == LENS_OPENCV
    np.testing.assert_array_equal(tb.render_batch(cam[None], W, H)[0], fg.render_batch(cam[None], W, H)[0])
    tb.set_camera_to_training_view(1)                                   # a view without a lens: perspective again
    assert tb.nerf.render_with_lens_distortion and tb.nerf.render_lens["mode"] == 0
    np.testing.assert_array_equal(tb.render_batch(cam[None], W, H)[0], pin[0])
    tb.set_camera_to_training_view(2)
    assert (tb.render_batch(cam[None], W, H)[1] != pin[1]).sum() > 20
    tb.close()
    # the flag in the file
    cfgd = msgpack.unpackb(zlib.decompress(open(path, "rb").read()), raw=False)
    cfgd["snapshot"]["nerf"]["render_with_lens_distortion"] = True
    open(path, "wb").write(zlib.compress(msgpack.packb(cfgd, use_bin_type=True), 1))
    tb = engine.Testbed.from_snapshot(ctx, path)
    assert tb.nerf.render_with_lens_distortion and tb.snapshot_unknown_keys == 0
    tb.close()
    for t in (fg, bg, plain_fg, plain_bg):
        t.close()
