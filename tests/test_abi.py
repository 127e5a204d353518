"""The C-ABI library loads and exports every symbol include/d2r.h declares; without a GPU the
compute entry points fail loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

from dream2real_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    h = open(os.path.join(REPO, "include", "d2r.h")).read()
    return sorted(set(re.findall(r"D2R_API\s+[\w\s\*]+?\b(d2r_\w+)\s*\(", h)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in d2r.h but not exported"
    assert sorted(_lib.EXPORTS) == syms
    assert lib.d2r_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define D2R_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "d2r.h")).read()).group(1))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.d2r_ctx_create(0, ctypes.byref(h))
    assert rc == -2 and not h.value                                   # D2R_ERR_DEVICE
    assert b"no HIP device" in lib.d2r_last_error(None) or b"gfx950" in lib.d2r_last_error(None)
    from dream2real_amd import engine
    with pytest.raises(_lib.D2RError):
        engine.Context(0)


def test_product_does_not_use_oracle():
    """The product never imports, links or dlopens anything under oracle/ (comments may cite it)."""
    import subprocess
    pkg = os.path.join(REPO, "dream2real_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libd2r_oracle" not in src and "oracle/_build" not in src, f
                assert not re.search(r'#include\s*[<"].*oracle', src), f
    deps = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps


@pytest.mark.gpu
def test_bad_arguments_return_error_codes():
    """include/d2r.h: every function returns a negative d2r_status on bad input, with a message in d2r_last_error — no
    exception crosses the ABI, nothing aborts.  Null handles, null buffers, zero sizes, mismatched sizes, unknown options,
    handles of one context used with another's entry points."""
    import numpy as np
    C = ctypes
    lib = _lib.load()
    from dream2real_amd import engine
    from dream2real_amd.clip_model import CLIP_CONFIGS, pack_vision_weights, random_clip_state_dict
    from tests.scenes import make_scene
    ctx = engine.Context(0)
    h = ctx.h
    null = C.c_void_p(0)
    INVALID, DEVICE, UNSUPPORTED = -1, -2, -4

    def msg():
        return lib.d2r_last_error(h).decode()

    out = C.c_void_p()
    assert lib.d2r_ctx_create(12345, C.byref(out)) == INVALID and not out.value      # device index out of range
    assert lib.d2r_ctx_create(0, None) == INVALID
    assert lib.d2r_ctx_set_option(h, b"no_such_option", C.c_int64(1)) == INVALID and "no_such_option" in msg()
    assert lib.d2r_ctx_set_option(h, b"chunk", C.c_int64(0)) == INVALID
    assert lib.d2r_ctx_set_option(h, None, C.c_int64(0)) == INVALID
    assert lib.d2r_ctx_set_option(null, b"chunk", C.c_int64(1)) == INVALID
    assert lib.d2r_ctx_synchronize(null) == INVALID
    v = C.c_int64(-7)
    assert lib.d2r_ctx_get_option(h, b"chunk", C.byref(v)) == 0 and v.value == 4096             # ABI 8: tunables read back
    assert lib.d2r_ctx_set_option(h, b"mlp_f16", C.c_int64(5)) == 0 and lib.d2r_ctx_get_option(h, b"mlp_f16", C.byref(v)) == 0 and v.value == 1
    assert lib.d2r_ctx_set_option(h, b"mlp_f16", C.c_int64(1)) == 0
    assert lib.d2r_ctx_get_option(h, b"march_compact", C.byref(v)) == 0 and v.value == 1
    assert lib.d2r_ctx_get_option(h, b"ray_sort", C.byref(v)) == 0 and v.value == 1 and lib.d2r_ctx_get_option(h, b"ray_sort_log2", C.byref(v)) == 0 and v.value == 4
    assert lib.d2r_ctx_get_option(h, b"march_lds_slots", C.byref(v)) == 0 and v.value == 0      # no march launch on this context yet
    assert lib.d2r_ctx_get_option(h, b"no_such_option", C.byref(v)) == INVALID and "no_such_option" in msg()
    assert lib.d2r_ctx_get_option(h, b"chunk", None) == INVALID and lib.d2r_ctx_get_option(null, b"chunk", C.byref(v)) == INVALID
    for key, bad in ((b"gbrick_slots", 9), (b"brick_slots_total", -1), (b"lds_slots_max", 6), (b"gbrick_max_mib", 513), (b"march_compact", 2), (b"refill_min", 65), (b"march_threads", 100), (b"march_threads", 832), (b"march_threads", 2048), (b"ray_sort", 2), (b"ray_sort_log2", 5)):
        assert lib.d2r_ctx_set_option(h, key, C.c_int64(bad)) == INVALID, key
    # model creation: null descriptor, unsupported layout, bad aabb_scale
    assert lib.d2r_nerf_create(h, None, C.byref(out)) < 0
    scene = make_scene("pool_triangle")
    tb = engine.Testbed(ctx, scene.fg)
    k, lv = tb._keep, scene.fg.levels
    def desc(**over):
        d = _lib.NerfDesc(lv.n_levels, lv.n_features, _lib.ptr(k["scale"]), _lib.ptr(k["res"]), _lib.ptr(k["size"]), _lib.ptr(k["offset"]),
                          lv.n_entries, _lib.ptr(k["grid"]), _lib.ptr(k["dw1"]), _lib.ptr(k["dw2"]), _lib.ptr(k["cw1"]), _lib.ptr(k["cw2"]),
                          _lib.ptr(k["cw3"]), _lib.ptr(k["occ"]), 1, (C.c_float * 6)(*([0.0] * 6)))
        for name, v in over.items():
            setattr(d, name, v)
        return d
    for over in (dict(n_levels=12), dict(n_features=3), dict(aabb_scale=3), dict(aabb_scale=256), dict(grid_fp16=None), dict(occupancy_bits=None)):
        d = desc(**over)
        rc = lib.d2r_nerf_create(h, C.byref(d), C.byref(out))
        assert rc in (INVALID, UNSUPPORTED), (over, rc)
        assert msg()
    # rendering: null model / view / cameras, zero-sized view
    view = _lib.view_c(tb.view(32, 18))
    cams = np.zeros((1, 12), np.float32)
    rgba = np.zeros((1, 18, 32, 4), np.float32)
    assert lib.d2r_render(h, null, C.byref(view), _lib.ptr(cams), 1, _lib.ptr(rgba), None, None) == INVALID
    assert lib.d2r_render(h, tb.h, None, _lib.ptr(cams), 1, _lib.ptr(rgba), None, None) == INVALID
    assert lib.d2r_render(h, tb.h, C.byref(view), None, 1, _lib.ptr(rgba), None, None) == INVALID
    bad_view = _lib.view_c(tb.view(32, 18))
    bad_view.width = 0
    assert lib.d2r_render(h, tb.h, C.byref(bad_view), _lib.ptr(cams), 1, _lib.ptr(rgba), None, None) == INVALID
    # composite / render_score before any background was set
    poses = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (2, 1))
    frames = np.zeros((2, 18, 32, 3), np.uint8)
    eye = np.eye(4, dtype=np.float32).reshape(16)
    ctx2 = engine.Context(0)                                             # a context that never saw d2r_set_background
    tb2 = engine.Testbed(ctx2, scene.fg)
    assert lib.d2r_render_composite(ctx2.h, tb2.h, C.byref(view), _lib.ptr(eye), _lib.ptr(eye), _lib.ptr(poses), 2, _lib.ptr(frames)) == INVALID
    assert "background" in lib.d2r_last_error(ctx2.h).decode()
    assert lib.d2r_set_background(h, C.byref(view), None, None) == INVALID
    # CLIP: wrong blob size, head dim != 64, null frames, zero captions
    cfg = CLIP_CONFIGS["vit_tiny"]
    sd = random_clip_state_dict(cfg, seed=6, text=False)
    blob = pack_vision_weights(sd, cfg)
    cd = _lib.ClipDesc(cfg["image_size"], cfg["patch_size"], cfg["hidden_size"], cfg["num_layers"], cfg["num_heads"], cfg["mlp"], cfg["proj"])
    assert lib.d2r_clip_create(h, C.byref(cd), _lib.ptr(blob), C.c_size_t(blob.size - 1), C.byref(out)) == INVALID and "blob size" in msg()
    cd_bad = _lib.ClipDesc(cfg["image_size"], cfg["patch_size"], cfg["hidden_size"], cfg["num_layers"], cfg["num_heads"] + 1, cfg["mlp"], cfg["proj"])
    assert lib.d2r_clip_create(h, C.byref(cd_bad), _lib.ptr(blob), C.c_size_t(blob.size), C.byref(out)) in (INVALID, UNSUPPORTED)
    sc = engine.ClipScorer(ctx, cfg, sd)
    text = np.zeros((2, cfg["proj"]), np.float32)
    lg = np.zeros((2, 2), np.float32)
    assert lib.d2r_clip_score_frames(h, sc.h, None, 2, 32, 18, 1, _lib.ptr(text), 2, C.c_float(100.0), _lib.ptr(lg), None) == INVALID
    assert lib.d2r_clip_score_frames(h, sc.h, _lib.ptr(frames), 2, 32, 18, 1, _lib.ptr(text), 0, C.c_float(100.0), _lib.ptr(lg), None) == INVALID
    assert lib.d2r_clip_score_frames(h, null, _lib.ptr(frames), 2, 32, 18, 1, _lib.ptr(text), 2, C.c_float(100.0), _lib.ptr(lg), None) == INVALID
    # collective: gather without a communicator on a context told it is one of several ranks; double init
    assert lib.d2r_comm_init(h, None, 0, 2) == INVALID                   # world 2 needs an id blob
    assert lib.d2r_comm_init(h, None, 3, 2) == INVALID                   # rank outside the world
    assert lib.d2r_allgather_scores(h, None, C.c_size_t(4), None) == INVALID
    # physics: offsets that do not increase, pose count that does not match the grid
    v = np.zeros((4, 3), np.float32)
    off_bad = np.array([0, 0], np.uint32)
    assert lib.d2r_phys_create(h, _lib.ptr(v), _lib.ptr(off_bad), 1, None, None, 0, C.byref(out)) == INVALID
    off = np.array([0, 4], np.uint32)
    ph = C.c_void_p()
    assert lib.d2r_phys_create(h, _lib.ptr(v), _lib.ptr(off), 1, None, _lib.ptr(np.zeros(1, np.uint32)), 0, C.byref(ph)) == 0
    prm = _lib.PhysParams((C.c_uint32 * 6)(2, 2, 1, 1, 1, 1), (C.c_float * 16)(*eye), 0.0, 0.02, (C.c_float * 3)(0, 0, -1), 0.04, 1, 0, 0.0)
    valid = np.ones(3, np.uint8)
    assert lib.d2r_phys_check(h, ph, C.byref(prm), _lib.ptr(np.zeros((3, 16), np.float32)), 3, _lib.ptr(valid)) == INVALID and "sample_res" in msg()
    prm.margin = -1.0
    assert lib.d2r_phys_check(h, ph, C.byref(prm), _lib.ptr(np.zeros((4, 16), np.float32)), 4, _lib.ptr(np.ones(4, np.uint8))) == INVALID
    lib.d2r_phys_destroy(ph)
    # snapshot loader through the device entry: garbage and truncated bytes come back as error codes too
    info, views = _lib.IngpInfo(), (_lib.IngpView * 4)()
    for data in (b"\x81\xd9\xc8abc", b"\x78\x9c" + b"\x00" * 40, bytes(range(256)) * 8):
        buf = np.frombuffer(data, np.uint8)
        assert lib.d2r_nerf_load_ingp(h, _lib.ptr(buf), C.c_size_t(buf.size), C.byref(out), C.byref(info), views, 4) == INVALID
    # the host-array form of the fused pass: null poses / logits, a bad frame sink; no background on ctx2
    lg2 = np.zeros((2, 2), np.float32)
    args = lambda c, nerf, p, l, sink: lib.d2r_render_score_host(c, nerf, sc.h, C.byref(view), _lib.ptr(eye), _lib.ptr(eye), p, 2, _lib.ptr(text), 2,
                                                                C.c_float(100.0), l, None, sink)
    assert args(h, tb.h, None, _lib.ptr(lg2), None) == INVALID
    assert args(h, tb.h, _lib.ptr(poses), None, None) == INVALID
    assert args(h, null, _lib.ptr(poses), _lib.ptr(lg2), None) == INVALID
    bad_sink = _lib.FrameSink(b"/tmp", 0, 0, 12)
    assert args(h, tb.h, _lib.ptr(poses), _lib.ptr(lg2), C.byref(bad_sink)) == INVALID and "d2r_frame_sink" in msg()
    sc2 = engine.ClipScorer(ctx2, cfg, sd)
    assert lib.d2r_render_score_host(ctx2.h, tb2.h, sc2.h, C.byref(view), _lib.ptr(eye), _lib.ptr(eye), _lib.ptr(poses), 2, _lib.ptr(text), 2,
                                     C.c_float(100.0), _lib.ptr(lg2), None, None) == INVALID and "background" in lib.d2r_last_error(ctx2.h).decode()
    sc2.close()
    # the context still works after all of that
    r2, d2 = tb.render_batch(cams.reshape(1, 3, 4), 32, 18)
    assert np.isfinite(r2).all()
    sc.close(); tb.close(); tb2.close(); ctx2.close(); ctx.close()


def test_host_only_entries_return_error_codes(tmp_path):
    """The host-only entries (frame files, text files, snapshot validation) need no device: bad arguments come back as
    negative codes with a message, like everything else behind the ABI."""
    import numpy as np
    C = ctypes
    lib = _lib.load()
    img = np.zeros((4, 6, 3), np.uint8)
    err = lambda: lib.d2r_last_error(None).decode()
    assert lib.d2r_png_write(None, 6, 4, b"/tmp/x.png", 1) == -1
    assert lib.d2r_png_write(_lib.ptr(img), 0, 4, os.fsencode(str(tmp_path / "a.png")), 1) == -1 and "size" in err()
    assert lib.d2r_png_write(_lib.ptr(img), 6, 4, os.fsencode(str(tmp_path / "no_dir" / "a.png")), 1) == -1 and "cannot open" in err()
    assert lib.d2r_png_write_batch(_lib.ptr(img), 1, 6, 4, None, 0, 0, 1) == -1
    assert lib.d2r_png_write_batch(_lib.ptr(img), 0, 6, 4, os.fsencode(str(tmp_path)), 0, 0, 1) == 0        # nothing to do
    assert lib.d2r_png_read_batch(os.fsencode(str(tmp_path)), None, 0, 1, 6, 4, None, 0) == -1
    assert lib.d2r_png_read_batch(os.fsencode(str(tmp_path)), None, 7, 1, 6, 4, _lib.ptr(img), 0) == -1 and "cb_rgb_0007.png" in err()
    w, h = C.c_uint32(), C.c_uint32()
    assert lib.d2r_png_size(os.fsencode(str(tmp_path / "missing.png")), C.byref(w), C.byref(h)) == -1
    lib.d2r_savetxt.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    a = np.arange(6, dtype=np.float64)
    assert lib.d2r_savetxt(None, _lib.ptr(a), 2, 3, 0) == -1
    assert lib.d2r_savetxt(os.fsencode(str(tmp_path / "t.txt")), None, 2, 3, 0) == -1
    assert lib.d2r_savetxt(os.fsencode(str(tmp_path / "t.txt")), _lib.ptr(a), 2, 0, 0) == -1
    assert lib.d2r_savetxt(os.fsencode(str(tmp_path / "no_dir" / "t.txt")), _lib.ptr(a), 2, 3, 0) == -1 and "cannot open" in err()
    assert lib.d2r_savetxt(os.fsencode(str(tmp_path / "empty.txt")), None, 0, 3, 0) == 0 and open(tmp_path / "empty.txt").read() == ""
    assert lib.d2r_savetxt(os.fsencode(str(tmp_path / "t.txt")), _lib.ptr(a), 2, 3, 1) == 0
    np.testing.assert_array_equal(np.loadtxt(tmp_path / "t.txt"), a.reshape(2, 3))
    lib.d2r_ingp_validate.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    assert lib.d2r_ingp_validate(None, 100, None) == -1 and lib.d2r_ingp_validate(b"abc", 3, None) == -1
    assert lib.d2r_ingp_validate(bytes(range(64)), 64, None) == -1 and "snapshot" in err()
    with pytest.raises(ValueError):
        _lib.savetxt(str(tmp_path / "z.txt"), np.float64(3.0))                 # 0-d, like np.savetxt


def test_public_header_is_strict_c99_and_a_plain_c_program_links_and_runs(tmp_path):
    """The drop-in boundary is a C ABI (tier rule 2): include/d2r.h compiles as C99 with -pedantic (no C++ types, no torch types), a C
    program (examples/c_caller.c) links against libd2r.so alone and runs the host-only entry points — ABI version, error convention,
    PNG writer / reader, np.savetxt-format writer — without a GPU."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libdir = os.path.join(REPO, "dream2real_amd")
    exe = str(tmp_path / "c_caller")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(REPO, "include"),
                        os.path.join(REPO, "examples", "c_caller.c"), "-L" + libdir, "-ld2r", "-Wl,-rpath," + libdir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)
    assert sorted(os.listdir(out)) == ["cb_rgb_0005.png", "cb_rgb_0006.png", "pose_scores.txt"]
    import numpy as np
    assert np.loadtxt(out / "pose_scores.txt").shape == (3, 2)
