"""The oracle (oracle/) pinned against the committed golden vectors that were produced by
the real reference code / Hugging Face / Pillow (tests/golden/make_goldens.py)."""
import numpy as np
import pytest

from oracle import clip_ref, host_ref, render_ref
from tests.golden.frames import seeded_render_frames
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict


def test_g1_converter(goldens):
    np.testing.assert_array_equal(host_ref.converter(goldens["g1_in"]), goldens["g1_out"])


def test_g2_convert_virtual_pose(goldens):
    for (a, b, c), want in zip(goldens["g2_in"], goldens["g2_out"]):
        np.testing.assert_allclose(host_ref.convert_virtual_pose(a, b, c), want, rtol=0, atol=1e-12)


def test_g3_renderer_render_composite(goldens):
    """reference renderer.render (combined_rendering.py:73-163) with fake Testbeds ==
    oracle camera chain + oracle composite, bit-exact uint8 frames."""
    H, W = goldens["g3_hw"]
    T_WO_1 = host_ref.converter(goldens["g3_obj_pose"][None].astype(np.float32))   # [1,4,4] f32, :82
    valid = host_ref.converter(goldens["g3_valid"])
    rp = host_ref.converter(goldens["g3_render_poses"])
    bg_rgba, bg_depth = seeded_render_frames(2 * 100 + int(abs(goldens["g3_bg_cam"][0, 3]) * 1000) % 7, H, W)
    for i in range(valid.shape[0]):
        cam = host_ref.convert_virtual_pose(T_WO_1, valid[i], rp[0])[0, :3]
        np.testing.assert_array_equal(cam, goldens["g3_fg_cams"][i])
        fg_rgba, fg_depth = seeded_render_frames(1 * 100 + int(abs(cam[0, 3]) * 1000) % 7, H, W)
        got = render_ref.composite(fg_rgba, fg_depth, bg_rgba, bg_depth)
        want = goldens["g3_frames"][i]
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)


def test_g4_clip_image_processor(goldens):
    """HF CLIPImageProcessor (PIL bicubic, centre crop, rescale, normalise) == oracle."""
    n = 0
    for key in goldens.files:
        if not key.startswith("g4_") or not key.endswith("_u8"):
            continue
        _, tag, hw, _ = key.split("_")
        S = int(tag)
        hh, ww = map(int, hw.split("x"))
        r = np.random.Generator(np.random.PCG64(1000 + hh + ww))
        img = r.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:hh, 0:ww]
        img[: hh // 2] = np.stack([(yy * 255 // hh), (xx * 255 // ww), ((yy + xx) % 256)], -1)[: hh // 2]
        pv, u8 = render_ref.clip_preprocess(img, S, rot90=False)
        np.testing.assert_array_equal(u8, goldens[key])
        base = key[:-3]
        np.testing.assert_allclose(pv[:, :8, :8], goldens[base + "_pv_slice"], rtol=0, atol=2e-7)
        assert abs(pv.astype(np.float64).sum() - goldens[base + "_pv_sum"][0]) < 1e-2
        n += 1
    assert n >= 9


def test_rot90_matches_numpy():
    r = np.random.Generator(np.random.PCG64(5))
    img = r.integers(0, 256, size=(36, 64, 3), dtype=np.uint8)
    # S equal to the short edge of the rotated frame and square crop -> no resampling
    _, u8 = render_ref.clip_preprocess(img, 36, rot90=True)
    rot = np.rot90(img, k=1, axes=(0, 1))          # clip_scoring.py:145
    top = (64 - 36) // 2
    np.testing.assert_array_equal(u8, rot[top:top + 36])


def _check_clip(goldens, name, pv, tol):
    cfg = CLIP_CONFIGS[name]
    sd = random_clip_state_dict(cfg, seed=6)
    hs = []
    ie = clip_ref.vision_embeds(pv, sd, cfg, hidden_out=hs)
    te = clip_ref.text_embeds(goldens[f"g5_{name}_ids"], sd, cfg)
    np.testing.assert_allclose(ie, goldens[f"g5_{name}_image_embeds"], rtol=0, atol=tol)
    np.testing.assert_allclose(te, goldens[f"g5_{name}_text_embeds"], rtol=0, atol=tol)
    lg = clip_ref.logits_per_image(ie, te, sd["logit_scale"])
    np.testing.assert_allclose(lg, goldens[f"g5_{name}_logits"], rtol=0, atol=100 * tol)
    np.testing.assert_allclose(hs[-1][:, :3, :16], goldens[f"g5_{name}_hlast_slice"], rtol=0, atol=50 * tol)


def test_g5_clip_tiny(goldens):
    _check_clip(goldens, "vit_tiny", goldens["g5_vit_tiny_pv"], 2e-6)


def test_g5_clip_vit_b16(goldens):
    cfg = CLIP_CONFIGS["vit_b16"]
    r = np.random.Generator(np.random.PCG64(77))
    pv = r.standard_normal((2, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
    _check_clip(goldens, "vit_b16", pv, 2e-5)


def test_cone_stepping_lattice_follows_the_step_recurrence():
    """aabb_scale 2: t_{k+1} = t_k + clamp(t_k / 256, dt_min, .) — the closed form the
    oracle and the kernels share must reproduce that recurrence, stay monotone, and be a pure
    geometric sequence for rays that start beyond 256 dt_min."""
    from oracle import render_ref
    dt = np.float32(np.sqrt(3.0) / 1024.0)
    for t0 in (1e-6, 0.05, 0.3, 0.4329, 0.5, 1.7):
        t = render_ref.cone_lattice(t0, 1500).astype(np.float64)
        assert t[0] == np.float32(t0) and (np.diff(t) > 0).all()
        step = np.maximum(dt, t[:-1] / 256.0)
        np.testing.assert_allclose(np.diff(t), step, rtol=2e-3, atol=2e-7)
        seq = [float(np.float32(t0))]                      # the recurrence itself, in float64
        for _ in range(1499):
            seq.append(seq[-1] + max(float(dt), seq[-1] / 256.0))
        # the closed form switches to geometric growth at the first lattice point past 256 dt_min, the
        # recurrence a fraction of a step earlier or later: they agree to a fraction of one step
        assert np.max(np.abs(t - np.array(seq)) / np.maximum(dt, np.array(seq) / 256.0)) < 1.0
    far = render_ref.cone_lattice(0.9, 600).astype(np.float64)
    np.testing.assert_allclose(far[1:] / far[:-1], 1.0 + 1.0 / 256.0, rtol=1e-6)


# ---- round 5: tests/golden/hf_clip_r05.npz (generator: tests/golden/make_hf_goldens_r05.py, Hugging Face CLIPModel run in
# the build container) — full-depth ViT-L/14 and the adversarial-statistics weights

@pytest.mark.parametrize("key,name,weights,n,tol", [("g5adv_vit_tiny", "vit_tiny", "adversarial", 4, 2e-5),
                                                    ("g5adv_vit_b16", "vit_b16", "adversarial", 2, 3e-5),
                                                    ("g5_vit_l14", "vit_l14", "gaussian", 2, 2e-5)])
def test_oracle_vit_against_hf_goldens_round5(key, name, weights, n, tol):
    """The numpy fp32 tower against Hugging Face's CLIPModel (a) at FULL depth for ViT-L/14 (224) — g5 held the 2-layer model
    and ViT-B/16 only — and (b) under clip_model.adversarial_clip_state_dict, the regime the round-5 parity tests use the oracle
    in (massive activations, negative / spiked LayerNorm gains, near one-hot softmax rows): embeddings within `tol`, and the
    committed file is the one the generator wrote (SHA-256 of the embedding bytes)."""
    import hashlib
    import os
    from dream2real_amd.clip_model import adversarial_clip_state_dict
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_clip_r05.npz"))
    want = g[key + "_image_embeds"]
    assert hashlib.sha256(want.tobytes()).digest() == g[key + "_sha256"].tobytes()
    cfg = CLIP_CONFIGS[name]
    sd = random_clip_state_dict(cfg, 6, text=False) if weights == "gaussian" else adversarial_clip_state_dict(cfg, 6)
    r = np.random.Generator(np.random.PCG64(77))
    pv = r.standard_normal((n, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
    got = clip_ref.vision_embeds(pv, sd, cfg)
    err = float(np.abs(got - want).max())
    print(f"[pin] oracle vs HF CLIPModel, {name} ({weights} weights, {cfg['num_layers']} layers): max |d embedding| = {err:.2e} (bar {tol:.0e})")
    assert want.shape == (n, cfg["proj"]) and err <= tol


def test_half_arithmetic_emulation_of_the_oracle():
    """oracle arith modes 1 / 2 (the emulation of tiny-cuda-nn's half arithmetic, used only for the distance table of the GPU suite):
    the float -> half rounding is numpy's, bit for bit incl. subnormals, ties and overflow; mode 0 is untouched by the switch; the
    emulated field differs from the specification by half-precision noise and no more."""
    from oracle import render_ref
    from tests.scenes import make_scene
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(3000) * 10.0 ** rng.uniform(-9, 5, 3000),
                        [0, -0.0, 65504, 65519.9, 65520, 70000, 6e-8, 2.98e-8, 2.99e-8, 5.96e-8, 6.1e-5, 6.0e-5, 1.0009765625, 1.00048828125,
                         1.00146484375, -3.5e-8, np.inf, -np.inf]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(render_ref.round_half(x), want)
    scene = make_scene("shopping_trained")
    m = render_ref.OracleNerf(scene.fg)
    occ = np.argwhere(scene.fg.occupancy_bool())
    cells = occ[rng.integers(0, len(occ), 2000)]
    xyz = ((cells[:, ::-1] + rng.random((2000, 3))) / 128.0).astype(np.float32)
    d = rng.standard_normal((2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = render_ref.eval_points(m, xyz, d)
    assert render_ref.set_arith(1) == 0
    try:
        b = render_ref.eval_points(m, xyz, d)
        feats = render_ref.encode_points(m, xyz[:64])
        np.testing.assert_array_equal(feats, render_ref.round_half(feats))          # the grid's outputs are halves
        render_ref.set_arith(2)
        c = render_ref.eval_points(m, xyz, d)
    finally:
        render_ref.set_arith(0)
    np.testing.assert_array_equal(render_ref.eval_points(m, xyz, d), a)
    act = (a[:, 0] * 0.0016914558 > 1e-4) & (a[:, 0] * 0.0016914558 < 30.0)
    for v in (b, c):
        dl = np.abs(np.log(np.maximum(v[:, 0], 1e-30)) - np.log(np.maximum(a[:, 0], 1e-30)))[act]
        assert 1e-4 < dl.max() < 0.06 and np.abs(v[:, 1:] - a[:, 1:]).max() < 0.02
