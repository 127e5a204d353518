"""Snapshot reader round trip on a synthetic .ingp written in the believed instant-ngp layout
(no real snapshot is available offline — DESIGN.md section 1)."""
import numpy as np
import pytest

from dream2real_amd import ingp
from tests.ingp_writer import save_ingp
from tests.scenes import make_scene


def test_roundtrip(tmp_path):
    scene = make_scene("pool_triangle")
    views = [dict(fx=900.0, fy=910.0, cx=640.0, cy=350.0, w=1280, h=720), dict(fx=450.0, fy=455.0, cx=320.0, cy=175.0, w=640, h=360)]
    path = str(tmp_path / "fg_base.ingp")
    save_ingp(path, scene.fg, training_views=views, dataset_scale=1.0, dataset_offset=(0.0, 0.3, 0.5))
    model, info = ingp.load_ingp(path)
    for name in ("grid", "dw1", "dw2", "cw1", "cw2", "cw3", "occ_bits"):
        np.testing.assert_array_equal(getattr(model, name), getattr(scene.fg, name))
    np.testing.assert_array_equal(model.levels.offset, scene.fg.levels.offset)
    assert info["dataset_offset"] == (0.0, 0.3, 0.5) and info["aabb_scale"] == 1
    assert len(info["training_views"]) == 2
    assert abs(info["training_views"][1]["cx"] - 320.0) < 1e-9 and info["training_views"][0]["w"] == 1280


def test_morton_order_is_a_permutation():
    m = ingp._morton_order()
    assert m[0] == 0 and m[1] == 1 and m[2] == 128 and m[4] == 128 * 128 and m[7] == 1 + 128 + 128 * 128
    assert np.array_equal(np.sort(m), np.arange(128 ** 3))


def test_rejects_unknown_layouts(tmp_path):
    import msgpack, zlib
    scene = make_scene("pool_triangle")
    path = str(tmp_path / "x.ingp")
    save_ingp(path, scene.fg)
    cfg = msgpack.unpackb(zlib.decompress(open(path, "rb").read()), raw=False)
    cfg["snapshot"]["params_binary"] = cfg["snapshot"]["params_binary"][:-2]
    open(path, "wb").write(zlib.compress(msgpack.packb(cfg, use_bin_type=True)))
    with pytest.raises(ValueError):
        ingp.load_ingp(path)
    cfg["snapshot"]["nerf"]["aabb_scale"] = 3            # not a power of two
    open(path, "wb").write(zlib.compress(msgpack.packb(cfg, use_bin_type=True)))
    with pytest.raises(NotImplementedError):
        ingp.load_ingp(path)


def test_aabb_scale_2_snapshot_round_trip(tmp_path):
    """two Morton-ordered occupancy cascades and the wider level spacing (2048 * aabb_scale at the
    finest level) survive save -> load"""
    scene = make_scene("shelf")
    path = str(tmp_path / "shelf.ingp")
    save_ingp(path, scene.bg)
    model, info = ingp.load_ingp(path)
    assert model.aabb_scale == 2 and info["aabb_scale"] == 2
    np.testing.assert_array_equal(model.occ_bits, scene.bg.occ_bits)
    np.testing.assert_array_equal(model.levels.res, scene.bg.levels.res)
    np.testing.assert_array_equal(model.grid, scene.bg.grid)


def test_occupancy_threshold_follows_instant_ngp():
    """Graded densities: the threshold is min(0.01, mean of max(d, 0) over ALL cells of cascade 0) — for a
    sparse object the mean is far below 0.01 — and coarser cascades OR in the max-pool of the finer one."""
    r = np.random.Generator(np.random.PCG64(11))
    order = ingp._morton_order()
    lin = np.zeros((2, 128, 128, 128), np.float32)              # [c, z, y, x]
    lin[0, 60:68, 60:68, 60:68] = r.uniform(0.0, 0.02, (8, 8, 8)).astype(np.float32)
    lin[0, 10, 10, 10] = -3.0                                   # negative densities count as 0 in the mean
    lin[1, 5, 6, 7] = 0.5
    dens = lin.reshape(2, -1)[:, order]                         # Morton order, as stored
    occ = ingp.occupancy_from_density(dens).reshape(2, 128, 128, 128)
    mean = np.maximum(lin[0], 0).sum(dtype=np.float64) / 128 ** 3
    assert mean < 1e-5                                          # sparse object: threshold = mean, not 0.01
    np.testing.assert_array_equal(occ[0], lin[0] > mean)
    assert occ[0].sum() > 400                                   # nearly all 512 graded cells (min(0.01, positive-mean) would keep ~half)
    want1 = lin[1] > mean
    want1[32:96, 32:96, 32:96] |= occ[0].reshape(64, 2, 64, 2, 64, 2).any(axis=(1, 3, 5))
    np.testing.assert_array_equal(occ[1], want1)
    assert occ[1, 5, 6, 7] and occ[1, 32 + 30, 32 + 30, 32 + 30]
    # a dense scene: mean above 0.01 -> the constant threshold
    dens2 = np.full((1, 128 ** 3), 0.05, np.float32)
    dens2[0, :1000] = 0.009
    occ2 = ingp.occupancy_from_density(dens2)
    assert occ2.sum() == 128 ** 3 - 1000


def test_l8f4_layout_background_and_render_aabb_round_trip(tmp_path):
    """The other layout a snapshot may carry (L = 8, F = 4), the saved background colour and a cropped
    render_aabb survive save -> load; layouts with another input width are refused."""
    import dataclasses
    from dream2real_amd.scene import grid_levels
    from tests.scenes import ellipsoid_occupancy, make_synthetic_nerf
    levels = grid_levels(n_levels=8, n_features=4, log2_hashmap_size=14)
    model = make_synthetic_nerf(ellipsoid_occupancy((0.5, 0.5, 0.5), (0.1, 0.1, 0.1)), seed_grid=3, seed_mlp=4, levels=levels)
    model = dataclasses.replace(model, render_aabb=(0.1, 0.2, 0.3, 0.9, 0.8, 0.7))
    path = str(tmp_path / "m.ingp")
    save_ingp(path, model, background_color=(0.0, 0.0, 0.0, 0.0))
    got, info = ingp.load_ingp(path)
    assert (got.levels.n_levels, got.levels.n_features) == (8, 4) and got.grid.shape == (levels.n_entries, 4)
    np.testing.assert_array_equal(got.grid, model.grid)
    np.testing.assert_array_equal(got.dw1, model.dw1)
    np.testing.assert_allclose(got.render_aabb, model.render_aabb, rtol=0, atol=1e-7)
    assert info["background_color"] == [0.0, 0.0, 0.0, 0.0]
    # render_aabb equal to the whole box is "no crop"
    save_ingp(path, dataclasses.replace(model, render_aabb=(0.0, 0.0, 0.0, 1.0, 1.0, 1.0)))
    assert ingp.load_ingp(path)[0].render_aabb is None
    import msgpack, zlib
    cfg = msgpack.unpackb(zlib.decompress(open(path, "rb").read()), raw=False)
    cfg["encoding"]["n_levels"] = 12
    open(path, "wb").write(zlib.compress(msgpack.packb(cfg, use_bin_type=True)))
    with pytest.raises(ValueError):
        ingp.load_ingp(path)


# ---- the C reader (libd2r.so): host-only entry d2r_ingp_inspect shares the msgpack / zlib code of d2r_nerf_load_ingp

def test_c_inspect_lists_the_tree_and_what_the_loader_reads(tmp_path):
    """d2r_ingp_inspect on a synthetic snapshot: every leaf is listed, the ones the loader reads are marked, extra keys
    (what a real instant-ngp file will carry beyond the believed layout) show up as ignored, and the derived parameter
    counts agree with the file."""
    import msgpack, zlib
    from dream2real_amd import _lib
    scene = make_scene("pool_triangle")
    path = str(tmp_path / "fg_base.ingp")
    save_ingp(path, scene.fg, training_views=[dict(fx=900.0, fy=910.0, cx=640.0, cy=350.0, w=1280, h=720)] * 3)
    cfg = msgpack.unpackb(zlib.decompress(open(path, "rb").read()), raw=False)
    cfg["snapshot"]["nerf"]["cone_angle_constant"] = 0.0                 # checked: tied to aabb_scale
    cfg["snapshot"]["camera"] = {"fov_axis": 1, "zoom": 1.0}             # the GUI's own camera: known not to matter
    cfg["snapshot"]["mystery"] = {"knob": 3}                             # a key the loader has never heard of
    cfg["dir_encoding"] = {"otype": "SphericalHarmonics", "degree": 4}
    blob = zlib.compress(msgpack.packb(cfg, use_bin_type=True), 1)
    text = _lib.ingp_inspect(blob)
    lines = [ln for ln in text.splitlines() if not ln.startswith("#")]
    by_path = {ln.split(" ")[3]: ln for ln in lines}
    assert by_path["snapshot.params_binary"].startswith("R bin ")
    assert by_path["snapshot.density_grid_binary"].startswith("R bin ")
    assert by_path["encoding.n_levels"].startswith("R int 1 ") and by_path["encoding.n_levels"].endswith("= 16")
    assert by_path["snapshot.nerf.cone_angle_constant"].startswith("C float ")
    assert by_path["snapshot.camera.zoom"].startswith("- float ") and by_path["snapshot.mystery.knob"].startswith("? int ")
    assert by_path["dir_encoding.degree"].startswith("C int ") and by_path["dir_encoding.otype"].endswith('= "SphericalHarmonics"')
    assert "# unknown_keys 1 " in text
    info = _lib.ingp_validate(blob)
    assert info.n_unknown_keys == 1 and info.n_views == 3 and (info.n_levels, info.n_features, info.aabb_scale) == (16, 2, 1)
    assert by_path["snapshot.nerf.dataset.metadata[]"] == "- array 3 snapshot.nerf.dataset.metadata[]"
    assert by_path["snapshot.nerf.dataset.metadata[].focal_length"].startswith("R array 2 ")
    derived = [ln for ln in text.splitlines() if ln.startswith("# derived: grid_entries")][0].split()
    vals = dict(zip(derived[2::2], derived[3::2]))
    assert int(vals["n_params_expected"]) == int(vals["params_binary_halves"]) == len(cfg["snapshot"]["params_binary"]) // 2
    assert int(vals["grid_entries"]) == scene.fg.levels.n_entries
    assert sum(ln.startswith("# level ") for ln in text.splitlines()) == 16
    # the same bytes uncompressed (.msgpack) read the same
    assert _lib.ingp_inspect(msgpack.packb(cfg, use_bin_type=True)).splitlines()[2:] == text.splitlines()[2:]


def test_c_reader_returns_errors_on_truncated_and_malformed_input(tmp_path):
    """Nothing a file can contain may take the process down (no exception crosses the C ABI, include/d2r.h): truncated
    msgpack (also inside a map key, the case that used to build a std::string from a null pointer), truncated and
    corrupted zlib streams, garbage, absurd lengths."""
    import msgpack, zlib
    from dream2real_amd import _lib
    scene = make_scene("pool_triangle")
    path = str(tmp_path / "fg_base.ingp")
    save_ingp(path, scene.fg)
    blob = open(path, "rb").read()
    raw = zlib.decompress(blob)
    bad = [
        bytes([0x81, 0xd9, 0xc8]) + b"abc",                    # map of 1, str8 key of 200 bytes, 3 present
        bytes([0x81, 0xa3]) + b"ab",                           # fixstr key cut short
        bytes([0x82, 0xa1]) + b"a" + bytes([0x01, 0xa1]),      # second key missing its payload
        bytes([0xdf, 0xff, 0xff, 0xff, 0xff]),                 # map32 of 4 G entries, nothing behind it
        bytes([0xdd, 0xff, 0xff, 0xff, 0xff, 0x01]),           # array32 of 4 G entries
        bytes([0xc6, 0xff, 0xff, 0xff, 0xff]) + b"x" * 8,      # bin32 of 4 GiB
        bytes([0x93, 0x01, 0x02, 0x03]) + b"\x00" * 4,         # valid msgpack, not a map
        raw[: len(raw) // 2], raw[:100], raw[:5],              # truncated uncompressed snapshots
        blob[: len(blob) // 2], blob[:20],                     # truncated zlib streams
        blob[:200] + bytes(200) + blob[400:],                  # corrupted zlib stream
        zlib.compress(bytes([0x81, 0xd9, 0xc8]) + b"abc"),     # the map-key case behind a valid zlib stream
        zlib.compress(bytes(range(256)) * 64),                 # garbage behind a valid zlib stream
        bytes([0x1f, 0x8b, 0x08, 0x00]) + b"\x00" * 32,        # gzip magic, nonsense body
        bytes([0x91] * 100),                                   # nesting deeper than the reader allows
    ]
    for k, data in enumerate(bad):
        with pytest.raises(_lib.D2RError):
            _lib.ingp_inspect(data)
    # sanity: the intact file still reads
    assert "snapshot.params_binary" in _lib.ingp_inspect(blob)


def test_c_loader_names_the_key_it_refuses(tmp_path):
    """The layout is believed, so nothing is defaulted: d2r_ingp_validate (every check d2r_nerf_load_ingp makes, host only)
    on mutated snapshots — missing keys, fp32 parameters, wrong kinds and sizes, extra network input dimensions, another
    activation / interpolation / SH degree, a rotated crop box, exposure, lens distortion, unknown keys in the sections
    that define the network — each refusal names the key."""
    import copy
    import msgpack, zlib
    from dream2real_amd import _lib
    scene = make_scene("pool_triangle")
    path = str(tmp_path / "fg_base.ingp")
    save_ingp(path, scene.fg)
    base = msgpack.unpackb(zlib.decompress(open(path, "rb").read()), raw=False)
    pack = lambda c: zlib.compress(msgpack.packb(c, use_bin_type=True), 1)
    _lib.ingp_validate(pack(base))                                      # the intact file passes

    def mutated(fn):
        c = copy.deepcopy(base)
        fn(c)
        return pack(c)

    def drop(*keys):
        def fn(c):
            for k in keys[:-1]:
                c = c[k]
            del c[keys[-1]]
        return fn

    def put(value, *keys):
        def fn(c):
            for k in keys[:-1]:
                c = c[k] if isinstance(c, list) else c.setdefault(k, {})
            c[keys[-1]] = value
        return fn

    n_half = len(base["snapshot"]["params_binary"]) // 2
    cases = [
        (drop("encoding", "n_levels"), "encoding.n_levels is missing"),
        (drop("encoding", "log2_hashmap_size"), "encoding.log2_hashmap_size is missing"),
        (drop("network", "n_neurons"), "network.n_neurons is missing"),
        (drop("rgb_network"), "<root>.rgb_network is missing"),
        (drop("dir_encoding"), "<root>.dir_encoding is missing"),
        (drop("snapshot", "params_type"), "snapshot.params_type is missing"),
        (drop("snapshot", "density_grid_size"), "snapshot.density_grid_size is missing"),
        (drop("snapshot", "density_grid_binary"), "snapshot.density_grid_binary is missing"),
        (drop("snapshot", "nerf", "dataset", "scale"), "snapshot.nerf.dataset.scale is missing"),
        (drop("snapshot", "nerf", "dataset", "offset"), "snapshot.nerf.dataset.offset is missing"),
        (lambda c: (c["snapshot"]["nerf"].pop("aabb_scale"), c["snapshot"]["nerf"]["dataset"].pop("aabb_scale")), "aabb_scale"),
        (put(2, "snapshot", "nerf", "dataset", "aabb_scale"), "disagree"),
        (put("float", "snapshot", "params_type"), "snapshot.params_type = 'float'"),
        (lambda c: (put("float", "snapshot", "params_type")(c), put(np.zeros(n_half, np.float32).tobytes(), "snapshot", "params_binary")(c)),
         "size of fp32 parameters"),
        (put(np.zeros(n_half, np.float32).tobytes(), "snapshot", "params_binary"), "the size of fp32 parameters"),
        (put(n_half + 64, "snapshot", "n_params"), "snapshot.n_params"),
        (put("abc", "snapshot", "params_binary"), "snapshot.params_binary has msgpack kind 'str'"),
        (put(np.zeros(128 ** 3, np.float32).tobytes(), "snapshot", "density_grid_binary"), "size of fp32 densities"),
        (put(np.zeros(128 ** 3 // 8, np.uint8).tobytes(), "snapshot", "density_grid_binary"), "size of a bitfield"),
        (put(1, "snapshot", "nerf", "dataset", "n_extra_learnable_dims"), "n_extra_learnable_dims"),
        (put(4, "snapshot", "nerf", "n_extra_dims"), "snapshot.nerf.n_extra_dims"),
        (put("Squareplus", "network", "activation"), "network.activation = 'Squareplus'"),
        (put("Sigmoid", "rgb_network", "output_activation"), "rgb_network.output_activation"),
        (put("Smoothstep", "encoding", "interpolation"), "encoding.interpolation = 'Smoothstep'"),
        (put("Tiled", "encoding", "type"), "encoding.type = 'Tiled'"),
        (put(2, "encoding", "n_dims_to_encode"), "encoding.n_dims_to_encode"),
        (put(True, "encoding", "stochastic_interpolation"), "unknown key encoding.stochastic_interpolation"),
        (put(0.1, "network", "dropout"), "unknown key network.dropout"),
        (put({"otype": "SphericalHarmonics", "degree": 3}, "dir_encoding"), "dir_encoding.degree = 3"),
        (put({"otype": "Frequency", "n_frequencies": 4}, "dir_encoding"), "dir_encoding.otype = 'Frequency'"),
        (put({"otype": "Composite", "nested": [{"otype": "SphericalHarmonics", "degree": 4, "n_dims_to_encode": 3}, {"otype": "OneBlob", "n_bins": 4}]},
             "dir_encoding"), "dir_encoding.nested[1]"),
        (put([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], "snapshot", "render_aabb_to_local"), "snapshot.render_aabb_to_local is not the identity"),
        (put(1.5, "snapshot", "exposure"), "snapshot.exposure = 1.5"),
        # lens models the path does not implement refuse the snapshot, naming the view (perspective and OpenCV are read)
        (put({"k1": 0.1, "k2": 0.0, "k3": 0.01, "k4": 0.0}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "metadata[0].lens is an OpenCV fisheye lens"),
        (put({"latlong": True}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "metadata[0].lens is a lat-long"),
        (put({"ftheta_p0": 1.0, "w": 10, "h": 10}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "metadata[0].lens is an f-theta lens"),
        (put({"k1": 0.1, "k2": 0.0}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "without all of k1, k2, p1, p2"),
        (put({"k1": float("nan"), "k2": 0.0, "p1": 0.0, "p2": 0.0}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "not finite"),
        (put({"mode": 3, "params": [0.0] * 7}, "snapshot", "nerf", "dataset", "metadata", 0, "lens"), "mode other than perspective"),
        (put(True, "snapshot", "nerf", "dataset", "is_hdr"), "is_hdr"),
        (put(512, "snapshot", "nerf", "dataset", "envmap_resolution"), "envmap_resolution"),
        (put([0, 512], "snapshot", "nerf", "dataset", "envmap_resolution"), "envmap_resolution = [0, 512]"),
        (put(0.00390625, "snapshot", "nerf", "cone_angle_constant"), "snapshot.nerf.cone_angle_constant"),
        (put(1, "snapshot", "nerf", "rgb_activation"), "snapshot.nerf.rgb_activation"),
        (put(256, "snapshot", "density_grid_size"), "density_grid_size must be 128"),
        (lambda c: c["snapshot"]["nerf"]["dataset"]["metadata"][0].pop("focal_length"), "metadata[0] lacks focal_length"),
        (put([0.0, 0.3], "snapshot", "nerf", "dataset", "offset"), "offset must hold 3 numbers"),
        (put(128, "network", "n_neurons"), "network.n_neurons 128"),
        (put(12, "encoding", "n_levels"), "encoding.n_levels 12"),
    ]
    for k, (fn, needle) in enumerate(cases):
        with pytest.raises(_lib.D2RError, match=__import__("re").escape(needle)):
            _lib.ingp_validate(mutated(fn))
    # accepted variations: identity crop rotation, per_level_scale written back by instant-ngp, a Composite direction
    # encoding with the Identity member base.json carries, CutlassMLP
    ok = mutated(lambda c: (put([[1, 0, 0], [0, 1, 0], [0, 0, 1]], "snapshot", "render_aabb_to_local")(c), put(0.0, "snapshot", "exposure")(c),
                            put("CutlassMLP", "network", "otype")(c), put("Linear", "encoding", "interpolation")(c),
                            put({"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4},
                                                                  {"otype": "Identity", "n_bins": 4, "degree": 4}]}, "dir_encoding")(c),
                            put(3, "snapshot", "nerf", "density_activation")(c), put(0.0, "snapshot", "nerf", "cone_angle_constant")(c),
                            # upstream writes ivec2 / bool fields where this reader once demanded a scalar 0
                            put([0, 0], "snapshot", "nerf", "dataset", "envmap_resolution")(c), put(False, "snapshot", "nerf", "dataset", "is_hdr")(c),
                            put(False, "snapshot", "nerf", "dataset", "from_mitsuba")(c), put(False, "snapshot", "nerf", "render_with_lens_distortion")(c)))
    info = _lib.ingp_validate(ok)
    assert info.n_unknown_keys == 0 and info.render_with_lens_distortion == 0
    # the flag is read, not refused (round 6): the reference sets it (reconstruction/train_ngp.py:70) and every render goes through
    # set_camera_to_training_view, which sets it anyway
    on = mutated(lambda c: (put(True, "snapshot", "nerf", "render_with_lens_distortion")(c),
                            put({"k1": 0.096692, "k2": -0.166479, "p1": -0.000194, "p2": 0.002049}, "snapshot", "nerf", "dataset", "metadata", 0, "lens")(c)))
    info = _lib.ingp_validate(on)
    assert info.n_unknown_keys == 0 and info.render_with_lens_distortion == 1
