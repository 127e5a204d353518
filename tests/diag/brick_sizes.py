import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dream2real_amd import engine
from synthetic_scenes import make_scene
for name, opts in (("shopping", {}), ("shopping", {"lds_slots_max": 0}), ("shopping", {"lds_slots_max": 3}), ("shopping_big", {}), ("shopping_huge", {}), ("shelf", {}), ("pool_triangle", {})):
    scene = make_scene(name)
    ctx = engine.Context(0)
    for k, v in opts.items(): ctx.set_option(k, v)
    tb = engine.Testbed(ctx, scene.fg)
    cam = np.asarray(scene.cam_poses, np.float32)
    from dream2real_amd.accio2ngp import converter
    c = converter(cam)[:1, :3]
    tb.render_batch(c, 64, 36)
    print(name, opts, {k: ctx.get_option(k) for k in ("march_lds_slots", "march_hbm_brick_slots", "march_hbm_brick_bytes", "march_threads_used")})
    tb.close(); ctx.close()
