"""Where the time of cb_render/*.png goes on a box: encode threads, target filesystem, zlib level (host only)."""
import os, sys, time, shutil, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import _lib
lib = _lib.load()
n, w, h = 2048, 640, 360
r = np.random.default_rng(0)
base = r.integers(0, 256, (64, h, w, 3), dtype=np.uint8)
base[:, :, :200] = (base[:, :, :200] // 32) * 32          # partly compressible, like a render
frames = np.concatenate([base] * (n // 64))
print("cores", os.cpu_count(), "frames", n, f"{w}x{h}")
for root in ("/tmp/png_probe", "/dev/shm/png_probe"):
    for threads in (8, 16, 32):
        for level in (-1, 1, 0):
            shutil.rmtree(root, ignore_errors=True); os.makedirs(root)
            t = time.time()
            _lib.check(lib.d2r_png_write_batch(_lib.ptr(frames), n, w, h, os.fsencode(root), 0, threads, level))
            dt = time.time() - t
            sz = sum(os.path.getsize(os.path.join(root, f)) for f in os.listdir(root)[:64]) / 64
            t = time.time(); shutil.rmtree(root); dr = time.time() - t
            print(f"{root:22s} threads {threads:3d} level {level}: {n / dt:8.0f} files/s ({dt:.2f}s), {sz / 1024:.0f} KiB/file, rmtree {dr:.2f}s", flush=True)
