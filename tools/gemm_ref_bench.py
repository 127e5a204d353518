"""Reference point for the ViT GEMM shapes: what the vendor library (hipBLASLt through torch) reaches on
the same shapes on this box, and the shader clock the chip sustains under that load.  Development tool:
nothing in the product uses torch for compute."""
import subprocess
import sys
import threading
import time

import torch

M = 4096 * 197
shapes = {"qkv": (M, 2304, 768), "out": (M, 768, 768), "fc1": (M, 3072, 768), "fc2": (M, 768, 3072)}
dev = torch.device("cuda:0")
clocks = []
stop = False


def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            clocks.append([l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l])
        except Exception as e:      # noqa
            clocks.append([str(e)])
        time.sleep(0.5)


th = threading.Thread(target=poll)
th.start()
for name, (m, n, k) in shapes.items():
    a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
    w = torch.randn((n, k), device=dev, dtype=torch.bfloat16)
    b = torch.randn((n,), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    it = 20
    for _ in range(it):
        c = torch.nn.functional.linear(a, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name}: {m}x{n}x{k}  {ms:.3f} ms  {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s (torch/hipBLASLt bf16, bias epilogue)", flush=True)
    del a, w, b, c
stop = True
th.join()
print("clock/power samples during the run:")
for c in clocks[:: max(1, len(clocks) // 12)]:
    print("  ", c)
