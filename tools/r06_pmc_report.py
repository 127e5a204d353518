"""gpurun_out/<tag>/p*/r_counter_collection.csv (tools/pmc.sh passes over `bench.py --config 1 --steps 1 --warmup 1`) -> profiles-ready
r06_pmc.json: what bench.py cannot measure itself (it does not run under rocprofv3 --pmc) and reports from this file when the workload
matches — HBM-side traffic of one vision-tower forward (roofline.traffic) and the marcher's counters (march.pmc).

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3): every counter group in its own pass; FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE
reports HALF the bytes of wide (16 B per lane) coalesced reads — global_load and LDS-DMA alike — so it is DOUBLED for the GEMM /
attention kernels, whose reads are of that kind, and left uncorrected for k_march (4- and 8-byte gathers: uncalibrated);
WRITE_SIZE is uncalibrated and taken as is.  Usage: python tools/r06_pmc_report.py <gpurun_out/tag> <launches per forward = chunks>"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "r_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))


def mean(k, c):
    v = agg[k].get(c)
    return sum(v) / len(v) if v else None


VIT = ("k_gemm", "k_attention", "k_rowstats", "k_embed_ln", "k_head", "k_bcast", "k_gather", "k_scatter", "k_touch_list", "k_preprocess")
# per forward: the counters of the LAST n dispatches of a kernel would need ordering; the passes run the same program, so a kernel's SUM
# over the run divided by the number of forwards in it (warmup + steps + the setup forward of one frame, which is negligible) is used
n_forwards = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
vit_fetch = vit_write = 0.0
per_kernel = {}
for k, cs in agg.items():
    if not k.startswith(VIT):
        continue
    fz, wz = sum(cs.get("FETCH_SIZE", [])), sum(cs.get("WRITE_SIZE", []))
    wide = k.startswith(("k_gemm", "k_attention"))
    vit_fetch += fz * 1024 * (2 if wide else 1)
    vit_write += wz * 1024
    per_kernel[k[:48]] = {"dispatches": len(cs.get("FETCH_SIZE", [])), "fetch_GB": round(fz * 1024 * (2 if wide else 1) / 1e9, 3), "write_GB": round(wz * 1024 / 1e9, 3)}
mk = max((k for k in agg if k.startswith("k_march<true")), key=lambda k: sum(agg[k].get("SQ_INSTS_VALU", [0])), default=None)
march = {}
if mk:
    g = lambda c: mean(mk, c)
    march = {"kernel": mk,
             "mfma_busy": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), 4) if g("SQ_BUSY_CU_CYCLES") else None,
             "valu_busy": round(g("SQ_INSTS_VALU") / g("SQ_BUSY_CU_CYCLES"), 4) if g("SQ_BUSY_CU_CYCLES") and g("SQ_INSTS_VALU") else None,
             "l2_hit": round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4) if g("TCC_HIT_sum") else None,
             "issue_share_of_wave_cycles": round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
             "wait_share_of_wave_cycles": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
             "valu_insts_per_wave": round(g("SQ_INSTS_VALU") / g("SQ_WAVES")) if g("SQ_WAVES") else None,
             "fetch_bytes_per_launch": int(g("FETCH_SIZE") * 1024) if g("FETCH_SIZE") else None,
             "write_bytes_per_launch": int(g("WRITE_SIZE") * 1024) if g("WRITE_SIZE") else None,
             "how": "means per full-size k_march launch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); valu_busy = SQ_INSTS_VALU x 4 cycles / "
                    "(4 SIMDs x SQ_BUSY_CU_CYCLES) (wave64 VALU instructions, MFMA included, at four cycles each: a lower bound — transcendentals take sixteen); "
                    "l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); FETCH / WRITE_SIZE uncorrected (4- and 8-byte gathers)"}
out = {"workload": {"scene": "shopping", "width": 640, "height": 360, "chunk": 4096, "clip": "vit_b16"},
       "command": "tools/pmc.sh r06/pmc <one counter group per pass> -- --config 1 --steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 --product-steps 0",
       "vit_traffic_bytes_per_forward": int((vit_fetch + vit_write) / n_forwards), "vit_fetch_bytes_per_forward": int(vit_fetch / n_forwards),
       "vit_write_bytes_per_forward": int(vit_write / n_forwards), "forwards_in_the_run": n_forwards,
       "vit_how": "sum over the vision tower's kernels of 2 x FETCH_SIZE (GEMM / attention: wide coalesced reads, the guide's gfx950 correction) or 1 x FETCH_SIZE (the "
                  "small kernels) + WRITE_SIZE (uncalibrated), KB -> bytes, divided by the forwards in the run",
       "vit_per_kernel": per_kernel, "march": march}
json.dump(out, open(os.path.join(os.path.dirname(src.rstrip("/")), "r06_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "vit_per_kernel"}, indent=1))
