"""gpurun_out/r05_march_regimes.jsonl (+ gpurun_out/r05_pmc_*_pmc.md) -> gpurun_out/r05_march_regimes.md: one row per regime of the marcher
(tools/r05_march_regimes.sh).  Product path only."""
import json
import os
import re
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
SUFFIX = sys.argv[1] if len(sys.argv) > 1 else ""


def pmc_means(tag):
    """counter -> mean per launch of the k_march kernel in gpurun_out/r05_pmc_<tag>_pmc.md"""
    path = os.path.join(ROOT, "gpurun_out", f"r05_pmc_{tag}_pmc.md")
    if not os.path.exists(path):
        return {}
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"## (?:void )?(k_march[^\n]*)\n\n\| counter.*?\n\|---.*?\n(.*?)\n\n", txt, re.S):
        for line in m.group(2).splitlines():
            c = [x.strip() for x in line.strip("|").split("|")]
            out[c[0]] = float(c[2])
        out["kernel"] = m.group(1).strip()
        break
    return out


def main():
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", f"r05_march_regimes{SUFFIX}.jsonl")) if l.strip()]
    md = ["# The marcher's regimes, round 5 (`tools/r05_march_regimes.sh`, one MI355X; vit_tiny behind the render: only `k_march` is read)", "",
          "`frac` = samples x 512 B / launch time / 8 TB/s (SURVEY.md 8(d): algorithmic bytes); `traffic` = FETCH_SIZE + WRITE_SIZE of the `k_march` launch, each counter in its own",
          "rocprofv3 pass, KB -> bytes, UNCORRECTED (4- and 8-byte gathers: the guide's x2 applies to wide coalesced reads only) -> `traffic_frac` = traffic / launch time / 8 TB/s = the",
          "HBM-side GB/s the hash lookup draws.  TA busy = TA_TA_BUSY_sum / GRBM_GUI_ACTIVE / 256 CUs; L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS); issue / wait = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY",
          "over SQ_WAVE_CYCLES; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES).", "",
          "| regime | args | bricks (LDS, HBM) | candidates | samples / launch | ms / launch | frac (algorithmic) | lane util. | traffic GB | traffic_frac | TA busy | L2 hit | issue / wait | MFMA busy |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        b = r["bench"]
        if not b:
            md.append(f"| {r['regime']} | `{r['args']}` | bench failed | | | | | | | | | | | |")
            continue
        rf = b["roofline"]
        pm = pmc_means(r["regime"])
        ms = rf["avg_launch_ms"]
        tr = (pm.get("FETCH_SIZE", 0) + pm.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm else None
        cell = lambda v, f: (f % v) if v is not None else ""
        ta = pm["TA_TA_BUSY_sum"] / pm["GRBM_GUI_ACTIVE"] / 256 if pm.get("GRBM_GUI_ACTIVE") else None
        l2 = pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"]) if pm.get("TCC_HIT_sum") else None
        iw = f"{pm['SQ_ACTIVE_INST_ANY'] / pm['SQ_WAVE_CYCLES']:.2f} / {pm['SQ_WAIT_INST_ANY'] / pm['SQ_WAVE_CYCLES']:.2f}" if pm.get("SQ_WAVE_CYCLES") else ""
        mf = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * pm["SQ_BUSY_CU_CYCLES"]) if pm.get("SQ_BUSY_CU_CYCLES") else None
        bc = rf.get("brick_config", {})
        md.append(f"| {r['regime']} | `{r['args']}` | {bc.get('lds_slots')}, {bc.get('hbm_brick_slots')} | {b['config']['poses_per_step']} | {rf['samples_per_launch']:.3g} | {ms:.3f} | "
                  f"**{rf['frac']:.3f}** | {rf['lane_utilisation']:.2f} | {cell(tr / 1e9 if tr else None, '%.2f')} | {cell(tr / (ms * 1e-3) / 8e12 if tr else None, '%.3f')} | "
                  f"{cell(ta, '%.3f')} | {cell(l2, '%.2f')} | {iw} | {cell(mf, '%.2f')} |")
    open(os.path.join(ROOT, "gpurun_out", f"r05_march_regimes{SUFFIX}.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
