#!/bin/bash
# Round 4: wave-state / instruction / traffic counters of the marcher, each group in its own pass (rocprofv3 --pmc with
# --kernel-trace only), on configs[1] (k_march<true,5,2,5,false>) and on configs[4]'s cone-stepped instantiation.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B="--steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0"
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
PMC_PASS_TIMEOUT=300 bash $ROOT/tools/pmc.sh r04_pmc_cfg1 "${PASSES[@]}" -- $B > $ROOT/gpurun_out/r04_pmc_cfg1.log 2>&1
PMC_PASS_TIMEOUT=400 bash $ROOT/tools/pmc.sh r04_pmc_cfg4 "${PASSES[@]}" -- $B --config 4 --slice-of 64 > $ROOT/gpurun_out/r04_pmc_cfg4.log 2>&1
cd $ROOT
python - <<'PY'
import re, json
def section(path, key):
    txt = open(path).read()
    m = re.search(r"## (void )?" + re.escape(key) + r".*?\n\n\| counter.*?\n\|---.*?\n(.*?)\n\n", txt, re.S)
    rows = {}
    if m:
        for line in m.group(2).splitlines():
            c = [x.strip() for x in line.strip("|").split("|")]
            rows[c[0]] = (int(c[1]), float(c[2]))
    return rows
out = ["# Marcher counters, round 4 (tools/r04_pmc_march.sh: rocprofv3 --kernel-trace --pmc, one pass per group; per-dispatch means)\n"]
for tag, key, what in (("r04_pmc_cfg1", "k_march<true, 5, 2, 5, false>", "configs[1]: shopping, 4096 candidates, 640x360 (LDS bricks 5 slots, HBM bricks 2 slots)"),
                       ("r04_pmc_cfg4", "k_march<true, 0, 0, 5, true>", "configs[4] slice: shelf (aabb_scale 2, cone stepping), 4096 candidates, generic tables")):
    r = section(f"gpurun_out/{tag}_pmc.md", key)
    if not r:
        r = {}
        txt = open(f"gpurun_out/{tag}_pmc.md").read()
        for m in re.finditer(r"## (void )?(k_march[^\n]*)", txt):
            key = m.group(2).strip()
            r = section(f"gpurun_out/{tag}_pmc.md", key)
            if r: break
    out.append(f"\n## {key} — {what}\n\n| counter | dispatches | mean per launch |\n|---|---|---|")
    for c, (n, v) in sorted(r.items()):
        out.append(f"| {c} | {n} | {v:.6g} |")
    g = lambda k: r.get(k, (0, 0.0))[1]
    if g("SQ_WAVE_CYCLES"):
        out.append(f"\nderived: waves issuing {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} of their cycles (VALU {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f}), "
                   f"waiting on an instruction {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}, any wait {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}; "
                   f"VALU instructions per wave {g('SQ_INSTS_VALU') / max(1, g('SQ_WAVES')):.0f}, LDS {g('SQ_INSTS_LDS') / max(1, g('SQ_WAVES')):.0f}, VMEM reads {g('SQ_INSTS_VMEM_RD') / max(1, g('SQ_WAVES')):.0f}; "
                   f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1, 4 * g('SQ_BUSY_CU_CYCLES')):.3f}; TA busy {g('TA_TA_BUSY_sum') / max(1, g('GRBM_GUI_ACTIVE')) / 256:.3f} per CU; "
                   f"FETCH_SIZE {g('FETCH_SIZE') / 1e6:.2f} GB + WRITE_SIZE {g('WRITE_SIZE') / 1e6:.2f} GB per launch (KB counters, uncorrected)")
    if tag == "r04_pmc_cfg1" and g("FETCH_SIZE"):
        json.dump({"kernel": "k_march<true,5,2,5,false>", "workload": {"scene": "shopping", "width": 640, "height": 360, "chunk": 4096, "clip": "vit_b16"},
                   "how": "FETCH_SIZE + WRITE_SIZE in their own passes, uncorrected (4- and 8-byte gathers)",
                   "source": "tools/r04_pmc_march.sh, the round-4 kernel (packed weight products, packed ReLU)",
                   "fetch_size_kb_per_launch": g("FETCH_SIZE"), "write_size_kb_per_launch": g("WRITE_SIZE"),
                   "traffic_bytes_per_launch": int((g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)}, open("gpurun_out/r04_march_traffic.json", "w"), indent=1)
open("gpurun_out/r04_pmc_march.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
