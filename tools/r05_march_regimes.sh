#!/bin/bash
# Round 5 (VERDICT r04 next #3a): the marcher OUTSIDE its sweet spot.  One bench line per regime (vit_tiny behind it: the ViT is not
# what is measured), then counter passes (each group its own rocprofv3 --kernel-trace --pmc run) for the regimes that matter:
#   - configs[1] as shipped (5 LDS slots + 2 HBM-brick slots), with bricks off (generic table kernel), and with the LDS slots
#     capped at 4 .. 0 (the HBM bricks take over behind them: what an object too large for five LDS slots gets since round 5)
#   - an object 2.2x and 5x the apple's size (shopping_big / shopping_huge)
#   - configs[2] (pool_triangle, BASELINE's "hash-grid HBM-bound stress") and configs[4]'s cone-stepped shelf slice
# -> gpurun_out/r05_march_regimes.{jsonl,md}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r05_march_regimes${R05_SUFFIX}.jsonl
: > $OUT
B="--steps 3 --warmup 1 --cpu-sample 0 --power-seconds 0 --clip vit_tiny"
run() { tag=$1; shift; if [ -n "$R05_ONLY" ] && ! echo " $R05_ONLY " | grep -q " $tag "; then return; fi; echo "== $tag: $*" >&2; line=$(timeout 600 python bench.py $B "$@" 2>/dev/null | tail -1); [ -z "$line" ] && line=null; echo "{\"regime\": \"$tag\", \"args\": \"$*\", \"bench\": $line}" >> $OUT; }
run cfg1_default
run cfg1_bricks_off --opt bricks=0
for n in 4 3 2 1 0; do run cfg1_lds$n --opt lds_slots_max=$n; done
run cfg1_lds5_hbm0 --opt gbrick_slots=0
run cfg1_lds0_total6 --opt lds_slots_max=0 --opt brick_slots_total=6
run big_default --scene shopping_big --sample-res 32,32,1,1,1,1
run big_bricks_off --scene shopping_big --sample-res 32,32,1,1,1,1 --opt bricks=0
run big_total8 --scene shopping_big --sample-res 32,32,1,1,1,1 --opt brick_slots_total=8 --opt gbrick_max_mib=512
run huge_default --scene shopping_huge --sample-res 16,16,1,1,1,1
run huge_bricks_off --scene shopping_huge --sample-res 16,16,1,1,1,1 --opt bricks=0
run huge_cap512 --scene shopping_huge --sample-res 16,16,1,1,1,1 --opt gbrick_max_mib=512
run huge_cap16 --scene shopping_huge --sample-res 16,16,1,1,1,1 --opt gbrick_max_mib=16
run cfg2_default --config 2 --sample-res 64,64,1,1,1,1
run cfg2_bricks_off --config 2 --sample-res 64,64,1,1,1,1 --opt bricks=0
run cfg4_default --config 4 --slice-of 64
run cfg4_bricks_off --config 4 --slice-of 64 --opt bricks=0
run cfg4_lds_only --config 4 --slice-of 64 --opt gbrick_slots=0
# counters: FETCH_SIZE / WRITE_SIZE each in its own pass, TA busy, L2 hit / miss, wave state
PASSES=("FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES")
P="--steps 1 --warmup 1 --cpu-sample 0 --power-seconds 0 --clip vit_tiny"
pmc() { tag=$1; shift; if [ -n "$R05_ONLY" ] && ! echo " $R05_ONLY " | grep -q " $tag "; then return; fi; PMC_PASS_TIMEOUT=240 bash tools/pmc.sh r05_pmc_$tag "${PASSES[@]}" -- $P "$@" > gpurun_out/r05_pmc_$tag.log 2>&1; }
if [ "${R05_PMC:-1}" = "1" ]; then
  pmc cfg1_default
  pmc cfg1_bricks_off --opt bricks=0
  pmc cfg1_lds0 --opt lds_slots_max=0
  pmc big_default --scene shopping_big --sample-res 32,32,1,1,1,1
  pmc huge_default --scene shopping_huge --sample-res 16,16,1,1,1,1
  pmc cfg2_default --config 2 --sample-res 64,64,1,1,1,1
  pmc cfg4_default --config 4 --slice-of 64
fi
python tools/r05_march_regimes_report.py $R05_SUFFIX
