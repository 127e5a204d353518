#!/bin/bash
# Round 6: upper bounds of the two attention changes VERDICT r05 asked for, by ablation BEFORE building them (tools/attn_ablate.sh's method: the
# kernel's average duration over 2048 images x 12 heads per launch, 197 tokens): 512 = the partial last query tile computes nothing (what a
# 16x16x32 tail tile could return at most), 1024 = no scale-subtract (accumulator initialised with -m_run, log2(e)/8 folded into W_q), 1536 = both.
MASKS="0 512 1024 1536" exec $(dirname $0)/attn_ablate.sh
