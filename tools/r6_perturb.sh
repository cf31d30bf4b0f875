#!/bin/bash
# Round 6: which physical registers hipcc hands the hand-scheduled statement moves the loop by +-1.5 % (register banks), and unrelated edits move that choice.  Equivalent
# spellings of the statement's scratch operands (-DTFA_IL_PERTURB=n, tfa_fwd_il_tile_loop.inc) built as arm libraries (tools/r5_arm.sh ptN -DTFA_IL_PERTURB=N), measured here in one
# process against the default spelling, the build without the early requests, and the build before them.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARMS="r6a=$L/lib_r6a/libtfa_hip.so:30 noearly=$L/lib_noearly/libtfa_hip.so:30 p0=$L/lib/libtfa_hip.so:30"
for n in 1 2 3 4 5 6 7 8; do [ -f $L/lib_pt$n/libtfa_hip.so ] && ARMS="$ARMS p$n=$L/lib_pt$n/libtfa_hip.so:30"; done
( timeout 600 python tools/ab_multi.py $ARMS --cfgs cfg3,cfg5 --rounds 7 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py $ARMS --cfgs cfg3 --rounds 5 --iters 30 --data zeros ) > gpurun_out/r6_perturb.txt 2>&1
cat gpurun_out/r6_perturb.txt
