#!/bin/bash
# Rebuild only the named units of libtfa_hip.so (default: the four D=128 forward units + tfa_api) with the Makefile's flags and audit, relink with the
# objects already in build/.  For kernel-header edits that cannot change the other units (x4 / backward / 64-wide): `make` would rebuild all of them.
# usage: tools/r4_quick.sh [unit ...]      e.g. tools/r4_quick.sh tfa_fwd_inst_bf16_128_c1 tfa_api
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
UNITS=${@:-"tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0 tfa_fwd_inst_f16_128_c1 tfa_fwd_inst_f16_128_c0 tfa_api"}
for u in $UNITS; do
  touch $u.hip
  make -o tfa_fwd_kernel_il.h ../build/$u.o > /tmp/r4_quick_$u.log 2>&1 &
done
wait
for u in $UNITS; do tail -2 /tmp/r4_quick_$u.log | grep -v warning || true; done
for u in $UNITS; do [ -f ../build/$u.o ] || { echo "BUILD FAILED: $u (see /tmp/r4_quick_$u.log)"; tail -5 /tmp/r4_quick_$u.log; exit 1; }; done
objs=$(ls ../build/*.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/libtfa_hip.so
# mark everything else up to date so that a later plain `make` does not rebuild the untouched units
touch ../build/*.o ../lib/libtfa_hip.so
echo "relinked ../lib/libtfa_hip.so"
