#!/bin/bash
# Build lib_x4ab/libtfa_hip.so = the product library + the x4 timing-only ablations (bf16, D=128 units rebuilt with
# -DTFA_X4_ABLATE, everything else taken from ../build).  Extra -D flags for the x4 kernel via $X4FLAGS.  Output dir via $OUT.
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
OUT=${OUT:-x4ab}
mkdir -p ../build_$OUT ../lib_$OUT
make -j8 EXTRA="-DTFA_X4_ABLATE $X4FLAGS" OBJDIR=../build_$OUT OUTDIR=../lib_$OUT ../build_$OUT/tfa_fwd_inst_bf16_128_c0.o ../build_$OUT/tfa_x4_inst_bf16_128_c0_o16.o 2>&1 | grep -E "error|audit" || true
for f in ../build/*.o; do b=$(basename $f); [ -f ../build_$OUT/$b ] || cp $f ../build_$OUT/$b; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../build_$OUT/*.o -o ../lib_$OUT/libtfa_hip.so
ls -la ../lib_$OUT/libtfa_hip.so
