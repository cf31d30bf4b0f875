#!/bin/bash
# tools/sched_view.sh FILE.s — condensed per-basic-block schedule string of a gfx950 .s listing
#   M mfma, e v_exp, p v_pk_*, v other VALU, r ds_read, w ds_write, B buffer op, | s_waitcnt, # s_barrier, n s_nop
awk '
/^\.LBB/ {printf "\n%s ", $1; next}
/^[ \t]+v_mfma/ {printf "M"; next}
/^[ \t]+v_exp/ {printf "e"; next}
/^[ \t]+v_pk_/ {printf "p"; next}
/^[ \t]+v_/ {printf "v"; next}
/^[ \t]+ds_read/ {printf "r"; next}
/^[ \t]+ds_write/ {printf "w"; next}
/^[ \t]+buffer_/ {printf "B"; next}
/^[ \t]+s_waitcnt/ {printf "|"; next}
/^[ \t]+s_barrier/ {printf "#"; next}
/^[ \t]+s_nop/ {printf "n"; next}
END {printf "\n"}
' "$1"
