#!/bin/bash
# usage: tools/audit_agpr.sh file.s [limit] — list compiler-generated (outside ;;#ASMSTART..;;#ASMEND) uses of AGPRs below a<limit>
# (default 192: the x4 forward kernel owns a0..a191; the 256-wide backward / split-KV kernels own a0..a127: limit 128)
LIMIT=${2:-192}
awk -v LIM="$LIMIT" '/#ASMSTART/{inasm=1} /#ASMEND/{inasm=0} { if(!inasm) { line=$0; sub(/;.*/,"",line); n=split(line, tok, /[ ,\t]+/); for(i=1;i<=n;i++){ t=tok[i]; if (t ~ /^a\[?[0-9]+/) { r=t; gsub(/[^0-9:]/,"",r); split(r, rr, ":"); if (rr[1]+0 < LIM) { print NR": "$0; break } } } } }' "$1"
