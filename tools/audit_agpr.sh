#!/bin/bash
# usage: tools/audit_agpr.sh file.s — list compiler-generated (outside ;;#ASMSTART..;;#ASMEND) uses of AGPRs below a192
awk '/#ASMSTART/{inasm=1} /#ASMEND/{inasm=0} { if(!inasm) { line=$0; sub(/;.*/,"",line); n=split(line, tok, /[ ,\t]+/); for(i=1;i<=n;i++){ t=tok[i]; if (t ~ /^a\[?[0-9]+/) { r=t; gsub(/[^0-9:]/,"",r); split(r, rr, ":"); if (rr[1]+0 < 192) { print NR": "$0; break } } } } }' "$1"
