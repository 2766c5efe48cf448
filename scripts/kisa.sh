#!/bin/bash
# kisa.sh <asm file> <mangled-name substring> : prints the assembly of one kernel (from a -save-temps build) to stdout
awk -v k="$2" 'index($0,k) && /:.*; @/ {p=1} p {print} p && /s_endpgm/ {exit}' "$1"
