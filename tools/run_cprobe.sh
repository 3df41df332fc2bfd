#!/bin/bash
# binaries: make -C tools (tools/Makefile records each variant's -D flags)
for v in base nobar nodma same novm; do
  echo "== $v (1 WG/CU: +40000 B LDS)"
  tools/bin/cp_$v 512 256 16 128 128 40000
  tools/bin/cp_$v 512 512 8 256 256 0
done
echo "== base, 2 WG/CU"
tools/bin/cp_base 512 256 16 128 128 0
