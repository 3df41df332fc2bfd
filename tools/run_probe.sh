#!/bin/bash
# binaries: make -C tools (tools/Makefile records each variant's -D flags)
P=tools/bin/igemm2_probe
for v in 0 1; do
$P 131072 256 2048 $v
$P 262144 256 2048 $v
$P 65536 512 2048 $v
$P 131072 256 1024 $v
$P 131072 256 512 $v
done
$P 524288 128 1024 2
$P 524288 128 512 2
