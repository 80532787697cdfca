#!/bin/bash
# on the GPU box: time the stitched 30x shape with every bio_amd/variants/lib_*.so (parity not checked)
for f in bio_amd/variants/lib_*.so; do
  r=$(BIO_AMD_LIB=$PWD/$f timeout -k 10 200 python3 tools/config_bench.py c1s 2>&1 | grep -o "kernel [0-9.]* ms.*parity.*")
  echo "$f $r"
done
