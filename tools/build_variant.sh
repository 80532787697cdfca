#!/bin/bash
# tools/build_variant.sh <name> <-D flags...> : builds bio_amd/variants/lib_<name>.so for tuning runs
name=$1; shift
mkdir -p bio_amd/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -shared \
  -o bio_amd/variants/lib_$name.so bio_amd/csrc/*.hip bio_amd/csrc/host/*.cpp -lz -lpthread -ldl \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A7 "pileup_kernelILb1ELb0ELb0ELb0" | grep -E "VGPRs:|VGPRs Spill|LDS Size" | tr '\n' ' '
echo " <- $name"
