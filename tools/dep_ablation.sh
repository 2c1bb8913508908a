#!/bin/bash
# Ablation of k_deposit_tiles at pass 0: builds variants of the library with parts of the kernel compiled
# out (the results are wrong, the timing of the FIRST launch of a render is comparable: same state, same items) and prints
# the kernel's time.  usage: here `bash tools/dep_ablation.sh build`, then on the GPU box `bash tools/dep_ablation.sh run`
set -e
cd "$(dirname "$0")/.."
V="whole: easy:-DDEP_EXP_SKIP_EASY full:-DDEP_EXP_SKIP_FULL loops:-DDEP_EXP_SKIP_EASY,-DDEP_EXP_SKIP_FULL,-DDEP_EXP_SKIP_NEAR"
V="$V stage_noclass:-DDEP_EXP_SKIP_EASY,-DDEP_EXP_SKIP_FULL,-DDEP_EXP_SKIP_NEAR,-DDEP_EXP_NOCLASS,-DDEP_EXP_NOWRITE"
V="$V stage_2loads:-DDEP_EXP_SKIP_EASY,-DDEP_EXP_SKIP_FULL,-DDEP_EXP_SKIP_NEAR,-DDEP_EXP_NOCLASS,-DDEP_EXP_NOWRITE,-DDEP_EXP_LOAD2"
V="$V stage_noloads:-DDEP_EXP_SKIP_EASY,-DDEP_EXP_SKIP_FULL,-DDEP_EXP_SKIP_NEAR,-DDEP_EXP_NOCLASS,-DDEP_EXP_NOWRITE,-DDEP_EXP_NOLOAD"
case "$1" in
build)
  mkdir -p build
  for v in $V; do
    n=${v%%:*}; f=$(echo "${v#*:}" | tr ',' ' ')
    (cd raytracer_amd/csrc && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-result -Wno-unused-value $f -o ../../build/libsppm_abl_$n.so sppm_kernels.hip) &
  done
  wait ;;
run)
  for v in $V; do n=${v%%:*}; bash tools/ab.sh abl_$n "SPPM_HIP_LIB=build/libsppm_abl_$n.so" --steps 1 --warmup 3; done ;;
esac
