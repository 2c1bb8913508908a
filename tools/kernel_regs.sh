#!/bin/bash
# registers, scratch and LDS of every kernel in raytracer_amd/libsppm_hip.so (from the code object's metadata)
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$1" --output=$T/co.elf --unbundle 2>/dev/null || {
  # a shared library: the bundle sits in the .hip_fatbin section
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/co.elf --unbundle
}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.elf | python3 -c '
import sys, re
name = None; row = {}
for line in sys.stdin:
    m = re.match(r"\s+\.(\w+):\s+(.*)", line) or re.match(r"\s+- \.(\w+):\s+(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and not v.startswith("_Z") and not v.startswith("k_") : continue
    if k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count", "vgpr_spill_count"): row[k] = v
    if k == "symbol": row["symbol"] = v
    if k == "wavefront_size":  # the last key of an entry (keys are in alphabetical order)
        print(row.get("symbol", "?").replace(".kd", "")[:70].ljust(70), " ".join(f"{a}={row.get(b)}" for a, b in (("vgpr", "vgpr_count"), ("agpr", "agpr_count"), ("sgpr", "sgpr_count"), ("scratch", "private_segment_fixed_size"), ("lds", "group_segment_fixed_size"), ("spill", "vgpr_spill_count"))))
        row = {}
'
rm -rf $T
