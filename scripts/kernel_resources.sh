#!/bin/bash
# Register / spill / LDS table of every resident_fit_kernel instance (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).
# usage: scripts/kernel_resources.sh [pairs...]   (default: all nine (model, method) pairs)   -> stdout
cd "$(dirname "$0")/../brdf_amd/csrc" || exit 1
PAIRS=${@:-00 01 02 10 11 12 20 21 22}
TMP=$(mktemp -d)
for p in $PAIRS; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -DRI_PAIR=$p -Rpass-analysis=kernel-resource-usage \
    --cuda-device-only -c resident_inst.hip -o $TMP/r$p.o 2> $TMP/r$p.txt &
done
wait
echo "model method fast batched | VGPRs AGPRs spilled_VGPRs SGPRs spilled_SGPRs scratch_B/lane LDS_B occupancy_waves/SIMD"
for p in $PAIRS; do
  python3 - $TMP/r$p.txt <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for blk in t.split("Function Name: ")[1:]:
    m = re.match(r"_ZN4brdf19resident_fit_kernelILi(\d)ELi(\d)ELb(\d)ELb(\d)E", blk)
    if not m:
        continue
    g = lambda k: re.search(k + r": (\d+)", blk).group(1)
    print(*m.groups(), "|", g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
          g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]"))
PY
done
rm -rf $TMP
