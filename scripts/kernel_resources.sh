#!/bin/bash
# VGPRs / scratch (spills) / LDS of every kernel in the product library: hipcc's own resource remarks, no GPU needed.
#   scripts/kernel_resources.sh [file.hip ...]        (default: every .hip under foundationpose_amd/csrc)
cd "$(dirname "$0")/../foundationpose_amd/csrc" || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(*.hip)
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
    python3 -c "
import re, sys, subprocess
name = None; row = {}
for line in sys.stdin:
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]; row = {}; continue
    m = re.search(r'remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', line)
    if m and name:
        row[m.group(1)] = int(m.group(2))
        if m.group(1).startswith('LDS'):
            print(f\"$f  {name[:90]:90s} vgpr {row.get('VGPRs',0):3d} agpr {row.get('AGPRs',0):3d} scratch {row.get('ScratchSize [bytes/lane]',0):4d} occ {row.get('Occupancy [waves/SIMD]',0)} static_lds {row.get('LDS Size [bytes/block]',0)}\")
"
done
