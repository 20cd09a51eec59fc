#!/bin/bash
# Compile ONE instantiation of a kernel of raytracing_amd/csrc/rt_kernels.h for gfx950 and print its register / scratch / spill
# figures (seconds instead of the minute the whole library takes); -S keeps the assembly.  No GPU needed.
# usage: tools/one_kernel.sh 'k_render<0, false, false>' [extra hipcc flags, e.g. -DRT_SB0=4] [-S out.s]
set -e
inst=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
asm=""
args=()
while [ $# -gt 0 ]; do
  if [ "$1" = "-S" ]; then asm=$2; shift 2; else args+=("$1"); shift; fi
done
name=${inst%%<*}
cat > $tmp/one.hip <<EOT
#include "$root/raytracing_amd/csrc/rt_kernels.h"
template __global__ void $inst(
EOT
# the parameter list of the kernel, copied from its definition
python3 - "$root/raytracing_amd/csrc/rt_kernels.h" "$name" >> $tmp/one.hip <<'EOT'
import re, sys
src = open(sys.argv[1]).read()
m = re.search(r"void " + sys.argv[2] + r"\((.*?)\)\s*\{", src, re.S)
params = re.sub(r"\s+", " ", m.group(1))
# drop the parameter names: keep the types
out = []
for p in params.split(","):
    p = p.strip()
    out.append(re.sub(r"\s*\b\w+$", "", p) if not p.endswith("*") else p)
print(", ".join(out) + ");")
EOT
flags="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function --offload-device-only"
/opt/rocm/bin/hipcc $flags "${args[@]}" -Rpass-analysis=kernel-resource-usage -S -o $tmp/one.s $tmp/one.hip 2>&1 | grep -E "VGPRs:|AGPRs|ScratchSize|Occupancy|Spill|LDS Size|TotalSGPRs|error" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | tail -8 | paste -sd' ' -
[ -n "$asm" ] && cp $tmp/one.s "$asm"
rm -rf $tmp
