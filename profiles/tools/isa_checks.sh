#!/bin/bash
# Check-in scans of the final ISA of the two files with inline-asm LDS-DMA (run from humannerf_amd/csrc; ~2 min):
#  1. every scalar write of M0 is followed by at least one other instruction before the first global_load_lds (wait state:
#     every piece carries its own s_nop);
#  2. M0 is touched by nothing but those writes (the compiler keeps it live across the pieces of a run);
#  3. no kernel of these files uses scratch except the ones listed as known (fp32-operand chain: 20 B).
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
T=$(mktemp -d)
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -c hnrf_mlp_f16.hip -o $T/a.o -save-temps=obj 2>/dev/null
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c hnrf_mlp_bwd.hip -o $T/b.o -save-temps=obj 2>/dev/null
for S in $T/hnrf_mlp_f16-hip-amdgcn-amd-amdhsa-gfx950.s $T/hnrf_mlp_bwd-hip-amdgcn-amd-amdhsa-gfx950.s; do
  awk -v f=$(basename $S) '
  /^[ \t]*;/ {next} /^[ \t]*$/ {next}
  { if (pend) { n++; if ($1 ~ /^global_load_lds/) { if (n < 2) bad1++; ok++; pend = 0 } }
    if ($1 ~ /^s_/ && $2 ~ /^m0,/) { pend = 1; n = 0; w++ }
    else if ($0 ~ /[^a-z0-9_]m0/ && $1 !~ /^s_/) other++ }
  END { printf "%s: %d M0 writes, %d followed by LDS-DMA; DMA directly behind the write: %d; other M0 uses: %d\n", f, w, ok, bad1, other
        if (bad1 || other) exit 1 }' $S
done
grep -h "ScratchSize" $T/*.s | sort | uniq -c
rm -rf $T
