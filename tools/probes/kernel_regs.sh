#!/bin/bash
# Per-kernel register / scratch / LDS usage of a built object (genima_amd/build/*.o): unbundle the gfx950 code object and read its notes.
#   tools/probes/kernel_regs.sh genima_amd/build/gemm.o
set -e
o=$(readlink -f "$1"); t=$(mktemp -d); cd "$t"; cp "$o" x.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | python3 -c "
import sys,re
recs=[]; cur=None
for ln in sys.stdin:
    if re.match(r'\s*- \.', ln):   # a new kernel entry of amdhsa.kernels (keys are sorted: .name comes in the middle)
        cur={}; recs.append(cur)
    m=re.search(r'\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)', ln)
    if m and cur is not None: cur[m.group(1)]=m.group(2)
for r in sorted((r for r in recs if 'vgpr_count' in r), key=lambda r: r.get('name','')):
    print(r.get('vgpr_count','?').rjust(4), r.get('agpr_count','0').rjust(4), 'scratch', r.get('private_segment_fixed_size','0').rjust(5), 'lds', r.get('group_segment_fixed_size','0').rjust(6), r.get('name','?')[:110])
"
rm -rf "$t"
