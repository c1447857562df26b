#!/bin/bash
# development aid: compile the X-drop kernels alone and print the slice kernel's registers, spills and loop mix.  usage: xs_probe.sh [-D...]
mkdir -p /tmp/isa && cd /tmp/isa && printf '#include <hip/hip_runtime.h>\n#include <cstdint>\n#include "bella_hip.h"\n#include "core.hpp"\n#include "util.hpp"\n#include "xdrop_packed.hpp"\n' > xs.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" -I/root/repo/include -I/root/repo/bella_amd/csrc --cuda-device-only -S -o xs.s xs.hip 2>&1 | grep -E "error" -A4
grep -E "k_xdrop_slice\w*\.(num_vgpr|private_seg_size)" xs.s
python3 /root/repo/tools/dev/isa_loop.py xs.s k_xdrop_slice | sed -n 1,3p
python3 - <<'PY'
import re
lines=open('/tmp/isa/xs.s').read().split('\n')
start=next(i for i,l in enumerate(lines) if re.match(r'^_ZN\w*k_xdrop_slice\w*:',l))
end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
body=lines[start:end]
# the latch: instructions between the exit branch and the back edge
for i,l in enumerate(body):
    if re.search(r's_branch\s+\.LBB\d+_\d+',l):
        # count v_mov just before
        j=i-1; n=0
        while j>0 and (body[j].strip().startswith('v_mov') or body[j].strip().startswith('s_mov') or not body[j].strip() or body[j].strip().startswith(';')):
            if body[j].strip().startswith('v_mov'): n+=1
            j-=1
        if n>4: print('v_mov before back edge/branch at line',i,':',n)
PY
