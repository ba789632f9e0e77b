#!/bin/bash
# hipcc -save-temps build of the kernels into /tmp/asm (ISA for tools/isa_loops.py); prints resources of one kernel
mkdir -p /tmp/asm && cd /tmp/asm && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fconstexpr-steps=100000000 $EXTRA -I/root/repo/include -c -o /tmp/asm/k.o /root/repo/otezip_amd/csrc/otz_kernels.hip -save-temps -Rpass-analysis=kernel-resource-usage 2> /tmp/asm/res.txt
grep -A8 "Function Name: ${1:-otz_index_kernel} " /tmp/asm/res.txt | grep -E "SGPRs|VGPRs:|Scratch|Occupancy" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '; echo
