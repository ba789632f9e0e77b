#!/bin/bash
# where the time of a warm zip_file_add .. zip_close pass goes (development): OTEZIP_TRACE of the last of three passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
gcc -O1 -std=c99 -Iinclude tests/c/zip_roundtrip.c -o /tmp/zrt -Lotezip_amd -lotezip_amd -lpthread -Wl,-rpath,$GRAFT_REPO_ROOT/otezip_amd -Wl,-rpath,/opt/rocm/lib
python3 -c "
import sys; sys.path.insert(0,'.')
from otezip_amd import corpus
bufs=corpus.text_buffers(10000,65536,first=50000); open('/tmp/add_in.bin','wb').write(b''.join(bufs))"
OTEZIP_TRACE=1 /tmp/zrt bench-add /tmp/add_out.zip /tmp/add_in.bin 10000 65536 3 > gpurun_out/add_trace.txt 2>&1
grep -n "TOTAL" gpurun_out/add_trace.txt
awk '/TOTAL/{n++} n==2' gpurun_out/add_trace.txt | grep -v TOTAL | cut -c1-120 | head -60
