cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/warm_corpus.py 2000 > /dev/null 2>&1
gcc -O1 -std=c99 -Iinclude tests/c/zip_roundtrip.c -o /tmp/zrt -Lotezip_amd -lotezip_amd -lpthread -Wl,-rpath,$GRAFT_REPO_ROOT/otezip_amd -Wl,-rpath,/opt/rocm/lib
python3 -c "
import sys; sys.path.insert(0,'.')
from otezip_amd import corpus
arc,_=corpus.build_archive(10000,65536); open('/tmp/big.zip','wb').write(arc)"
OTEZIP_TRACE=1 OTEZIP_READ_THREADS=8 /tmp/zrt bench-extract /tmp/big.zip 0 3 2>&1 | grep -v "^[0-9]* [0-9a-f]* " | tail -120 > gpurun_out/api_trace.txt
tail -75 gpurun_out/api_trace.txt
