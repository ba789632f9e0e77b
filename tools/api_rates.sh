#!/bin/bash
# extract rate through zip_open .. zip_fopen_index .. zip_close for a few settings of the reader's knobs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/warm_corpus.py 2000 > /dev/null 2>&1
gcc -O1 -std=c99 -Iinclude tests/c/zip_roundtrip.c -o /tmp/zrt -Lotezip_amd -lotezip_amd -lpthread -Wl,-rpath,$GRAFT_REPO_ROOT/otezip_amd -Wl,-rpath,/opt/rocm/lib
python3 -c "
import sys; sys.path.insert(0,'.')
from otezip_amd import corpus
arc,_=corpus.build_archive(10000,65536); open('/tmp/big.zip','wb').write(arc)"
for cfg in "4 1" "4 2" "8 2" "8 3" "16 2"; do
	set -- $cfg
	echo "threads $1 ahead $2: $(OTEZIP_READ_THREADS=$1 OTEZIP_READ_AHEAD=$2 /tmp/zrt bench-extract /tmp/big.zip 0 6 | awk '/TOTAL/ {printf "%.1f ", $2/$3/1073741824}') GiB/s"
done
