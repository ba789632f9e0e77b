#!/bin/bash
# quick GPU check of the Zstandard path: parity tests, then the secondary measurements with a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-zs}
timeout -k 10 600 python3 tools/warm_corpus.py ${2:-4000} > gpurun_out/${tag}_warm.log 2>&1 || tail -3 gpurun_out/${tag}_warm.log
timeout -k 10 400 python -m pytest tests/test_zstd_real.py -x -q -m gpu -k "not full_count" > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
export OTEZIP_CORPUS_NO_FORK=1 # (from here on processes run under the profiler: a cache miss generates serially, it does not fork)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 tools/bench_secondary.py ${2:-4000} > gpurun_out/${tag}_sec.log 2>&1
grep -h "^{" gpurun_out/${tag}_sec.log | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
for k,v in r.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})"
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | xargs cat | grep otz_ | cut -d, -f1-4
