#!/bin/bash
# quick GPU check used during development: parity tests, per-kernel stats of bench.py, phase breakdown
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-q}
timeout -k 10 600 python3 tools/warm_corpus.py 2000 > gpurun_out/${tag}_warm.log 2>&1 || tail -3 gpurun_out/${tag}_warm.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
export OTEZIP_CORPUS_NO_FORK=1 # (from here on processes run under the profiler: a cache miss generates serially, it does not fork)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1
tail -1 gpurun_out/${tag}_bench.log | cut -c1-330
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | xargs cat | grep otz_ | cut -d, -f1-4
timeout -k 10 200 python3 tools/prof_phases.py 2000 > gpurun_out/${tag}_phases.txt 2>&1; grep -v "  0 cyc\|amdgpu.ids" gpurun_out/${tag}_phases.txt
