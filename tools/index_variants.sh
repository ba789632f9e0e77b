#!/bin/bash
# development: build the library with different index-kernel shapes ON the GPU box and time each
# usage: tools/index_variants.sh tag "FLAGS1" "FLAGS2" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 600 python3 tools/warm_corpus.py 10000 > gpurun_out/${tag}_warm.log 2>&1 || tail -3 gpurun_out/${tag}_warm.log
k=0
for flags in "$@"; do
	k=$((k+1))
	echo "==== variant $k: $flags"
	touch otezip_amd/csrc/otz_kernels.hip
	make -C otezip_amd/csrc EXTRA="$flags" 2>&1 | grep -E "error|VGPRs|Scratch" | head -2
	timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
export OTEZIP_CORPUS_NO_FORK=1 # (from here on processes run under the profiler: a cache miss generates serially, it does not fork)
	timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$k -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_$k.log 2>&1
	find gpurun_out/prof_${tag}_$k -name "*kernel_stats.csv" | xargs cat | grep "otz_index\|otz_apply" | cut -d, -f1-4
	timeout -k 10 200 python3 tools/prof_phases.py 2000 2>&1 | grep -E "status ok|hdr|tables|walkA|fixpoint|accept|^total|rounds|rewalks" | tail -9
done
