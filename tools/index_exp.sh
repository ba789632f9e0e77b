#!/bin/bash
# development: walk-A timing experiments of the index kernel (builds that do not decode correctly): phase cycles of
# the index kernel alone (OTEZIP_DEBUG_INDEX_ONLY=1) at full and at low occupancy
# usage: tools/index_exp.sh tag "FLAGS1" "FLAGS2" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 600 python3 tools/warm_corpus.py 10000 > gpurun_out/${tag}_warm.log 2>&1 || tail -3 gpurun_out/${tag}_warm.log
export OTEZIP_DEBUG_INDEX_ONLY=1
k=0
for flags in "$@"; do
	k=$((k+1))
	echo "==== variant $k: $flags"
	touch otezip_amd/csrc/otz_kernels.hip
	make -C otezip_amd/csrc EXTRA="$flags" 2>&1 | grep -E "error|VGPRs|Scratch" | head -2
	for n in 10000 2000; do
		echo "-- $n entries"
		timeout -k 10 200 python3 tools/prof_phases.py $n 2>&1 | grep -E "hdr|tables|slots|walkA|fixpoint|accept|^total|rewalks|fix_iters|steps|per entry|prologue" | tail -14
	done
done
