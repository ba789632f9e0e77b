#!/bin/bash
# development: build the library with different apply-kernel shapes ON the GPU box and time each
# usage: tools/apply_variants.sh tag "FLAGS1" "FLAGS2" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 600 python3 tools/warm_corpus.py 2000 > gpurun_out/${tag}_warm.log 2>&1 || tail -3 gpurun_out/${tag}_warm.log
k=0
for flags in "$@"; do
	k=$((k+1))
	echo "==== variant $k: $flags"
	touch otezip_amd/csrc/otz_kernels.hip
	make -C otezip_amd/csrc EXTRA="$flags" 2>&1 | grep -E "error|VGPRs|Scratch" | tail -2
	timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
	timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' 
	timeout -k 10 200 python3 tools/prof_phases.py 2000 2>&1 | grep -E "status ok|tb_|st_|tables|stage|place|flush|resolve|far|B_iters|^total|plain|walkA|fixpoint" | head -20
done
