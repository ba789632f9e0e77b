#!/bin/bash
# SQ counter passes over bench.py (each --pmc set in its own run, kernel-trace only), reduced to
# one JSON by tools/sq_summary.py.  Usage (GPU box): tools/sq_counters.sh <tag> [program args...]
# (default program: bench.py --steps 3 --warmup 1 --no-cpu-baseline; e.g. `tools/sq_counters.sh enc tools/prof_encode.py 10000`)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-sq}
shift
if [ $# -eq 0 ]; then set -- bench.py --steps 3 --warmup 1 --no-cpu-baseline; fi
prog=("$@")
run() { # name counters...
	local name=$1; shift
	echo "[sq] pass ${name}: $@"
export OTEZIP_CORPUS_NO_FORK=1 # (from here on processes run under the profiler: a cache miss generates serially, it does not fork)
	timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 "${prog[@]}" > gpurun_out/pmc_${tag}_${name}.log 2>&1 || { echo "[sq] pass ${name} failed"; tail -3 gpurun_out/pmc_${tag}_${name}.log; exit 1; }
}
# (at most four SQ counters per pass: a seven-counter pass hung on this pool)
run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 tools/sq_summary.py gpurun_out ${tag} > gpurun_out/sq_summary_${tag}.json; cat gpurun_out/sq_summary_${tag}.json
