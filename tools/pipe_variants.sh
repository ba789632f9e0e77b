#!/bin/bash
# development: time the benchmark under different launch shapes of a large DEFLATE batch (one box, one build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tools/warm_corpus.py 2000 > gpurun_out/pipe_warm.log 2>&1
run() { echo -n "$1: "; env $1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for v in "$@"; do run "$v"; done
