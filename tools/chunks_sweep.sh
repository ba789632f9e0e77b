cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tools/warm_corpus.py 2000 > gpurun_out/ch_warm.log 2>&1
for c in 1 2 3 4 0; do
echo "chunks=$c"; OTEZIP_GPU_CHUNKS=$c timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
