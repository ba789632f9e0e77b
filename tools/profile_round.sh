#!/bin/bash
# Everything profiles/<round>/ is built from, in one run on the GPU box (each rocprofv3 pass is its
# own process; --pmc passes use --kernel-trace only; at most four SQ counters per pass):
#   tools/profile_round.sh r02     ->  gpurun_out/profile_r02/...   (then copy into profiles/r02/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-rXX}
out=gpurun_out/profile_${tag}
mkdir -p $out
echo "[profile] corpora (unprofiled: generation forks, a profiled process must not)"
timeout -k 10 600 python3 tools/warm_corpus.py 10000 > $out/warm.log 2>&1 || tail -3 $out/warm.log
echo "[profile] bench line"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err || tail -3 $out/bench.err
cut -c1-300 $out/bench.json
echo "[profile] kernel trace of bench.py"
export OTEZIP_CORPUS_NO_FORK=1 # (from here on processes run under the profiler: a cache miss generates serially, it does not fork)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/trace_bench.log 2>&1
find $out/trace_bench -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
grep otz_ $out/bench_kernel_stats.csv | cut -d, -f1-4
echo "[profile] kernel trace of the secondary workloads"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_secondary -- python3 tools/bench_secondary.py 10000 > $out/secondary.log 2>&1
find $out/trace_secondary -name "*kernel_stats.csv" -exec cp {} $out/secondary_kernel_stats.csv \;
grep otz_ $out/secondary_kernel_stats.csv | cut -d, -f1-4
grep -h "^{" $out/secondary.log | tail -1 > $out/secondary_measurements.json
echo "[profile] HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
	timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || echo "[profile] pass $c failed"
done
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_traffic.json > /dev/null && grep -E "traffic_bytes_per_launch|kernel_build_id" $out/pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do find $out/pmc_$c -name "*counter_collection.csv" -exec sh -c 'grep -E "Counter_Name|otz_" "$1" > "$2"' _ {} $out/pmc_${c}_otz.csv \; ; done
echo "[profile] SQ counters"
tools/sq_counters.sh ${tag} > $out/sq.log 2>&1; cp gpurun_out/sq_summary_${tag}.json $out/sq_summary.json 2>/dev/null; grep -c SQ_ $out/sq_summary.json
echo "[profile] phases"
timeout -k 10 200 python3 tools/prof_phases.py 4000 > $out/phases.txt 2>&1; grep -v "amdgpu.ids\|  0 cyc" $out/phases.txt
echo "[profile] Zstandard kernel phases"
timeout -k 10 200 python3 tools/prof_zstd.py 2500 262144 > $out/zstd_phases.txt 2>&1; grep -v "amdgpu.ids" $out/zstd_phases.txt
echo "[profile] encoder: throughput, ratio, phases"
timeout -k 10 200 python3 tools/prof_encode.py 10000 > $out/encode_phases.txt 2>&1; grep -v "amdgpu.ids" $out/encode_phases.txt
echo "[profile] LDS access costs"
tools/ubench/lds_unaligned > $out/lds_access_costs.txt 2>&1 || true
rm -rf $out/trace_bench $out/trace_secondary $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls $out
