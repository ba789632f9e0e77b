cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/warm_corpus.py 10000 > gpurun_out/sec_warm.log 2>&1 || tail -3 gpurun_out/sec_warm.log
timeout -k 10 600 python3 tools/bench_secondary.py 10000 > gpurun_out/sec.log 2>&1; grep -h "^{" gpurun_out/sec.log | tail -1 | python3 -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})"
