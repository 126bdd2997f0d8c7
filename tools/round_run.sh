#!/bin/bash
# One GPU-box pass that produces the records of a round: full GPU test suite, the default bench line (with cpu_baseline), the
# configs[2] line, the 2-rank control-flow run (gloo on one GPU) and a rocprofv3 --kernel-trace --stats profile of the bench.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --config 2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
MD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --small --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_small.json 2> $O/bench_2rank.err; echo "2-rank rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof/*.db 2>/dev/null
python - <<PY
import json
for f in ("bench.json","bench_cfg2.json","bench_2rank_gloo_small.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],1), d["config"]["workload"][:40], d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
