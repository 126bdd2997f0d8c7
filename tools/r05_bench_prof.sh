#!/bin/bash
# bench.py in full + rocprofv3 kernel stats of the same command on ONE box (round-5 records after the cpu_baseline quota fix)
TAG=${1:-r5bench}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SECONDS=0; MD_BENCH_DUMP=$O/shapes_all.txt timeout 1200 python bench.py > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$? wall ${SECONDS}s"
python - <<PY
import json
d=json.loads(open("$O/bench_cfg1.json").read().strip().splitlines()[-1]); c=d.get("cpu_baseline") or {}
print(round(d["value"],3), "ms/step", round(d["ms_per_step"],1), d.get("e2e_frames_per_s"), d["roofline"].get("frac"), c.get("value"), c.get("cores"), c.get("usable_cores"), c.get("cgroup_quota_cpus"))
print(c.get("sample","")[-600:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof
grep -A3 "dominant kernel" $O/kernel_stats.md | tail -3
