#!/bin/bash
# Round 3, final build: full GPU suite, default bench line, per-shape dump, rocprofv3 kernel stats
TAG=${1:-r3t}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -14 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
MD_BENCH_DUMP=$O/shapes_all.txt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae > $O/bench_dump.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof
python - <<PY
import json
for f in ("bench_default.json","bench_dump.json","prof_bench.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), "ms/step", round(d["ms_per_step"],1), {k:round(v["ms_per_clip"]) for k,v in d["kernel_families"].items()}, d.get("e2e_frames_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
