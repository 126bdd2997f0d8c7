#!/bin/bash
# Round 5 records on one box: the -m gpu suite as the driver runs it + smoke, bench.py in full (per-shape dump, live PMC traffic, all-core
# CPU baseline, VAE), configs 2 / 4, rocprofv3 kernel stats of the bench, FETCH / WRITE / SQ counters of the fused-normalisation kernels and
# their literal counterparts (separate --pmc passes, kernel-trace only).
TAG=${1:-r5rec}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SECONDS=0; timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; RC=$?; echo "pytest rc=$RC wall ${SECONDS}s" | tee $O/pytest.time; tail -22 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
SECONDS=0; MD_BENCH_DUMP=$O/shapes_all.txt timeout 1200 python bench.py > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$? wall ${SECONDS}s"
timeout 500 python bench.py --config 2 --no-cpu-baseline --no-vae --no-pmc > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
MD_BENCH_DUMP=$O/shapes_cfg4.txt timeout 800 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python - <<PY
import json
for f in ("bench_cfg1.json","bench_cfg2.json","bench_cfg4.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); c=d.get("cpu_baseline") or {}
        print(f, round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],1), d.get("e2e_frames_per_s"), d["roofline"].get("traffic"), d["roofline"].get("frac"), c.get("value"), c.get("cores"), c.get("usable_cores"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof
cd /tmp
export MD_ITERS=3 MD_WARM=1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $O/pmc_sq -o sq -- python $R/tools/bench_kernels.py fused skinny > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python $R/tools/bench_kernels.py fused skinny > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write -o write -- python $R/tools/bench_kernels.py fused skinny > $O/pmc_write.log 2>&1
cd $R
{ python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch --match _kernel; python tools/pmc_raw.py $O/pmc_write --match _kernel; } > $O/pmc_fused.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write
head -40 $O/pmc_fused.txt
