#!/bin/bash
# Round 3 records on one box: sp main-loop stamps, bench lines (configs 1 with the per-shape dump, 2, 4), rocprofv3 kernel stats of the
# bench, SQ / FETCH / WRITE counters of the GEMM + conv family and of attention / temporal attention / norms (separate --pmc passes).
TAG=${1:-r3q}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
timeout 300 python tools/sp_trace.py > $O/sp_trace.log 2>&1; echo "trace rc=$?"; tail -4 $O/sp_trace.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
MD_BENCH_DUMP=$O/shapes_all.txt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$?"
timeout 600 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-vae > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
timeout 900 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-vae > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python - <<PY
import json
for f in ("bench_cfg1.json","bench_cfg2.json","bench_cfg4.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],1), d["config"]["workload"][:50], d.get("e2e_frames_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof
cd /tmp
export MD_ITERS=3 MD_WARM=1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $O/pmc_sq -o sq -- python $R/tools/bench_kernels.py conv gemm shapes small > $O/pmc_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python $R/tools/bench_kernels.py conv gemm shapes small > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write -o write -- python $R/tools/bench_kernels.py conv gemm shapes small > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $O/pmc_sq2 -o sq -- python $R/tools/bench_kernels.py attn xattn temporal norm > $O/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch2 -o fetch -- python $R/tools/bench_kernels.py attn xattn temporal norm > $O/pmc_fetch2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write2 -o write -- python $R/tools/bench_kernels.py attn xattn temporal norm > $O/pmc_write2.log 2>&1
cd $R
{ python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch --match gemm; python tools/pmc_raw.py $O/pmc_write; } > $O/pmc_gemm.txt 2>&1
{ python tools/pmc_table.py $O/pmc_sq2 $O/pmc_fetch2 --match _kernel; python tools/pmc_raw.py $O/pmc_write2 --match _kernel; } > $O/pmc_other.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_sq2 $O/pmc_fetch2 $O/pmc_write2
head -20 $O/pmc_gemm.txt
