#!/bin/bash
# Round 4 records on one box: bench lines (configs 1 in full with the per-shape dump and the live PMC traffic, 2, 4 with its dump),
# e2e parity at configs[4] geometry, VAE per-shape table, rocprofv3 kernel stats of the bench, SQ / FETCH / WRITE counters of the GEMM + conv
# family, of attention / temporal attention / norms and of the VAE (separate --pmc passes, kernel-trace only).
TAG=${1:-r4rec}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
MD_BENCH_DUMP=$O/shapes_all.txt timeout 1500 python bench.py > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$?"
timeout 600 python bench.py --config 2 --no-cpu-baseline --no-vae --no-pmc > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
MD_BENCH_DUMP=$O/shapes_cfg4.txt timeout 900 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python - <<PY
import json
for f in ("bench_cfg1.json","bench_cfg2.json","bench_cfg4.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],1), d.get("e2e_frames_per_s"), d["roofline"].get("traffic"), d["roofline"].get("frac"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python tests/e2e_parity.py --config4 --out $O/e2e_parity_cfg4.json > $O/e2e_parity_cfg4.log 2>&1; echo "e2e cfg4 rc=$?"; grep -v "^{" $O/e2e_parity_cfg4.log | tail -4
MD_VAE_DUMP=$O/vae_shapes.txt timeout 600 python tools/bench_vae.py > $O/vae.json 2> $O/vae.err; echo "vae rc=$?"; cat $O/vae.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
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
# the write counter twice more on the attention kernel alone: its run-to-run spread is part of the record
for i in 1 2; do timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write_attn$i -o write -- python $R/tools/bench_kernels.py attn > /dev/null 2>&1; done
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch3 -o fetch -- python $R/tools/bench_vae.py > $O/pmc_fetch3.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write3 -o write -- python $R/tools/bench_vae.py > $O/pmc_write3.log 2>&1
cd $R
{ python tools/pmc_table.py $O/pmc_sq $O/pmc_fetch --match gemm; python tools/pmc_raw.py $O/pmc_write; } > $O/pmc_gemm.txt 2>&1
{ python tools/pmc_table.py $O/pmc_sq2 $O/pmc_fetch2 --match _kernel; python tools/pmc_raw.py $O/pmc_write2 --match _kernel; echo "attention WRITE_SIZE, two more collections:"; python tools/pmc_raw.py $O/pmc_write_attn1 --match attn; python tools/pmc_raw.py $O/pmc_write_attn2 --match attn; } > $O/pmc_other.txt 2>&1
{ python tools/pmc_table.py $O/pmc_fetch3 --match _kernel; python tools/pmc_raw.py $O/pmc_write3 --match _kernel; } > $O/pmc_vae.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_sq2 $O/pmc_fetch2 $O/pmc_write2 $O/pmc_fetch3 $O/pmc_write3 $O/pmc_write_attn1 $O/pmc_write_attn2
head -12 $O/pmc_other.txt; head -8 $O/pmc_vae.txt
