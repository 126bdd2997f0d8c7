#!/bin/bash
# Round 6: ONE parameterised script for every GPU call.   gpurun -- 'bash tools/r06_gpu.sh TAG step [step ...]'
# Each step writes under gpurun_out/TAG/ and is bounded by its own timeout.  Steps:
#   suite         the -m gpu suite as the driver runs it (+ smoke), wall time recorded
#   budget        record the parity budgets (tests/parity_budget.py) of the loop / UNet parity tests into gpurun_out/TAG/parity_budget.json
#   degraded      the same tests on the two deliberately degraded builds (tools/ab/lib_degrade_{p8,gelu}.so) against the recorded budgets
#   e2e           tests/e2e_parity.py at the headline configuration (f = 16, 20 steps, 96 x 96) -> e2e_parity.json
#   bench         bench.py in full (per-shape dump) -> bench_cfg1.json ; bench2 / bench4: configs 2 / 4
#   prof          rocprofv3 --kernel-trace --stats of bench.py --steps 1 -> kernel_stats.md
#   cumask        tools/ubench/cu_mask_map.hip (mask bit -> CU calibration) + tools/cu_partition.py  (LAST: masked queues are new ground)
#   kern ARGS     tools/bench_kernels.py with MD_KERN="gemm shapes ..." (30 iterations)
#   trace         tools/sp_trace.py (needs tools/ab/lib_trace.so) on the shapes in MD_TRACE
#   ab            same-box end-to-end A/B of the libraries named in MD_AB="base cand" (tools/ab/lib_*.so), two rounds, family table
#   kernab                     tools/bench_kernels.py $MD_KERN under each library of MD_AB
#   abenv / abflag / kernenv   same-box A/B of one environment knob (MD_AB_ENV, MD_AB_VALUES) or one bench.py flag (MD_AB_FLAG) in ONE library; the knob on the micro-benchmarks
#   rccl1                      one-rank job with a forced process group: every collective of dp.py through RCCL on one GPU
#   ranks2 / queues            2-rank runs of bench.py on the one GPU (gloo); the two-queue proxy measurement
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SUBSET="tests/test_unets_gpu.py tests/test_e2e_parity_gpu.py tests/test_full_size_gpu.py"
summ() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d.get("cpu_baseline") or {}; fam = d.get("kernel_families", {})
        print(f.split("/")[-1], "%.3f f/s" % d["value"], "%.1f ms" % d["ms_per_step"], "e2e", d.get("e2e_frames_per_s_by_config") or d.get("e2e_frames_per_s"),
              "frac", round(d["roofline"].get("frac", 0), 4), "cpu", c.get("value"), c.get("cores"), c.get("frames_linearity"),
              " ".join("%s %.0f" % (k, v["ms_per_clip"]) for k, v in list(fam.items())[:8]))
    except Exception as e:
        print(f, "ERR", e)
PY
}
for step in "$@"; do
case $step in
suite)
  SECONDS=0; timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s" | tee $O/pytest.time; tail -20 $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log ;;
budget)
  rm -f $O/parity_budget.json
  MD_PARITY_RECORD=$O/parity_budget.json timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_budget.log 2>&1; echo "budget rc=$?"; tail -3 $O/pytest_budget.log
  grep PARITY_MEASURE $O/pytest_budget.log | sort | uniq > $O/parity_measured.txt; cat $O/parity_measured.txt
  python -c "import json; d=json.load(open('$O/parity_budget.json')); print(len(d['checks']), 'loop/UNet budgets,', len(d['kernel_checks']), 'kernel budgets')" ;;
degraded)
  # the recorded budget becomes the suite's budget for this step (on the box only), then each degraded library takes the product's place
  cp $O/parity_budget.json tests/golden/parity_budget.json
  cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
  for v in degrade_p8 degrade_gelu; do
    cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
    timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_$v.log 2>&1; echo "== $v rc=$? (non-zero = the budgets caught it)"
    grep -E "^FAILED|passed|failed" $O/pytest_$v.log | cut -c1-260 | tail -40
  done 2>&1 | tee $O/degraded.log
  cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_real_vs_budget.log 2>&1; echo "real build vs recorded budget rc=$?" | tee -a $O/degraded.log; tail -2 $O/pytest_real_vs_budget.log | tee -a $O/degraded.log ;;
e2e)
  SECONDS=0; timeout 1500 python tests/e2e_parity.py --frames 16 --steps 20 --out $O/e2e_parity.json > $O/e2e.log 2>&1; echo "e2e rc=$? wall ${SECONDS}s"; grep -v "^{" $O/e2e.log | tail -5
  python -c "import json; d=json.load(open('$O/e2e_parity.json')); print({k: (v['rel_l2'], v['cosine']) for k, v in d.items() if isinstance(v, dict) and 'rel_l2' in v})" ;;
bench)
  SECONDS=0; MD_BENCH_DUMP=$O/shapes_all.txt timeout 1500 python bench.py > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "cfg1 rc=$? wall ${SECONDS}s"; summ $O/bench_cfg1.json ;;
bench2)
  timeout 600 python bench.py --config 2 --no-cpu-baseline --no-pmc > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"; summ $O/bench_cfg2.json ;;
bench4)
  MD_BENCH_DUMP=$O/shapes_cfg4.txt timeout 900 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"; summ $O/bench_cfg4.json ;;
prof)
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
  cd $R
  DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
  rm -rf $O/prof; head -30 $O/kernel_stats.md ;;
cumask)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/ubench/cu_mask_map.hip -o /tmp/cu_mask_map && timeout 60 /tmp/cu_mask_map > $O/cu_mask_map.log 2>&1; echo "cu_mask_map rc=$?"; cat $O/cu_mask_map.log
  timeout 400 python tools/cu_partition.py > $O/cu_partition.log 2> $O/cu_partition.err; echo "cu_partition rc=$?"; cat $O/cu_partition.log; tail -5 $O/cu_partition.err ;;
kern)
  MD_ITERS=${MD_ITERS:-30} MD_WARM=5 timeout 600 python tools/bench_kernels.py $MD_KERN 2>&1 | grep -v amdgpu | tee $O/kern.log ;;
ab)
  cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
  for r in 1 2; do for v in $MD_AB; do
    cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
    MD_BENCH_DUMP=$O/shapes_${v}_$r.txt timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc 2>/dev/null > $O/ab_${v}_$r.json; echo "== $v (round $r)"; summ $O/ab_${v}_$r.json
  done; done 2>&1 | tee $O/ab.log
  cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so ;;
trace)
  cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
  SP_TRACE_LIB=${SP_TRACE_LIB:-trace} timeout 300 python tools/sp_trace.py ${MD_TRACE:-k640 n1280 ffout} > $O/sp_trace.log 2>&1; echo "trace rc=$?"
  cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so
  grep -E "^==|median" $O/sp_trace.log; awk '{ for (i = 6; i <= NF; i += 7) if ($i + 0 > 5000 && $i + 0 < 100000) printf "%s kt %s: step0 + epilogue = %s cycles\n", FILENAME, $2, $i }' $O/sp_trace.log | head -12 ;;
abenv)
  # same-box A/B of ONE environment knob in ONE library: MD_AB_ENV=MD_WS_EXTRA MD_AB_VALUES="0 1" [MD_AB_FLAGS="--no-share"]; two rounds, family table + per-shape dumps
  for r in 1 2; do for v in ${MD_AB_VALUES:-0 1}; do
    env ${MD_AB_ENV:-MD_NONE}=$v MD_BENCH_DUMP=$O/shapes_${v}_$r.txt timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc $MD_AB_FLAGS 2>/dev/null > $O/ab_${v}_$r.json
    echo "== ${MD_AB_ENV}=$v (round $r)"; summ $O/ab_${v}_$r.json
  done; done 2>&1 | tee $O/ab.log ;;
abflag)
  # same-box A/B of a bench.py flag: MD_AB_FLAG="--no-share" (off = the flag given); two rounds
  for r in 1 2; do for f in "$MD_AB_FLAG" ""; do
    timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc $f 2>/dev/null > $O/ab.json; echo "== flag '${f:-(none)}' (round $r)"; summ $O/ab.json
  done; done 2>&1 | tee $O/ab.log ;;
kernab)
  # micro-benchmarks (tools/bench_kernels.py $MD_KERN) under each library of MD_AB (tools/ab/lib_*.so): ablation / A-B builds at the kernel level
  cp mikudance_amd/libmdance_hip.so /tmp/lib_keep.so
  for v in $MD_AB; do
    cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
    echo "== $v"; MD_ITERS=${MD_ITERS:-30} MD_WARM=5 timeout 300 python tools/bench_kernels.py ${MD_KERN:-fused} 2>&1 | grep -v amdgpu | grep -E "gemm|conv|attn|norm|temporal"
  done > $O/kernab.log 2>&1
  cp /tmp/lib_keep.so mikudance_amd/libmdance_hip.so; cat $O/kernab.log ;;
kernenv)
  for r in 1 2; do for v in ${MD_AB_VALUES:-0 1}; do echo "== ${MD_AB_ENV}=$v (round $r)"; env ${MD_AB_ENV:-MD_NONE}=$v MD_ITERS=30 MD_WARM=5 timeout 400 python tools/bench_kernels.py ${MD_KERN:-gemm skinny} 2>&1 | grep -v amdgpu | grep -E "gemm|conv|attn|norm|temporal"; done; done > $O/kern.log 2>&1; cat $O/kern.log ;;
ranks2)
  # bench.py's multi-rank paths with the REAL kernels on the one GPU a lease has (both ranks on cuda:0, collectives over gloo): clip data-parallel (rank-local and
  # --scatter) at configs[1], window-parallel at configs[4].  Control-flow records (per-rank diagnostics, collectives timed outside the region), NOT scaling figures.
  export MD_DIST_BACKEND=gloo
  r2() { timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc "${@:3}" > $O/$2.json 2> $O/$2.err; echo "$2 rc=$?"; }
  r2 29511 bench_2rank_dp; r2 29512 bench_2rank_dp_scatter --scatter; r2 29513 bench_cfg4_2rank_window_parallel --config 4 --window-parallel
  summ $O/bench_2rank_dp.json $O/bench_2rank_dp_scatter.json $O/bench_cfg4_2rank_window_parallel.json ;;
rccl1)
  # ONE rank, process group forced (MD_DIST_FORCE=1): every collective of mikudance_amd/dp.py -- scatter, gather, all_reduce, gather_object, broadcast_object_list,
  # the device-bound barrier -- goes through RCCL on the one GPU a lease has.  Proves the library loads and accepts the calls as made; says nothing about xGMI.
  export MD_DIST_FORCE=1
  B="--gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc"
  MASTER_PORT=29541 timeout 600 python bench.py $B --scatter > $O/rccl1_dp_scatter.json 2> $O/rccl1_dp_scatter.err; echo "rccl1 dp scatter rc=$?"
  MASTER_PORT=29542 timeout 600 python bench.py $B --window-parallel > $O/rccl1_window_parallel.json 2> $O/rccl1_window_parallel.err; echo "rccl1 window-parallel rc=$?"
  unset MD_DIST_FORCE
  summ $O/rccl1_dp_scatter.json $O/rccl1_window_parallel.json
  python -c "
import json
for f in ('$O/rccl1_dp_scatter.json', '$O/rccl1_window_parallel.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1]); m = d['multi_gpu']; print(f.split('/')[-1], m['collectives'], m['comm_ms_outside_timed_region'], m['mode'])
"; tail -3 $O/rccl1_dp_scatter.err ;;
queues)
  # one CFG clip / one guidance-free clip / two guidance-free clips from two processes / two CFG clips from two processes (profiles/r06_ab_two_queues.log)
  export MD_DIST_BACKEND=gloo
  B="--steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc"
  timeout 400 python bench.py $B > $O/a.json 2>/dev/null; timeout 400 python bench.py $B --guidance 1.0 > $O/b.json 2>/dev/null
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 $B --guidance 1.0 > $O/c.json 2>/dev/null
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 $B > $O/d.json 2>/dev/null
  summ $O/a.json $O/b.json $O/c.json $O/d.json ;;
*) echo "unknown step $step" ;;
esac
done
