#!/bin/bash
# Round 6, GPU call 9: bench.py's multi-rank paths with the REAL kernels on the one GPU a lease has (both ranks on cuda:0, collectives over gloo: MD_DIST_BACKEND=gloo):
# (a) clip data-parallel, 2 ranks, full width, configs[1]; (b) window-parallel, 2 ranks, configs[4] (3 windows of 30 frames: rank 0 takes two, rank 1 one).
# Control-flow records (per-rank diagnostics, collectives timed outside the region), NOT scaling figures: two ranks share one GPU.
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=$R/gpurun_out/c9; mkdir -p $O
export MD_DIST_BACKEND=gloo
run() { timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc "${@:3}" > $O/$2.json 2> $O/$2.err; echo "$2 rc=$?"; }
run 29511 bench_2rank_dp
run 29512 bench_2rank_dp_scatter --scatter
run 29513 bench_cfg4_2rank_window_parallel --config 4 --window-parallel
python - $O/bench_2rank_dp.json $O/bench_2rank_dp_scatter.json $O/bench_cfg4_2rank_window_parallel.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); m = d["multi_gpu"]
        print(f.split("/")[-1], "%.3f f/s %.1f ms" % (d["value"], d["ms_per_step"]), d["scaling"], d["config"]["parallelism"], "distinct GPUs", m["n_distinct_gpus"], "aliasing", m["gpu_aliasing"],
              [round(r["own_ms_per_step"], 1) for r in m["per_rank"]], m["comm_ms_outside_timed_region"], m["collectives"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/*.err | cut -c1-300
