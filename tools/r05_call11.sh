#!/bin/bash
# Round 5, GPU call 11: the driver's N = 2 command line with the REAL kernels at full width, two ranks sharing the one GPU over gloo
# (MD_DIST_BACKEND=gloo: RCCL needs one GPU per rank) -- control flow of the round-5 build (streamed weight synthesis, per-user cache,
# rank-local staging, gather), not a scaling figure.  Cold weight cache on this fresh box: rank 0 writes it while rank 1 synthesises beside it.
TAG=${1:-c11}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SECONDS=0
MD_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/bench_2rank_full_gloo.json 2> $O/bench_2rank.err; echo "2-rank rc=$? wall ${SECONDS}s"
tail -1 $O/bench_2rank_full_gloo.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','n_gpus','n_ranks_seen','ms_per_step','setup_s','peak_hbm_gb')}, d['config']['parallelism'], d['config']['input_staging'])"
tail -3 $O/bench_2rank.err
