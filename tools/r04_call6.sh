#!/bin/bash
# round-4 GPU call 6: bench.py's N > 1 path with REAL kernels: two ranks sharing the one GPU over gloo (control flow, not a scaling figure),
# both staging modes, reduced width; then full width rank-local
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c6; mkdir -p $O
export MD_DIST_BACKEND=gloo
for extra in "" "--scatter"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --small --no-vae $extra > $O/bench_2rank_small$extra.json 2> $O/bench_2rank_small$extra.err
  echo "small $extra rc=$?"; tail -c 600 $O/bench_2rank_small$extra.json | head -c 400; echo
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --no-vae > $O/bench_2rank_full.json 2> $O/bench_2rank_full.err
echo "full rc=$?"; python - <<'PY'
import json
for f in ("bench_2rank_small.json","bench_2rank_small--scatter.json","bench_2rank_full.json"):
    try:
        d=json.loads([l for l in open("gpurun_out/c6/"+f) if l.startswith("{")][-1]); print(f, d["n_gpus"], d["n_ranks_seen"], round(d["value"],2), d["config"]["input_staging"][:20], d["config"]["width"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/bench_2rank_full.err
