#!/bin/bash
# round-4 GPU call 5: FeedForward in row chunks (GEGLU output consumed from the memory-side cache?) -- same-box A/B end to end
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; mkdir -p $O
for r in 1 2; do for c in 0 49152 98304; do
  MD_FF_CHUNK=$c timeout 600 python bench.py --no-cpu-baseline --no-vae --no-pmc > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
d=json.load(open("$O/bench_$c.json"))
f=d["kernel_families"]
print("chunk $c round $r:", round(d["value"],3), "f/s", round(d["ms_per_step"],1), "ms  gemm", round(f["gemm"]["ms_per_clip"],1))
PY
done; done 2>&1 | tee $O/ab_ff_chunk.log
