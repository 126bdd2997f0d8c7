#!/bin/bash
# final check of the round's last commit: the GPU suite + smoke as the driver runs them, then the default bench line with its wall time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-final}; mkdir -p $O
bash tools/r04_suite.sh ${1:-final} | tail -8
SECONDS=0; timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s" | tee $O/bench.time
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","vae_ms_per_clip","e2e_frames_per_s","n_ranks_seen")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
