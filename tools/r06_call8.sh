#!/bin/bash
# Round 6, GPU call 8: same-box A/B of the shared first layers (bench.py --no-share), two rounds.
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=$R/gpurun_out/c8; mkdir -p $O
for r in 1 2; do for f in "--no-share" ""; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc $f 2>/dev/null > $O/ab.json
  python - $O/ab.json "share=${f:-on}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f = d["kernel_families"]
print(sys.argv[2], "round", "%.3f f/s %.1f ms" % (d["value"], d["ms_per_step"]), " ".join("%s %.0f" % (k, v["ms_per_clip"]) for k, v in list(f.items())[:5]))
PY
done; done 2>&1 | tee $O/ab.log
