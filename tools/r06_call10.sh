#!/bin/bash
# Round 6, GPU call 10: is there anything in running the two CFG halves as two kernel queues?  Proxy without new code: a clip WITHOUT guidance is one clip-half
# (16-frame UNet batches, every row reads the bank = the conditional chain).  (a) one CFG clip (B = 32 kernels), (b) one guidance-free clip (B = 16 kernels, serial),
# (c) TWO guidance-free clips at once from two processes on the one GPU (two queues of B = 16 kernels): if (c) takes less than (a), two queues of half batches beat one
# queue of full batches.  (d) two CFG clips at once (the 2-rank record again on this box).
R=${GRAFT_REPO_ROOT:-.}; cd $R; O=$R/gpurun_out/c10; mkdir -p $O
export MD_DIST_BACKEND=gloo
B="--steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-pmc"
timeout 400 python bench.py $B > $O/a.json 2>/dev/null
timeout 400 python bench.py $B --guidance 1.0 > $O/b.json 2>/dev/null
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 $B --guidance 1.0 > $O/c.json 2>/dev/null
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 $B > $O/d.json 2>/dev/null
python - $O <<'PY'
import json, sys
for k, what in (("a", "one CFG clip, one queue (B = 32 kernels)"), ("b", "one guidance-free clip, one queue (B = 16)"), ("c", "TWO guidance-free clips, two processes (2 queues of B = 16)"), ("d", "TWO CFG clips, two processes (2 queues of B = 32)")):
    try:
        d = json.loads(open(sys.argv[1] + "/" + k + ".json").read().strip().splitlines()[-1])
        print("(%s) %-62s %8.1f ms per step  %.3f frames/s" % (k, what, d["ms_per_step"], d["value"]))
    except Exception as e:
        print(k, "ERR", e)
PY
