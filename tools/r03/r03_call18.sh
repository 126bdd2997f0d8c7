#!/bin/bash
# Round 3, GPU call 18: K rows of the attention kernel padded to an odd number of bank quads at d = 80 / 160 (LDS bank conflicts)
TAG=${1:-r3u}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?"; tail -3 $O/pytest_attn.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in nopad kpad; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; timeout 300 python tools/bench_kernels.py attn xattn 2>&1 | grep -v amdgpu; done; done > $O/ab.log 2>&1
python - <<PY
import re,collections
rows=collections.OrderedDict(); cur=None
for l in open("$O/ab.log"):
    m=re.match(r"== (\S+) \(round", l)
    if m: cur=m.group(1); continue
    m=re.match(r"(.{44})\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m and cur: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(3)))
print("%-46s %10s %10s" % ("shape (TFLOP/s, best of 2)", "unpadded", "padded"))
for k,v in rows.items():
    g=lambda n: max(v[n]) if v[n] else float("nan")
    print("%-46s %10.1f %10.1f  %+5.0f %%" % (k, g("nopad"), g("kpad"), 100 * (g("kpad") / g("nopad") - 1)))
PY
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
timeout 600 python -m pytest tests/test_unets_gpu.py -m gpu -q -x -k "g9_full_size_unets or g8" > $O/pytest_g9.log 2>&1; echo "g8/g9 rc=$?"; tail -3 $O/pytest_g9.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))" | tee $O/e2e.log
