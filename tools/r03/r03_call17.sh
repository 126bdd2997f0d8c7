#!/bin/bash
# Round 3, GPU call 17: conv tap-change arithmetic spread over the pieces; transposed-output projections on gemm_sp_kernel (swapped operands)
TAG=${1:-r3s}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gemm_sp_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_sp or conv or gemm or transpos" > $O/pytest_k.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest_k.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in base tsp; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; timeout 300 python tools/bench_kernels.py tgemm conv 2>&1 | grep -v amdgpu; done; done > $O/ab.log 2>&1
python - <<PY
import re,collections
rows=collections.OrderedDict(); cur=None
for l in open("$O/ab.log"):
    m=re.match(r"== (\S+) \(round", l)
    if m: cur=m.group(1); continue
    m=re.match(r"(.{44})\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m and cur: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(3)))
print("%-46s %10s %10s" % ("shape (TFLOP/s, best of 2)", "before", "after"))
for k,v in rows.items():
    g=lambda n: max(v[n]) if v[n] else float("nan")
    print("%-46s %10.1f %10.1f  %+5.0f %%" % (k, g("base"), g("tsp"), 100 * (g("tsp") / g("base") - 1)))
PY
for v in base tsp; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  echo "== e2e $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"
done 2>&1 | tee $O/e2e.log
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
timeout 900 python -m pytest tests/test_unets_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "g9 or g8 or full_size_step" > $O/pytest_g9.log 2>&1; echo "g9 rc=$?"; tail -3 $O/pytest_g9.log
