#!/bin/bash
# Round 3, GPU call 15: tap-inner K order of the 3x3 convs (L2 reuse of the input across the nine taps): parity, A/B, FETCH_SIZE, end to end
TAG=${1:-r3p}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gemm_sp_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_sp or conv" > $O/pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -5 $O/pytest_conv.log
timeout 900 python -m pytest tests/test_unets_gpu.py -m gpu -q -x -k "g9 or g8" > $O/pytest_g9.log 2>&1; echo "g9 tests rc=$?"; tail -3 $O/pytest_g9.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in base tapin; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; timeout 300 python tools/bench_kernels.py conv shapes small 2>&1 | grep -v amdgpu | grep conv; done; done > $O/ab_tapin.log 2>&1
python - <<PY
import re,collections
rows=collections.OrderedDict(); cur=None
for l in open("$O/ab_tapin.log"):
    m=re.match(r"== (\S+) \(round", l)
    if m: cur=m.group(1); continue
    m=re.match(r"(.{44})\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m and cur: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(3)))
print("%-46s %10s %10s" % ("shape (TFLOP/s, best of 2)", "tap-outer", "tap-inner"))
for k,v in rows.items():
    g=lambda n: max(v[n]) if v[n] else float("nan")
    print("%-46s %10.1f %10.1f  %+5.0f %%" % (k, g("base"), g("tapin"), 100 * (g("tapin") / g("base") - 1)))
PY
for v in base tapin; do
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  echo "== e2e $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('%.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"
done 2>&1 | tee $O/e2e.log
cd /tmp && export TMPDIR=/tmp
export MD_ITERS=3 MD_WARM=1
for v in base tapin; do
  cp $R/tools/ab/lib_$v.so $R/mikudance_amd/libmdance_hip.so
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch_$v -o fetch -- python $R/tools/bench_kernels.py conv > $O/pmc_fetch_$v.log 2>&1
  echo "== FETCH $v"; python $R/tools/pmc_table.py $O/pmc_fetch_$v --match gemm_sp | tee $O/pmc_fetch_$v.txt
  rm -rf $O/pmc_fetch_$v
done
cp /tmp/lib_keep_ab.so $R/mikudance_amd/libmdance_hip.so
