#!/bin/bash
# Round 3, GPU call 9: gemm_sp_kernel with 64-deep K tiles (whole-cache-line DMA pieces, A ring of 3 / W ring of 2) vs the 32-deep build
TAG=${1:-r3i}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== sp parity (bk64)"; MD_GEMM_SP=1 timeout 600 python tests/gemm_sp_check.py > $O/sp_check.log 2>&1; echo "sp parity rc=$?"; tail -3 $O/sp_check.log
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in bk32 bk64; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; MD_GEMM_SP=1 timeout 300 python tools/bench_kernels.py conv gemm shapes 2>&1 | grep -v amdgpu | grep -v "x320x320\|x640x640\|x640x320 \|8192x8192x8192\|x1280x640 \|x1920x640 \|294912x320x640"; done; done > $O/ab_bk.log 2>&1
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
python - <<PY
import re,collections
rows=collections.OrderedDict(); cur=None
for l in open("$O/ab_bk.log"):
    m=re.match(r"== (\S+) \(round", l)
    if m: cur=m.group(1); continue
    m=re.match(r"(.{44})\s+([\d.]+) ms\s+([\d.]+) TFLOP", l)
    if m and cur: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(3)))
print("%-46s %10s %10s" % ("shape (TFLOP/s, best of 2)", "bk32", "bk64"))
for k,v in rows.items():
    g=lambda n: max(v[n]) if v[n] else float("nan")
    print("%-46s %10.1f %10.1f  %+5.0f %%" % (k, g("bk32"), g("bk64"), 100 * (g("bk64") / g("bk32") - 1)))
PY
echo "== e2e"
for d in 0 2; do MD_GEMM_SP=$d timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('MD_GEMM_SP=$d: %.3f f/s  ' % d['value'] + '  '.join('%s %.0f' % (k, v['ms_per_clip']) for k, v in f.items()))"; done | tee $O/e2e.log
