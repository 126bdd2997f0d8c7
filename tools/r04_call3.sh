#!/bin/bash
# round-4 GPU call 3: attention with the 16-row last O^T tile (d = 40, 8) and the 192 x 128 sp tile: parity, then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c3; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unets_gpu.py tests/test_gemm_sp_gpu.py tests/test_blocks_gpu.py \
  -k "attention or test_unets or test_blocks or 192x128" -x -q --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -14 $O/pytest.log
cp mikudance_amd/libmdance_hip.so /tmp/keep.so
for r in 1 2; do for v in r04 nosmallt; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu; done; done > $O/ab_attention_smallt.log 2>&1
cp /tmp/keep.so mikudance_amd/libmdance_hip.so
cat $O/ab_attention_smallt.log
for r in 1 2; do for nt in 0 2 32; do echo "== MD_GEMM_SP_NT=$nt (round $r)"; MD_GEMM_SP_NT=$nt python tools/bench_kernels.py small 2>&1 | grep -v amdgpu; done; done > $O/ab_tile_192x128.log 2>&1
echo "== shapes, automatic" >> $O/ab_tile_192x128.log; python tools/bench_kernels.py shapes 2>&1 | grep -v amdgpu >> $O/ab_tile_192x128.log
cat $O/ab_tile_192x128.log
timeout 600 python bench.py --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/c3/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["achieved"])
print({k:round(v["ms_per_clip"],1) for k,v in d["kernel_families"].items()})
PY
