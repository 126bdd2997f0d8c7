#!/bin/bash
# round-4 GPU call 4: attn4_kernel (d = 40, one wave per SIMD, software pipelined): parity, then same-box A/B against attn2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attention" -q --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
for r in 1 2; do for v in 1 0; do echo "== MD_ATTN_V4=$v (round $r)"; MD_ATTN_V4=$v python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu | grep "D=40"; done; done 2>&1 | tee $O/ab_attention_v4.log
