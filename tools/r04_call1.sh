#!/bin/bash
# round-4 GPU call 1: the new parity tests (20-step e2e at full width, eta / timestep / context_batch_size corners) and the f = 16 record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
MD_E2E_RECORD=gpurun_out/c1/e2e_f4.json timeout 1000 python -m pytest tests/test_e2e_parity_gpu.py tests/test_abi.py tests/test_unets_gpu.py \
  -k "20_steps or reduced_width or eta_positive or accepts_any_timestep or context_batch_size or library_exports or config1_full_width" \
  -x -q -s --durations=10 > gpurun_out/c1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/c1/pytest.log
tail -5 gpurun_out/c1/pytest.log
timeout 1200 python tests/e2e_parity.py --frames 16 --steps 20 --out gpurun_out/c1/e2e_f16.json > gpurun_out/c1/e2e_f16.log 2>&1
echo "f16 rc=$?"
grep -v "^{" gpurun_out/c1/e2e_f16.log | tail -5
