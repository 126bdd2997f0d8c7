#!/bin/bash
# round-4 GPU call 2: concat-free UNets, conv / norm input pitch, 3x1 taps, 256x128 tile, VAE at size; bench + VAE per-shape dump
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c2; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unets_gpu.py tests/test_blocks_gpu.py tests/test_vae_gpu.py tests/test_gemm_sp_gpu.py \
  -k "slice or 3x1 or eta_ddim or groupnorm or conv3x3 or instnorm or test_unets or test_blocks or test_vae or auto or 256x128" \
  -x -q --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log
MD_BENCH_DUMP=$O/shapes.txt timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/c2/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","vae_ms_per_clip","e2e_frames_per_s")})
print({k:round(v["ms_per_clip"],1) for k,v in d["kernel_families"].items()})
PY
MD_VAE_DUMP=$O/vae_shapes.txt timeout 600 python tools/bench_vae.py > $O/vae.json 2> $O/vae.err; echo "vae rc=$?"; cat $O/vae.json
