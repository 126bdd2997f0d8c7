cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gemm_sp_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
SP_TRACE_LIB=trace_la1 MD_TRACE="n1280_nores n1280" bash tools/r06_gpu.sh c3_old trace
SP_TRACE_LIB=trace MD_TRACE="n1280 k640 ffout" bash tools/r06_gpu.sh c3_new trace
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep0.so
for r in 1 2; do for v in la1 la; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; MD_ITERS=30 MD_WARM=5 timeout 300 python tools/bench_kernels.py gemm skinny shapes 2>&1 | grep -v amdgpu | grep -E "gemm"; done; done > gpurun_out/c3_kern.log 2>&1
cp /tmp/lib_keep0.so mikudance_amd/libmdance_hip.so
cat gpurun_out/c3_kern.log | head -120
MD_AB="la1 la" bash tools/r06_gpu.sh c3_ab ab
