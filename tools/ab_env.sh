#!/bin/bash
# same-box A/B of environment settings: tools/ab_env.sh "VAR=a" "VAR=b" ; WHAT selects the micro-benchmarks
for r in 1 2; do for v in "$@"; do echo "== $v (round $r)"; env $v python tools/bench_kernels.py ${WHAT:-gemm} 2>&1 | grep -v amdgpu; done; done
