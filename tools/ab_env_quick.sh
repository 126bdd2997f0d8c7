#!/bin/bash
# one round of tools/ab_env.sh (same-box A/B of environment settings)
for v in "$@"; do echo "== $v"; env $v python tools/bench_kernels.py ${WHAT:-gemm} 2>&1 | grep -v amdgpu; done
