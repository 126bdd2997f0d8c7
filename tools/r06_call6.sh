#!/bin/bash
# Round 6, GPU call 6: where do the 21-33 k cycles of a residual epilogue go?  Trace of diagnostic builds (results wrong by construction): without the stores / without the residual loads.
R=${GRAFT_REPO_ROOT:-.}; cd $R
for lib in trace trace_nostore trace_noload; do echo "#### $lib"; MD_SP_PF=0 SP_TRACE_LIB=$lib MD_TRACE="n1280 ffout n1280_nores" bash tools/r06_gpu.sh c6_$lib trace; grep -E "^kt (19|39) " gpurun_out/c6_$lib/sp_trace.log | cut -c1-160; done
