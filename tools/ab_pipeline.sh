#!/bin/bash
# same-box end-to-end A/B: bench.py (2 timed clips, no cpu baseline) per build / environment; usage: ab_pipeline.sh "lib[:ENV=V,...]" ...
cp mikudance_amd/libmdance_hip.so /tmp/lib_keep2.so
for r in 1 2; do for spec in "$@"; do v=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so
  env $envs python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['kernel_families']
print('== $spec (round $r): %.3f f/s  gemm %.0f ms  conv %.0f  attn40 %.0f  |' % (d['value'], f['gemm']['ms_per_clip'], f['conv3x3']['ms_per_clip'], f['attention D=40']['ms_per_clip']), '  '.join('%s %.1f' % (s['label'].replace('gemm ',''), s['ms_per_clip']) for s in d['top_launch_shapes'] if 'geglu' in s['label'] or 'N=320 K=320' in s['label'] or 'N=640 K=640' in s['label']))"
done; done
cp /tmp/lib_keep2.so mikudance_amd/libmdance_hip.so
