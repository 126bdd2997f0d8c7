cp mikudance_amd/libmdance_hip.so /tmp/lib_keep_ab.so
for r in 1 2; do for v in "$@"; do cp tools/ab/lib_$v.so mikudance_amd/libmdance_hip.so; echo "== $v (round $r)"; python tools/probe_temporal_f30.py 2>&1 | grep -v amdgpu; done; done
cp /tmp/lib_keep_ab.so mikudance_amd/libmdance_hip.so
