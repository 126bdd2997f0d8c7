#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command with the round's last build -> gpurun_out/$1/kernel_stats.md (tools/r04_records.sh step, alone)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof_final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-pmc > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
cd $R
DB=$(ls $O/prof/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocprof.py $DB $O/kernel_stats.md $O/prof_bench.json > /dev/null
rm -rf $O/prof
head -40 $O/kernel_stats.md
