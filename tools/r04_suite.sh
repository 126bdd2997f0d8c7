#!/bin/bash
# the whole GPU suite as the driver runs it (wall time recorded) + the smoke entry
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-suite}; mkdir -p $O
SECONDS=0; timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/pytest.log 2>&1; RC=$?; echo "wall ${SECONDS}s" > $O/pytest.time
echo "pytest rc=$RC" | tee -a $O/pytest.log
tail -40 $O/pytest.log; cat $O/pytest.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
