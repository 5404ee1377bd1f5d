#!/bin/bash
# round 6, gpurun call 19: closing check at HEAD -- the whole GPU suite and smoke()
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > gpurun_out/r06_gputest_head.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_gputest_head.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|real" gpurun_out/r06_gputest_head.log | cut -c1-300 | tail -6
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee gpurun_out/r06_smoke_head.txt
