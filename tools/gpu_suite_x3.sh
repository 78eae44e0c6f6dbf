#!/bin/bash
# The whole `-m gpu` suite three times back to back on one lease; tails kept under
# gpurun_out/suite_x3/ (copied to profiles/r04/ by hand afterwards).
mkdir -p gpurun_out/suite_x3
rc_all=0
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/suite_x3/run$i.log 2>&1
  rc=$?
  grep -E "passed|failed|error" gpurun_out/suite_x3/run$i.log > gpurun_out/suite_x3/run$i.tail.txt
  echo "run $i rc=$rc: $(cat gpurun_out/suite_x3/run$i.tail.txt)"
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
