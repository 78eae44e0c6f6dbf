#!/bin/bash
# configs[2] as SLAM on the GPU: export a drive to be mapped, replay it from the C++ host, min_count sweep
N=${1:-248}; STEPS=${2:-200}; WARM=${3:-40}
D=/tmp/mapdrive_$N
python bench.py --export-mapping-drive $D --mapping-frames $N 2>&1 | tail -1
for mc in ${MCS:-3 8 16 32}; do
for extra in ${MODES:-pipeline --no-pipeline}; do
  [ "$extra" = pipeline ] && extra=""
  echo "== min_count $mc $extra"
  VELO_TRACE_REGISTER=1 tools/stream_driver $D --mapping --steps $STEPS --warmup $WARM --threshold 1 --min-count $mc $extra 2> gpurun_out/mapping_err.txt | tail -1
  head -1 gpurun_out/mapping_err.txt; tail -2 gpurun_out/mapping_err.txt
done
done
