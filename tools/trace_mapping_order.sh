#!/bin/bash
# where the host's time inside registerCore goes in both orders of the pipelined update (VELO_TRACE_REGISTER): mapping stream, C++ host
D=/tmp/mapdrive_248
python bench.py --export-mapping-drive $D --mapping-frames 248 2>&1 | tail -1
for v in "" "VELO_UPDATE_BEFORE_START=1"; do
  echo "== $v"
  env $v VELO_TRACE_REGISTER=1 tools/stream_driver $D --mapping --steps 200 --warmup 40 --threshold 1 2> /tmp/e.txt | cut -c1-160
  grep registerCore /tmp/e.txt | tail -4
  grep -c registerCore /tmp/e.txt
done
