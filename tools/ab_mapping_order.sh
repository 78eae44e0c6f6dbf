D=/tmp/mapdrive_248
python bench.py --export-mapping-drive $D --mapping-frames 248 2>&1 | tail -1
for i in 1 2; do
for v in "" "VELO_UPDATE_BEFORE_START=1"; do
  echo "== $v"
  env $v tools/stream_driver $D --mapping --steps 200 --warmup 40 --threshold 1 --min-count 20 | cut -c1-200
done; done
