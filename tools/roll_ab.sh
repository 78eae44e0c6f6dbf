D=/tmp/drv; [ -d $D ] || python bench.py --export-drive $D > /dev/null 2>&1
for cfg in "192 4" "192 6" "128 6" "128 8" "96 8" "256 4"; do set -- $cfg
  echo -n "roll CUs $1 lead $2: "; VELO_ROLL_CUS=$1 ./tools/stream_driver $D --steps 400 --warmup 40 --roll-lead $2 | tail -1 | cut -c50-130
done
echo -n "lead 0: "; ./tools/stream_driver $D --steps 400 --warmup 40 --roll-lead 0 | tail -1 | cut -c50-130
