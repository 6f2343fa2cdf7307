for st in 10 13 16 20 30; do
  echo "== RT_SPLIT_START=$st"
  RT_SPLIT_START=$st timeout -k 10 160 python tools/rank_time.py 2 4 2>&1 | grep world
done
for ra in 64 128 512; do
  echo "== RT_SPLIT_RATIO=$ra"
  RT_SPLIT_RATIO=$ra timeout -k 10 160 python tools/rank_time.py 2 4 8 2>&1 | grep world
done
