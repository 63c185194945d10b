for w in 2 4 8; do for g in 0 1; do
echo "world=$w grouped=$g: $(timeout -k 5 100 python tools/dev_shard_timing.py 20 $w 8 0 $g 2>&1 | tail -1)"
done; done
