for k in 20 22 24; do python tools/dev_ntt_bench.py $k 30 2 2>&1 | tail -1; done
python tools/dev_prove_loop.py 20 20 2>&1 | tail -1
