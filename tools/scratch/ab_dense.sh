V=$1
run() { for i in 1 2 3; do python tools/bench_dense.py --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['avg_launch_us'],2), 'us', round(d['roofline']['frac'],3), round(d['value']/1e6,1), 'M/s')"; done; }
cp blackjax_amd/libbjxhip.so /tmp/libbjxhip_default.so
echo "== default"; run
cp $V blackjax_amd/libbjxhip.so
echo "== variant $V"; run
python -m pytest tests/test_dense_gpu.py tests/test_full_shape_gpu.py -x -q -k "dense or c5" 2>&1 | tail -2
cp /tmp/libbjxhip_default.so blackjax_amd/libbjxhip.so
echo "== default again"; run
