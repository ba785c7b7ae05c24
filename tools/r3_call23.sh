#!/bin/bash
# round 3, GPU call 23: the fused dense leapfrog launch against batch size (rounds of resident tiles)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c23
rm -rf $O; mkdir -p $O
cd $R
for N in 16384 32768 65536 131072; do
  timeout 300 python tools/bench_dense.py --chains $N --steps 3 --warmup 1 > $O/dense_n$N.json 2>> $O/dense.err
  python -c "import json; j=json.load(open('$O/dense_n$N.json')); r=j['roofline']; print('fused N=$N', round(r['avg_launch_us'],1), 'us', round(r['frac'],3), round(j['value']/1e6,1))"
  BJX_DENSE_ABLATE=6 timeout 300 python tools/bench_dense.py --chains $N --steps 3 --warmup 1 > $O/plain_n$N.json 2>> $O/dense.err
  python -c "import json; j=json.load(open('$O/plain_n$N.json')); r=j['roofline']; print('plain N=$N', round(r['avg_launch_us'],1), 'us', round(r['frac'],3))"
done
python tools/probe_lib_gemm.py 2>/dev/null | tail -1
tail -2 $O/dense.err
