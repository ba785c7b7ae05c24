#!/bin/bash
# round 3, GPU call 29: 64 x 128 tiles (four resident workgroups per CU at C5) against the 128 x 128 default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c29
rm -rf $O; mkdir -p $O
cd $R
BJX_DENSE_TILE64=1 timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_frows_dense_gpu.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
 for T in 0 1; do
  for A in 0 6; do
   BJX_DENSE_TILE64=$T BJX_DENSE_ABLATE=$A timeout 300 python tools/bench_dense.py > $O/t${T}_a${A}_$rep.json 2>> $O/dense.err
   python -c "import json; j=json.load(open('$O/t${T}_a${A}_$rep.json')); r=j['roofline']; print('tile64=$T ablate $A rep $rep', round(r['avg_launch_us'],1), 'us', round(r['frac'],3), round(j['value']/1e6,1))"
  done
 done
done
for N in 65536; do
 for T in 0 1; do
  BJX_DENSE_TILE64=$T timeout 300 python tools/bench_dense.py --chains $N --steps 3 --warmup 1 > $O/n${N}_t$T.json 2>> $O/dense.err
  python -c "import json; j=json.load(open('$O/n${N}_t$T.json')); r=j['roofline']; print('N=$N tile64=$T', round(r['avg_launch_us'],1), 'us', round(r['frac'],3))"
 done
done
