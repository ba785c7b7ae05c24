#!/bin/bash
# round 3, GPU call 22: ablation of the fused dense leapfrog launch (timing only; ablated results are invalid)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c22
rm -rf $O; mkdir -p $O
cd $R
for A in 0 1 2 4 3 5 6 7 0; do
  BJX_DENSE_ABLATE=$A timeout 300 python tools/bench_dense.py > $O/dense_a$A.json 2>> $O/dense.err
  python -c "import json; j=json.load(open('$O/dense_a$A.json')); r=j['roofline']; print('ablate $A', round(r['avg_launch_us'],1), 'us', round(r['frac'],3), round(j['value']/1e6,1))"
done
