#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c26
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_device_target.py tests/test_nuts_free_gpu.py -q -x) > $O/tests.log 2>&1
tail -12 $O/tests.log
