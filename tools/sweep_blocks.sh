#!/bin/bash
# Sweep the row-kernel grid cap (BJX_MAX_BLOCKS) on the headline bench.
for mb in 2048 4096 16384 65536; do
  BJX_MAX_BLOCKS=$mb python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ic-mode 2>&1 | tail -1 > /tmp/sw.json
  python -c "import json; d=json.load(open('/tmp/sw.json')); print($mb, round(d['value']/1e6,1), 'M/s leapfrog us', round(d['roofline']['avg_launch_us'],1))"
done
