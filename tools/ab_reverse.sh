#!/bin/bash
# A/B: reversed-row leapfrog (default lib) vs forward-row build (libbjxhip_fwd.so), same box.
# Build the forward variant first (CPU container):
#   for f in blackjax_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
#     -ffp-contract=off -DBJX_REVERSE_ROWS=0 -c $f -o /tmp/fwd_$(basename $f .hip).o; done
#   hipcc --offload-arch=gfx950 -shared -fPIC -o blackjax_amd/libbjxhip_fwd.so /tmp/fwd_*.o
# Measured on MI355X: reverse 180.7 M/s (leapfrog 230.1 us) vs forward 175-177 M/s (237 us).
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ic-mode 2>&1 | tail -1 > /tmp/ab.json; python -c "import json; d=json.load(open('/tmp/ab.json')); print('$1', round(d['value']/1e6,1), 'M/s leapfrog us', round(d['roofline']['avg_launch_us'],1), 'ms/step', round(d['ms_per_step'],2))"; }
cp blackjax_amd/libbjxhip.so /tmp/rev.so
for i in 1 2; do
  cp /tmp/rev.so blackjax_amd/libbjxhip.so; run reverse
  cp blackjax_amd/libbjxhip_fwd.so blackjax_amd/libbjxhip.so; run forward
done
cp /tmp/rev.so blackjax_amd/libbjxhip.so
