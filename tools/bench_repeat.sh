#!/bin/bash
# Repeat the headline bench in fresh processes to expose cold-start effects.
show() { python -c "import json; d=json.load(open('/tmp/br.json')); print('$1', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms/step')"; }
python bench.py --no-cpu-baseline --no-ic-mode 2>/dev/null | tail -1 > /tmp/br.json; show default_1
python bench.py --no-cpu-baseline --no-ic-mode 2>/dev/null | tail -1 > /tmp/br.json; show default_2
python bench.py --no-cpu-baseline --no-ic-mode --no-launch-timing 2>/dev/null | tail -1 > /tmp/br.json; show notimer
python bench.py --no-cpu-baseline --no-ic-mode --steps 20 --warmup 5 2>/dev/null | tail -1 > /tmp/br.json; show k20w5
