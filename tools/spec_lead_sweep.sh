#!/bin/bash
# A/B sweeps of the speculative NUTS tail at the C3 shape (tools/bench_nuts_spec.py): rows at which the tail is entered
python tools/bench_nuts_spec.py --T 100 400 --reps 1 --spec 0 128 512 2048 | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['runs']: print(r['T'], r['spec_rows'], round(r['M_per_s'],1), r['identical_to_first'], {k: (round(v,3) if isinstance(v,float) else v) for k,v in r['spec'].items()})
"
