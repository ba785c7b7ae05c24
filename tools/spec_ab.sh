#!/bin/bash
# sweep of one environment knob of the speculative NUTS tail at C3, T = 400 (tools/bench_nuts_spec.py)
run() { python tools/bench_nuts_spec.py --T ${T:-400} --reps 1 --spec 128 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['runs']: print(r['T'], r['spec_rows'], round(r['M_per_s'],1), round(r['spec']['seconds'],3), r['spec']['sequences'], round(r['spec']['book_us_per_record'],2), r['spec']['stale'])
"; }
for v in "$@"; do echo "== $v"; env $v bash -c "$(declare -f run); run"; done
