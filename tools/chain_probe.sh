#!/bin/bash
# the second loop without chains, with chains from the first round, and as shipped (from round 256) -- HMK_LOOP_CHAIN=0 / 1 / unset:
# antibodies, 1e5 and 1e6 in the default order, 3e5; third call of each
for e in "HMK_LOOP_CHAIN=0" "HMK_LOOP_CHAIN=1" "X=default"; do
  echo "[$e]"
  env $e python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('antibodies', round(d['wall_ms'],2), 'loop', round(d['device_loop_ms'],2), 'rounds', d['loop_rounds'], 'clusters', d['clusters'])"
  env $e python tools/greedy_phases.py --sorted 100000 300000 1000000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['call']==2: print(d['n'], round(d['wall_ms'],2), 'loop', round(d['device_loop_ms'],2), 'rounds', d['loop_rounds'], 'result_list', d['result_list'])"
done
