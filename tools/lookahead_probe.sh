for la in 1 2 3 4 6 8; do
  echo "[LOOKAHEAD=$la]"
  HMK_LOOP_LOOKAHEAD=$la python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('antibodies', round(d['wall_ms'],2), 'loop', round(d['device_loop_ms'],2), 'rounds', d['loop_rounds'])"
  HMK_LOOP_LOOKAHEAD=$la python tools/greedy_phases.py --sorted 100000 300000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['call']==2: print(d['n'], round(d['wall_ms'],2), 'loop', round(d['device_loop_ms'],2), 'rounds', d['loop_rounds'])"
done
