#!/bin/bash
# A/B of bench.py's legs under an environment switch, alternating in one box.  usage: tools/bench_ab.sh VAR "0 1" [repeats] [bench args...]
VAR=$1; VALS=$2; REP=${3:-3}; shift 3
for i in $(seq $REP); do for F in $VALS; do
  env $VAR=$F python bench.py --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
g=lambda k: (d.get(k) or {}).get('scans_per_s') if isinstance(d.get(k),dict) else d.get(k)
print('$VAR=$F value %.0f two_thread %s host_fed %s serial %s batched %s' % (d['value'], g('two_thread'), g('host_fed'), d.get('serial_scans_per_s', g('serial')), (d.get('batched') or {}).get('value')))
"; done; done
