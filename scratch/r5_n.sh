#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
for c in c4 c5; do for m in "" "--sync-encode"; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras --no-dropin $m 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c $m', 'ms/step',d['ms_per_step'],'value',d['value'],'parity',(d.get('parity_check') or {}).get('ok'), d['config']['encode_call'][:30])"
done; done
