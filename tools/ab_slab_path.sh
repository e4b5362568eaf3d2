#!/bin/bash
# One rank's share of configs[4] through bench.py --slab-path for engine builds under variants/: ms per substep + where the exchange sat.
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$R/variants/$lib; fi
  python3 $R/bench.py --slab-path --steps 60 --warmup 5 --no-cpu-baseline --no-breakdown 2>/dev/null | grep '^{' | python3 -c "
import json,sys
a=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step', round(a['ms_per_step'],4), 'exchange', a['exchange']['per_rank'][0] if a.get('exchange') else None)"
done
