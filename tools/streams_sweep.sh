#!/bin/bash
# videos-in-flight sweep of bench.py (value in clips/s); GPU_MAX_HW_QUEUES raises the number of hardware queues HIP streams map to
for q in 4 8; do
for v in 3 4 5 6 8; do GPU_MAX_HW_QUEUES=$q python bench.py --videos $v --no-cpu-baseline --no-post 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq',$q,'videos',$v,'value',round(d['value']),'ms',round(d['ms_per_step'],3))"; done
done
