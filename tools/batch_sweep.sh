#!/bin/bash
# forwards in flight (--videos) x videos per forward (--batch): clips/s of bench.py
for vb in "3 1" "1 3" "2 2" "2 3" "3 2" "1 6" "3 3" "2 4" "1 8" "3 4" "4 3" "2 6" "3 5" "2 8"; do set -- $vb
python bench.py --videos $1 --batch $2 --no-cpu-baseline --no-post 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes',$1,'batch',$2,'value',round(d['value']),'ms',round(d['ms_per_step'],3))"; done
