for vb in "3 8" "2 8" "2 12" "2 16" "3 12" "3 16" "4 8" "1 16" "3 8"; do set -- $vb
python bench.py --videos $1 --batch $2 --min-timed-s 1 --no-cpu-baseline --no-post 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes',$1,'batch',$2,'value',round(d['value']),'ms',round(d['ms_per_step'],3))"; done
