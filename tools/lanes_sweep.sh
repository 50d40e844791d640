for k in 1 2 3 4; do HOP_LANES=$k timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-ctus 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('lanes $k', d['value'], d['ms_per_step'], d.get('result_crc'))"; done
