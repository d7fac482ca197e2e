export WB_SKIP_DIRECT=1 KB_BATCH=64
for c in 1 2 3 4; do echo chunks=$c; MONAI_AMD_W2_CHUNKS=$c python tools/wino_bench.py 2>&1 | tail -1 | python -c "
import json,sys;r=json.loads(sys.stdin.read())
print([(x['cin'],x['cout'],x['edge'],x.get('wino2d',{}).get('ms')) for x in r['rows']])"; done
