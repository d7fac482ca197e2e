for nb in 50 56 64; do
  MONAI_AMD_SW_BATCH=$nb python bench.py --steps 2 --warmup 1 --cpu-windows 0 2>/dev/null | tail -1 | python -c "
import json,sys;l=json.loads(sys.stdin.read());print('nb=$nb', round(l['ms_per_step'],1), {k:round(v['ms_total'],1) for k,v in l['conv_ms_per_step'].items()})"
done
