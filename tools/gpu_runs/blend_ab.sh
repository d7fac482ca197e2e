# A/B of the blend over the two logits layouts + the two logits writers (tools/blend_bench.py); writes gpurun_out/blend/*
O=gpurun_out/blend; mkdir -p $O
timeout 600 python tools/blend_bench.py > $O/blend_bench.json 2> $O/blend_bench.err; python - <<'PY'
import json
r = json.load(open("gpurun_out/blend/blend_bench.json"))
for x in r["runs"]:
    print(f"{x['ms']:8.3f} ms {x['frac_of_8TBps']:.3f}  {x['variant']}  {x.get('bitwise_equal_to_window_major', '')}")
PY
tail -3 $O/blend_bench.err
