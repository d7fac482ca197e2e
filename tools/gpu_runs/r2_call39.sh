O=gpurun_out/r2c39; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "swin or layernorm or linear or conv3d_configs or unetr" 2>&1 | tail -3
python bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_line_swinunetr.json 2> $O/err.txt
cut -c1-260 $O/bench_line_swinunetr.json
python bench.py --steps 3 --warmup 1 --cpu-windows 0 2>/dev/null | cut -c1-260
