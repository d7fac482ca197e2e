# round 4: tile x-extent vs row length in the z-marching HBM kernels (tools/hbm_shape_probe.py), + the Gaussian two-planes-ahead form for 7+ taps only
export TMPDIR=/tmp
O=gpurun_out/r4hbm; mkdir -p $O
timeout 300 python tools/hbm_shape_probe.py > $O/shape.json 2>> $O/err.txt; cat $O/shape.json
timeout 200 python tools/hbm_prefetch_ab.py > $O/ab2.jsonl 2>> $O/err.txt; cat $O/ab2.jsonl | cut -c1-700
