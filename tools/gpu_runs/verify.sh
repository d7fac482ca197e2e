# the whole -m gpu suite + smoke() on the current tree; writes gpurun_out/verify/*
O=gpurun_out/verify; mkdir -p $O; export TMPDIR=/tmp
python -m monai_amd.build > /dev/null      # a stale library would fail every test: rebuild if any source is newer
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -60 > $O/gpu_tests.txt; grep "^E  .*assert\|^E  .*Error" $O/gpu_tests.txt | head -5; tail -4 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
