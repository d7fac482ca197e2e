mkdir -p gpurun_out
python tools/stream_chunk_sweep.py 2>&1 | tail -1 | tee gpurun_out/stream_chunks3.txt
for c in 3 4 5 8 11 16 22; do MONAI_AMD_RS_CHUNKS=$c MONAI_AMD_GS_CHUNKS=5 python tools/stream_chunk_sweep.py 2>&1 | tail -1 | tee -a gpurun_out/stream_chunks3.txt; done
