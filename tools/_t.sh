python -m pytest tests/test_gpu_render.py tests/test_gpu_trainers.py -q -x -k "graph_replayed or renderer_trainer" -s > gpurun_out/r6_t.log 2>&1
grep -n "Fatal\|passed\|failed\|graph-replayed\|overflow redo\|Error" gpurun_out/r6_t.log | head -20
python tools/train_hostprof.py 2>&1 | tail -1
