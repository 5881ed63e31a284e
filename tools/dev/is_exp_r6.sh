python -m pytest tests/test_ops_gpu.py -x -q -k "imgbank" 2>&1 | tail -4
python -m pytest tests/test_model_gpu.py -x -q -k "bf16x3" 2>&1 | tail -3
export MGNNS_BENCH_GRAPH=1
for r in 1 2 3; do
  for v in old v1 new; do
    if [ $v = new ]; then unset MGNNS_LIB; else export MGNNS_LIB=mgnns_amd/variants/lib_is_$v.so; fi
    echo "-- round $r $v"; python tools/bench_kernels.py imgbank 2>&1 | grep split
  done
done
unset MGNNS_BENCH_GRAPH
MGNNS_LIB=mgnns_amd/variants/lib_is_trace.so python tools/dev/is_trace.py
