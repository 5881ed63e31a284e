# r6: imgbank_split forms, three alternating rounds on one box: old = round 5's kernel, v1 = both half-waves convert, ad2 = + pooled maxima by
# DPP (A-fragment ring of two), new = the library's (ring of three), ad4 = ring of four; then the phase trace of the library's form
python -m pytest tests/test_ops_gpu.py -x -q -k "imgbank" 2>&1 | tail -2
export MGNNS_BENCH_GRAPH=1
for r in 1 2 3; do
  for v in ${IS_FORMS:-old v1 ad2 new ad4}; do
    if [ $v = new ]; then unset MGNNS_LIB; else export MGNNS_LIB=mgnns_amd/variants/lib_is_$v.so; fi
    echo "-- round $r $v"; python tools/bench_kernels.py imgbank 2>&1 | grep split
  done
done
unset MGNNS_BENCH_GRAPH
MGNNS_LIB=mgnns_amd/variants/lib_is_trace.so python tools/dev/is_trace.py
