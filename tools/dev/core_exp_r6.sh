export MGNNS_MHA_CORE=16 MGNNS_BENCH_GRAPH=1
V=mgnns_amd/variants
{
echo "== mha_clock (default library) =="; python tools/dev/mha_clock.py
echo "== mha_clock (trace build) =="; MGNNS_LIB=$V/lib_trace.so python tools/dev/mha_clock.py
for r in 1 2 3; do
  for v in default ring13 ring13g1; do
    if [ $v = default ]; then unset MGNNS_LIB; else export MGNNS_LIB=$V/lib_$v.so; fi
    echo "-- round $r $v"; python tools/bench_kernels.py mha_bf16 2>&1 | grep -i "L=196\|us" | head -3
  done
done
} > gpurun_out/r6_core_exp.txt 2>&1
tail -40 gpurun_out/r6_core_exp.txt
