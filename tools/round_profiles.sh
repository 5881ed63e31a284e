#!/bin/bash
# usage (GPU box, repo root): tools/round_profiles.sh <round-tag>   -- everything profiles/ is refreshed from:
# kernel stats (bf16 graph replay / bf16 serial eager / f32 / bf16 with the folded attention), three PMC passes (bf16, eager), bench lines, stress json -> gpurun_out/
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
# (bench.py's LAST stdout line is the compact result object; the full one is bench_detail.json, rewritten by every run: keep each run's copy)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_bf16.log 2>&1; tail -1 gpurun_out/${tag}_bench_bf16.log > gpurun_out/${tag}_bench_bf16.json
cp bench_detail.json gpurun_out/${tag}_bench_bf16_detail.json
timeout 400 python bench.py --steps 20 --warmup 5 --dtype f32 --no-variants > gpurun_out/${tag}_bench_f32.log 2>&1; tail -1 gpurun_out/${tag}_bench_f32.log > gpurun_out/${tag}_bench_f32.json
cp bench_detail.json gpurun_out/${tag}_bench_f32_detail.json
timeout 300 python tools/bench_stress.py 2>/dev/null | tail -1 > gpurun_out/${tag}_stress_gcn.json
timeout 300 python tools/bench_kernels.py textgcn tail lstm imgbank folded_c16 mha_bf16 > gpurun_out/${tag}_bench_kernels.txt 2>&1
MGNNS_BENCH_GRAPH=1 timeout 300 python tools/bench_kernels.py textgcn tail folded_c16 mha_bf16 >> gpurun_out/${tag}_bench_kernels.txt 2>&1
timeout 200 python tools/graph_timeline.py > gpurun_out/${tag}_timeline.txt 2>&1
timeout 400 tools/prof.sh ${tag}_bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-variants
timeout 400 tools/prof.sh ${tag}_bf16_serial --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-graph --single-stream
timeout 400 tools/prof.sh ${tag}_f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants --dtype f32
timeout 400 tools/prof.sh ${tag}_folded --steps 20 --warmup 5 --no-cpu-baseline --no-variants --attn folded
# the parity-grade mode with the reference's formulation (split-bf16 attention core): bench line, serial kernel stats, two PMC passes
timeout 400 python bench.py --steps 20 --warmup 5 --dtype bf16x3 --attn faithful --no-variants --no-cpu-baseline > gpurun_out/${tag}_bench_bf16x3.log 2>&1; tail -1 gpurun_out/${tag}_bench_bf16x3.log > gpurun_out/${tag}_bench_bf16x3.json
cp bench_detail.json gpurun_out/${tag}_bench_bf16x3_detail.json
timeout 400 tools/prof.sh ${tag}_bf16x3_serial --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-graph --single-stream --dtype bf16x3 --attn faithful
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 tools/pmc.sh ${tag}_x3_$c $c --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-graph --dtype bf16x3 --attn faithful > /dev/null
done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 400 tools/pmc.sh ${tag}_$c $c --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-graph > /dev/null
done
# configs[4] SpMM kernels: kernel trace + FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes (tools/dev/spmm_pmc.sh)
timeout 300 tools/dev/spmm_pmc.sh ${tag}_spmm_d4e-4_F1024 random 1024 direct 0 > gpurun_out/${tag}_spmm_pmc_d4e-4_F1024.txt 2>&1
timeout 300 tools/dev/spmm_pmc.sh ${tag}_spmm_d4e-4_F2048 random 2048 direct 0 > gpurun_out/${tag}_spmm_pmc_d4e-4_F2048.txt 2>&1
timeout 300 tools/dev/spmm_pmc.sh ${tag}_spmm_d1e-2_F1024 dense 1024 tiled 0 12 > gpurun_out/${tag}_spmm_pmc_d1e-2_F1024.txt 2>&1
timeout 300 python tools/dev/slabcopy_exp.py > gpurun_out/${tag}_slabcopy.jsonl 2>/dev/null
timeout 300 python tools/dev/gather_exp.py > gpurun_out/${tag}_gather.jsonl 2>/dev/null
# the judged summaries, written ON the box (the rocpd databases are too large to travel) into a directory that is merged back
MGNNS_PROFILES_OUT=$root/gpurun_out/profiles_${tag} python tools/write_profiles.py ${tag} ${tag}
# only the summaries travel back: drop the databases
rm -rf gpurun_out/prof_${tag}_* gpurun_out/pmc_${tag}_* gpurun_out/kt_${tag}_*
ls gpurun_out | grep ${tag} | head -60
