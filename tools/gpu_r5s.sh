#!/bin/bash
# round 5, call s: the round's evidence on the final build: PMC passes of the workloads (their summaries priced into the bench line that follows), bench.py as the
# driver runs it, rocprofv3 kernel stats of the same command, C2 with the reference's samplers, the mixed Cornell box on its three device paths, smoke
out=gpurun_out/r5s; mkdir -p $out
PREFIX=r05 STEPS=20 WARMUP=5 bash tools/gpu_profiles.sh r5s 2>&1 | tail -40
bash tools/gpu_r5g.sh > /dev/null 2>&1; cp gpurun_out/r5g/c2_samplers.txt $out/c2_samplers.txt; cat $out/c2_samplers.txt
bash tools/gpu_r5k.sh > /dev/null 2>&1; cp gpurun_out/r5k/cmixed_paths.txt $out/cmixed_paths.txt; cat $out/cmixed_paths.txt
python __graft_entry__.py smoke 2>&1 | tail -3 | tee $out/smoke.txt
