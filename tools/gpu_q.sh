#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log; tail -c 1500 gpurun_out/bench_final.json
