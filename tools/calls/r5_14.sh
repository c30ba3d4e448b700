#!/bin/bash
# call 14: carry large variant with 5 runs per lane (a 4K row in one piece); CPU oracle sweep after the allocation-free painter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_14; mkdir -p $O
export AB_KERNELS=1
timeout 300 python tools/ab_fast.py --rounds 3 --frames 60 rpt4.bin rpt5.bin > $O/ab_c3.log 2>&1; tail -5 $O/ab_c3.log; grep kernels $O/ab_c3.log | tail -2
timeout 300 python tools/cpu_sweep.py paris-like-30k-4k 8,16,32,48,64,96,128 > $O/cpu_sweep.log 2>&1; cat $O/cpu_sweep.log
OMP_PROC_BIND=spread timeout 300 python tools/cpu_sweep.py paris-like-30k-4k 16,32,64,128 > $O/cpu_sweep_spread.log 2>&1; cat $O/cpu_sweep_spread.log
