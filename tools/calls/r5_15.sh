#!/bin/bash
# call 15: round-4 library (+ the timing entry point) against the round-5 library on one box, every configuration; host CPU limits
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_15; mkdir -p $O
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  quota $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)  affinity $(taskset -p $$ | cut -c1-200)" | tee $O/host_cpu.txt
lscpu | grep -E "Model name|Socket|Thread|NUMA node|Core|^CPU\(s\)" | tee -a $O/host_cpu.txt
export AB_KERNELS=1
timeout 400 python tools/ab_fast.py --rounds 3 --frames 60 r4.bin r5.bin > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log; grep kernels $O/ab_c3.log | tail -2
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 r4.bin r5.bin > $O/ab_c4.log 2>&1; tail -4 $O/ab_c4.log; grep kernels $O/ab_c4.log | tail -2
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 r4.bin r5.bin > $O/ab_c2.log 2>&1; tail -4 $O/ab_c2.log; grep kernels $O/ab_c2.log | tail -2
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 2 r4.bin r5.bin > $O/ab_circ.log 2>&1; tail -4 $O/ab_circ.log; grep kernels $O/ab_circ.log | tail -2
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 2 r4.bin r5.bin > $O/ab_c3_band.log 2>&1; tail -4 $O/ab_c3_band.log; grep kernels $O/ab_c3_band.log | tail -2
AB_BAND=224,288 timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 r4.bin r5.bin > $O/ab_c4_band.log 2>&1; tail -4 $O/ab_c4_band.log
