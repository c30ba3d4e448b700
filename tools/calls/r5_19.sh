#!/bin/bash
# call 19: digit passes on 128 CUs with frame slots — the other configurations, and four slots
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_19; mkdir -p $O
N=new.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $N=sort_cus=0 new.bin $N=sort_cus=192 > $O/ab_c4.log 2>&1; tail -5 $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $N=sort_cus=0 new.bin $N=sort_cus=192 > $O/ab_c2.log 2>&1; tail -5 $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 2 $N=sort_cus=0 new.bin > $O/ab_circ.log 2>&1; tail -4 $O/ab_circ.log
timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 new.bin $N=sort_cus=96 > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log
AB_INFLIGHT=4 timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 new.bin $N=sort_cus=96 > $O/ab_c3_f4.log 2>&1; tail -4 $O/ab_c3_f4.log
AB_INFLIGHT=2 timeout 400 python tools/ab_fast.py --rounds 1 --frames 60 new.bin > $O/ab_c3_f2.log 2>&1; tail -3 $O/ab_c3_f2.log
