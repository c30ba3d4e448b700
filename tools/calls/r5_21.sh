#!/bin/bash
# call 21: final build — suite, the two C3 bench lines, two and four frame slots with the digit passes on all CUs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_21; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
N=new.bin@FORMA_HIP_DEBUG
AB_INFLIGHT=4 timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 new.bin $N=sort_cus=128 > $O/ab_c3_f4.log 2>&1; tail -4 $O/ab_c3_f4.log
AB_INFLIGHT=2 timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 new.bin $N=sort_cus=128 > $O/ab_c3_f2.log 2>&1; tail -4 $O/ab_c3_f2.log
timeout 400 python tools/d2h_bench.py > $O/d2h_bench.log 2>&1; tail -3 $O/d2h_bench.log
