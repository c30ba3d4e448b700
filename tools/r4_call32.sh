#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 300 env FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/rw1.bin python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -q -x --timeout 200 -p no:cacheprovider > $O/pytest8.log 2>&1
rc=$?; tail -2 $O/pytest8.log
[ $rc -ne 0 ] && exit 1
V="k9.bin rw1.bin"
( echo "== C3 full"; timeout 150 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 100 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== C3 band F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 200 python tools/ab_fast.py --rounds 1 --frames 150 $V
) > $O/ab10.txt 2>&1
grep -v "^---- " $O/ab10.txt | grep "crc\|==\|identical" | cut -c1-260
