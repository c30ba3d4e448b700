#!/bin/bash
# F=4 and the hardware queues, explicitly: the same band with 4 and with 8 queues, exported by the shell
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/band_proxy.py --slots 1,2,3,4 --frames 300 --out $O/band_c3_q$q.json > $O/band_c3_q$q.log 2>&1
  tail -1 $O/band_c3_q$q.log
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/band_proxy.py --slots 4,3,2 --frames 300 --out $O/band_c3_rev_q$q.json > $O/band_c3_rev_q$q.log 2>&1
  grep '"band"' $O/band_c3_rev_q$q.log | tail -1
done
