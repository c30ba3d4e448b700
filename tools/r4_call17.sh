#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
export PYTHONUNBUFFERED=1
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
for Q in 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$Q C3"
  GPU_MAX_HW_QUEUES=$Q timeout 600 python tools/band_proxy.py --slots 2,3,4,5,6 --frames 300 --out $O/band_c3_q$Q.json 2>&1 | grep '^{' | cut -c1-420
done
echo "== GPU_MAX_HW_QUEUES=8 C4"
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 3,4,5,6 --frames 300 --out $O/band_c4_q8.json 2>&1 | grep '^{' | cut -c1-420
