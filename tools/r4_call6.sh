#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -n 6 --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
V="base.bin new.bin new2.bin"
( echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 40 $V
  echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 1 --frames 60 $V
) > $O/ab.txt 2>&1
grep -v "^---- \|identical" $O/ab.txt | cut -c1-250
for Q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$Q"
  GPU_MAX_HW_QUEUES=$Q timeout 600 python tools/band_proxy.py --slots 3,4,6 --frames 300 --out $O/band_c3_q$Q.json 2>&1 | grep '^{' | cut -c1-400
done
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 3,4,6 --frames 300 --out $O/band_c4_q8.json 2>&1 | grep '^{' | cut -c1-400
