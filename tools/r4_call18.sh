#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
V="new4.bin new6.bin new6.bin@FORMA_HIP_DEBUG=no_sort_heads"
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== cubics"; timeout 400 python tools/ab_fast.py --workload cubics-1080p --rounds 1 --frames 60 $V
  echo "== C3 band F=4"; GPU_MAX_HW_QUEUES=8 AB_BAND=59,76 AB_INFLIGHT=4 timeout 400 python tools/ab_fast.py --rounds 1 --frames 150 $V
) > $O/ab.txt 2>&1
grep -v "^---- \|identical" $O/ab.txt | cut -c1-250
