#!/bin/bash
# the sort's histograms taken by the rasterizer (RasHist): A/B against HEAD, then the GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
export PYTHONUNBUFFERED=1
V="head.bin rh1.bin rh1.bin@FORMA_HIP_DEBUG=no_ras_hist"
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== cubics"; timeout 400 python tools/ab_fast.py --workload cubics-1080p --rounds 1 --frames 60 $V
  echo "== C3 band F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 400 python tools/ab_fast.py --rounds 1 --frames 150 $V
) > $O/ab.txt 2>&1
grep -v "^---- " $O/ab.txt | cut -c1-260
timeout 1200 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
