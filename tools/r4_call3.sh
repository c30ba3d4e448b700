#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -n 6 --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
V="base.bin new.bin new.bin@FORMA_HIP_DEBUG=no_prezero"
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C3 band 59,76 F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 400 python tools/ab_fast.py --rounds 2 --frames 150 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 40 $V
  echo "== C4 band 224,288 F=3"; AB_BAND=224,288 AB_INFLIGHT=3 timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 150 $V
) > $O/ab.txt 2>&1
grep -v "^---- \|identical" $O/ab.txt | cut -c1-250
