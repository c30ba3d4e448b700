#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
export PYTHONUNBUFFERED=1
for v in new fr; do
  echo "== sort tests on $v" >> $O/pytest.log
  FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/$v.bin timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu -k "sort or triangles or cubics or mixed or e2e" --timeout 600 -p no:cacheprovider -n 4 >> $O/pytest.log 2>&1
done
tail -12 $O/pytest.log
V="base.bin new.bin fr.bin"
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 40 $V
  echo "== cubics"; timeout 400 python tools/ab_fast.py --workload cubics-1080p --rounds 2 --frames 60 $V
  echo "== circles"; timeout 400 python tools/ab_fast.py --workload circles-20k --rounds 1 --frames 60 $V
) > $O/ab.txt 2>&1
grep -v "^---- \|identical" $O/ab.txt | cut -c1-250
