#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
export PYTHONUNBUFFERED=1
for D in "" "no_bias" "digit_bits=8" "digit_bits=8,no_bias" "sync"; do
  echo "== FORMA_HIP_DEBUG=$D" >> $O/dbg.log
  FORMA_HIP_DEBUG=$D timeout 600 python -m pytest "tests/test_gpu_exchange.py::test_exchange_full_size_triangles_10m_8k" -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  .*Assert|^FAILED" | cut -c1-200 >> $O/dbg.log
done
cat $O/dbg.log
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu -n 4 --timeout 600 -p no:cacheprovider 2>&1 | tail -5
V="base.bin new.bin new3.bin"
( echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 40 $V
  echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 1 --frames 60 $V
) > $O/ab.txt 2>&1
grep -v "^---- \|identical" $O/ab.txt | cut -c1-250
