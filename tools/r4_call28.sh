#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="sx.bin sx.bin@FORMA_X_SORT_WGS=224 sx.bin@FORMA_X_SORT_WGS=192 sx.bin@FORMA_X_SORT_WGS=160 sx.bin@FORMA_X_SORT_WGS=128"
( echo "== C3 full"; timeout 200 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C3 full F=4"; AB_INFLIGHT=4 timeout 200 python tools/ab_fast.py --rounds 1 --frames 60 $V
) > $O/ab6.txt 2>&1
grep -v "^---- " $O/ab6.txt | grep -v crc | cut -c1-260
