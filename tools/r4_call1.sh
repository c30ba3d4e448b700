#!/bin/bash
# round 4, call 1: where the band frame stands with frames in flight (the multi-GPU proxy), on the round-3 build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
export PYTHONUNBUFFERED=1
( for F in 1 3 4 6; do echo "== C3 full F=$F"; AB_INFLIGHT=$F timeout 300 python tools/ab_fast.py --rounds 1 --frames 60 base.bin; done
  for F in 3 4 6 8; do echo "== C3 band 59,76 F=$F"; AB_BAND=59,76 AB_INFLIGHT=$F timeout 300 python tools/ab_fast.py --rounds 1 --frames 120 base.bin; done
  for F in 3; do echo "== C3 band 0,17 F=$F"; AB_BAND=0,17 AB_INFLIGHT=$F timeout 300 python tools/ab_fast.py --rounds 1 --frames 120 base.bin; done
  for F in 1 3; do echo "== C4 full F=$F"; AB_INFLIGHT=$F timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 base.bin; done
  for F in 3 4 6 8; do echo "== C4 band 224,288 F=$F"; AB_BAND=224,288 AB_INFLIGHT=$F timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 120 base.bin; done
) > $O/log.txt 2>&1
tail -60 $O/log.txt
