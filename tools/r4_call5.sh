#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python tools/band_proxy.py --slots 1,2,3,4 --frames 300 --out $O/band_c3.json > $O/band_c3.log 2>&1
timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 1,2,3,4 --frames 300 --out $O/band_c4.json > $O/band_c4.log 2>&1
tail -5 $O/band_c3.log $O/band_c4.log
