#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -n 6 --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python tools/band_proxy.py --slots 1,3 --frames 300 --out $O/band_c3.json 2>&1 | grep '^{' | cut -c1-420
timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 1,3 --frames 300 --out $O/band_c4.json 2>&1 | grep '^{' | cut -c1-420
