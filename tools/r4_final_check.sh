#!/bin/bash
# the round's last look at the GPU side: smoke, the suite as the driver runs it, the suite under both poison bytes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_default.log 2>&1; tail -1 $O/pytest_default.log
for b in 0xFF 0x00; do
  FORMA_HIP_DEBUG=poison=$b,poison_frame=$b timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_poison_$b.log 2>&1; tail -1 $O/pytest_poison_$b.log
done
