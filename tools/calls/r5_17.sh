#!/bin/bash
# call 17: quad painter with the next layer's segments requested a layer ahead; the new full-size tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_17; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
FORMA_HIP_DEBUG=paint_quad=2 timeout 600 python -m pytest tests -m gpu -x -q -k "quad or cubics or config or channel or e2e or culling" > $O/pytest_quad.log 2>&1; echo "pytest(quad forced subset) rc $?"; tail -2 $O/pytest_quad.log
export AB_KERNELS=1
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 3 qnopf.bin qpf.bin > $O/ab_c4.log 2>&1; tail -5 $O/ab_c4.log; grep kernels $O/ab_c4.log | tail -2
