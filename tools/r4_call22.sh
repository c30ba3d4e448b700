#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V="head.bin rh1.bin rh1.bin@FORMA_HIP_DEBUG=no_ras_hist head.bin rh1.bin"
bash tools/ab_prof_all.sh $V
bash tools/ab_prof_all.sh triangles-10m-8k head.bin rh1.bin
