#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_prof_all.sh rh3.bin rhx1.bin rhx3.bin rh3.bin@FORMA_HIP_DEBUG=no_ras_hist rh3.bin rhx1.bin rhx3.bin 2>&1 | sed 's/carry_rows.*onesweep/../; s/runs_count.*//'
