#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/round_profiles.sh r05 > gpurun_out/r05_round.log 2>&1
tail -30 gpurun_out/r05_round.log
