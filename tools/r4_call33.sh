#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/pprof.bin
timeout 200 python tools/paint_prof.py triangles-10m-8k 2>&1 | tail -16
timeout 200 python tools/paint_prof.py paris-like-30k-4k 2>&1 | tail -16
