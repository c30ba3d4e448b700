#!/bin/bash
# where do the waves of the frame's kernels spend their cycles?  (for next round's plan)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
timeout 200 python tools/pmc_pass.py SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY > $O/pmc_waves_a.txt 2>&1
timeout 200 python tools/pmc_pass.py SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM > $O/pmc_waves_b.txt 2>&1
timeout 200 python tools/pmc_pass.py SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD > $O/pmc_waves_c.txt 2>&1
timeout 200 python tools/pmc_pass.py SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES > $O/pmc_waves_d.txt 2>&1
tail -n 22 $O/pmc_waves_a.txt $O/pmc_waves_b.txt $O/pmc_waves_c.txt $O/pmc_waves_d.txt | cut -c1-200
