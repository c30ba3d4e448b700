#!/bin/bash
# The whole `-m gpu` suite once per FORMA_HIP_DEBUG string (every variant a policy can pick, forced on every frame of every test),
# one line per string into gpurun_out/suite_switches.txt:   tools/suite_switches.sh ["sw1" "sw2" ...]
mkdir -p gpurun_out
OUT=gpurun_out/suite_switches.txt
: > $OUT
SW=("$@")
if [ ${#SW[@]} -eq 0 ]; then
  SW=("" "sync" "runs_blk=1,blk_round=2,runs_chain=0" "runs_blk=0,runs_chain=0" "runs_chain=1" "paint_split=4,split_first=40" "paint_split=0,tail_poll=0"
      "strip_tiles=100000000" "paint_quad=2" "force_cull" "order_thr=1,poison=0xFF,poison_frame=0xFF" "poison=0x00,poison_frame=0x00,force_cull"
      "sync,force_cull,strip_tiles=100000000" "digit_bits=4" "digit_bits=9,no_bias" "carry_half=2,carry_covl=0" "global_runsort" "no_ras_hist,no_prezero"
      "multi_layout=exchange,xchg=copy" "sort_cus=64,no_order,no_cull")
fi
for sw in "${SW[@]}"; do
  t0=$(date +%s)
  log=gpurun_out/suite_switches_$(echo "${sw:-default}" | tr -c 'A-Za-z0-9\n' '_').log
  FORMA_HIP_DEBUG="$sw" timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $log 2>&1
  rc=$?
  res=$(tail -n 40 $log)
  [ $rc -eq 0 ] && rm -f $log                            # (the whole log of a run that failed — or died — stays in gpurun_out/)
  last=$(echo "$res" | grep -E "passed|failed|error" | tail -1)
  fails=$(echo "$res" | grep -E "^FAILED|^ERROR" | cut -c1-160 | tr '\n' ';')
  echo "[${sw:-(default)}]  $last  ($(( $(date +%s) - t0 )) s)  $fails" | tee -a $OUT
done
