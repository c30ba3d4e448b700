#!/usr/bin/env python3
"""Static VALU mix of a kernel's ISA by issue class (tools/ubench_issue.hip, profiles/r06_valu_issue_peak.txt): full-rate instructions
(~1 000 G wave-instr/s chip-wide), half-rate ones (~570) and quarter-rate ones (~300), and the full-rate slots they add up to.
    hipcc ... -S --cuda-device-only forma_amd/csrc/lines.hip -o lines.s;  python tools/isa_mix.py lines.s _Z11k_rasterizeILb1E [first_line last_line]"""
import re, sys
FULL = {"v_mov_b32", "v_lshrrev_b32", "v_or_b32", "v_and_b32", "v_xor_b32", "v_sub_u32", "v_subrev_u32", "v_add_u32", "v_mul_f32", "v_add_f32",
        "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_bitop3_b32", "v_xnor_b32", "v_not_b32", "v_nop"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rcp_iflag_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
           "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_mul_f64"}
src, sym = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
on, n, cnt = False, 0, {"full": 0, "half": 0, "quarter": 0, "salu": 0, "lds": 0, "vmem": 0}
for line in open(src):
    if not on:
        on = line.startswith(sym) and ":" in line
        continue
    if line.startswith(".Lfunc_end"):
        break
    n += 1
    if n < lo or n > hi:
        continue
    m = re.match(r"\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+)", line)
    if not m:
        continue
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
    if op.startswith("s_"):
        cnt["salu"] += 1
    elif op.startswith("ds_"):
        cnt["lds"] += 1
    elif op.startswith("v_"):
        cnt["full" if op in FULL else "quarter" if op in QUARTER else "half"] += 1
    else:
        cnt["vmem"] += 1
valu = cnt["full"] + cnt["half"] + cnt["quarter"]
slots = cnt["full"] + 2 * cnt["half"] + 4 * cnt["quarter"]
print("%s lines %d-%d: VALU %d = full %d + half %d + quarter %d -> %d full-rate slots (%.2f per instruction); SALU %d, LDS %d, memory %d"
      % (sym, lo, min(hi, n), valu, cnt["full"], cnt["half"], cnt["quarter"], slots, slots / max(valu, 1), cnt["salu"], cnt["lds"], cnt["vmem"]))
