"""The measurement helpers the driver's bench run goes through, on CPU (no GPU, no rocprofv3 run): the counter tables of
`tools/pmc_round.py` from rocprofv3's CSV, `bench.live_traffic` falling back without losing the line, and the issue classes
`tools/isa_mix.py` prices a kernel's ISA with (profiles/r06_valu_issue_peak.txt)."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _csv(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write('"Kernel_Name","Counter_Name","Grid_Size","Counter_Value"\n')
        for r in rows:
            f.write('"%s","%s",%d,%f\n' % r)


def test_pmc_round_means_over_the_frame_sized_launches(tmp_path, monkeypatch):
    pmc_round = importlib.import_module("pmc_round")
    one = "void k_onesweep<8, false, true, 0>(unsigned long const*, unsigned long*)"
    sync = "void k_runs_wave<0>(unsigned long const*)"
    per_frame = "void k_runs_wave<2>(unsigned long const*)"

    def fake_run_pass(counters, d, timeout=None):
        rows = []
        for c in counters:
            v = {"FETCH_SIZE": 100.0, "WRITE_SIZE": 50.0}.get(c, 7.0)
            rows += [(one, c, 262144, v), (one, c, 262144, v + 2.0), (one, c, 1024, 1e9)]      # (a small launch of the same kernel: ignored)
            rows += [(sync, c, 4096, v)] + [(per_frame, c, 4096, v)] * 3
        _csv(os.path.join(d, "x", "1_counter_collection.csv"), rows)
        return {"config": {"pixel_segments": 1000, "canvas": [64, 64]}}
    monkeypatch.setattr(pmc_round, "run_pass", fake_run_pass)
    kern, line = pmc_round.collect([["FETCH_SIZE"], ["WRITE_SIZE"]], timeout=5)
    k = kern["k_onesweep<8, false, true, 0>"]
    assert k["FETCH_SIZE"] == 101.0 and k["WRITE_SIZE"] == 51.0 and k["dispatches"] == 2
    assert k["hbm_bytes_per_launch"] == int(101.0 * 1024 * 2 + 51.0 * 1024)                       # gfx950: FETCH_SIZE x 2
    assert kern["k_runs_wave<0>"]["dispatches"] == 1 and kern["k_runs_wave<2>"]["dispatches"] == 3
    assert line["config"]["pixel_segments"] == 1000
    assert "--no-pmc" in pmc_round.BENCH                                                           # (the child never profiles itself)


def test_live_traffic_falls_back_without_losing_the_line(monkeypatch):
    import bench
    out = {"kernels_us": {"k_onesweep": {"us_per_frame": 120.0}, "k_paint_wave": {"us_per_frame": 100.0}}}
    monkeypatch.setattr("shutil.which", lambda name: None)
    assert bench.live_traffic(out) is None                                                         # no rocprofv3
    monkeypatch.setattr("shutil.which", lambda name: "/opt/rocm/bin/rocprofv3")
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    assert bench.live_traffic(out) is None                                                         # the run is being profiled itself
    monkeypatch.delenv("ROCPROFILER_SOMETHING")
    pmc_round = importlib.import_module("pmc_round")

    def boom(passes, timeout=None):
        raise subprocess.TimeoutExpired("rocprofv3", timeout)
    monkeypatch.setattr(pmc_round, "collect", boom)
    assert bench.live_traffic(out) is None                                                         # a pass timed out
    # ... and with counters: the per-frame instantiations, this run's own launch time for the painter
    kern = {"k_onesweep<8, false, true, 0>": {"hbm_bytes_per_launch": 17000, "dispatches": 6},
            "k_onesweep<9, false, true, 0>": {"hbm_bytes_per_launch": 99, "dispatches": 1},
            "k_paint_wave<false, true, 4>": {"hbm_bytes_per_launch": 5000, "SQ_INSTS_VALU": 5.0e7, "SQ_LDS_IDX_ACTIVE": 100.0,
                                             "SQ_LDS_BANK_CONFLICT": 13.0, "dispatches": 3}}
    monkeypatch.setattr(pmc_round, "collect", lambda passes, timeout=None: (kern, {"config": {"pixel_segments": 1000, "canvas": [64, 64]}}))
    live = bench.live_traffic(out)
    assert live["roofline"]["traffic"] == 17000 and live["roofline"]["traffic_over_algorithmic"] == round(17000 / 16000.0, 4)
    assert live["roofline"]["traffic_source"].startswith("measured in this run")
    p = live["roofline_painter"]
    assert p["achieved"] == 500.0 and p["frac"] == round(500.0 / bench.VALU_PEAK_GINST, 4)
    assert p["frac_of_measured"] == round(500.0 * bench.PAINT_SLOTS_PER_INST / bench.VALU_MEASURED_GINST, 4) and p["lds_bank_conflict_ratio"] == 0.13


def test_isa_mix_prices_half_rate_instructions_double(tmp_path):
    s = tmp_path / "k.s"
    s.write_text("_Z3fooPv:\n"
                 "\tv_fma_f32 v1, v2, v3, v4\n\tv_add_u32_e32 v1, v2, v3\n\tv_bitop3_b32 v1, v2, v3, v4 bitop3:0x90\n"      # full rate
                 "\tv_fma_f64 v[0:1], v[2:3], v[4:5], v[6:7]\n\tv_cvt_i32_f32_e32 v1, v2\n\tv_cndmask_b32_e64 v1, v2, v3, s[0:1]\n"
                 "\tv_lshlrev_b32_e32 v1, 3, v2\n"                                                                              # half rate
                 "\tv_rcp_f32_e32 v1, v2\n"                                                                                     # quarter
                 "\ts_waitcnt vmcnt(0)\n\tds_read_b32 v1, v2\n\tglobal_load_dword v1, v[2:3], off\n"
                 ".Lfunc_end0:\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), str(s), "_Z3foo"], capture_output=True, text=True, check=True).stdout
    assert "VALU 8 = full 3 + half 4 + quarter 1 -> 15 full-rate slots" in out and "SALU 1, LDS 1, memory 1" in out
