#!/usr/bin/env python
"""Instruction-issue roofline of k_head_phase<0/1/2>: what bounds a launch when neither HBM nor the matrix pipe does.

Inputs:
  * the instruction mix per launch from the SQ counters (tools/visits/r6d.sh: separate rocprofv3 --pmc passes; tools/pmc_summary.py
    writes <dir>/pmci<tier>_summary.json): wave-instructions by class -- VALU (without MFMA), MFMA, LDS, VMEM, SALU;
  * the issue costs measured on the MI355X by tools/shadow_probe.hip (profiles/round6/r6b_shadow_probe_*.txt):
      - a wave64 VALU instruction occupies its SIMD's vector issue for 5.0 cycles (arm S1: +5.0 cycles per filler, both MFMA types);
      - v_mfma_f32_32x32x16_f16 holds the matrix pipe 32 cycles and runs BESIDE the VALU: up to 5 (one wave) / 6.4 (two waves) VALU per MFMA
        are free, i.e. a stream with more than 6.4 VALU per f16 MFMA is VALU-issue bound and its MFMAs cost nothing extra;
      - v_mfma_f32_32x32x2_f32 holds the pipe 64 cycles and does NOT overlap with VALU work at all (arm I: +9 cycles per filler at K = 2,
        +5 beyond; no arrangement of one or two waves hides a single VALU behind it): f32 MFMA and VALU time ADD.
So per SIMD (1024 of them) and launch:
      fp32 tier   cycles_min = (64 N_mfma + 5 N_valu) / 1024                     bound "f32 mfma + valu (serial)"
      f16 tiers   cycles_min = max(5 N_valu, 32 N_mfma) / 1024                   bound "valu issue" (N_valu / N_mfma = 10 split, 23 fast)
`frac` = cycles_min / 2.4 GHz (the clock the MFMA roofline is priced at) over the kernel's measured launch time (bench.py: HIP events, live);
`frac_at_measured_clock` uses the shader clock of the launch itself (GRBM_GUI_ACTIVE / 8 XCDs / duration of the counter pass).

    python tools/issue_roofline.py gpurun_out/r6d [samples_per_frame] > profiles/round6/r6d_issue_roofline.json
"""
import json
import os
import sys

VALU_CYCLES, MFMA_F16_CYCLES, MFMA_F32_CYCLES, SIMDS, NOMINAL_GHZ = 5.0, 32.0, 64.0, 1024, 2.4


def tier(summary_path, name, kernel):
    d = json.load(open(summary_path))
    k = d[kernel]
    n = {c: k[f"SQ_INSTS_{c}"] for c in ("VALU", "MFMA", "LDS", "VMEM_RD", "VMEM_WR", "SALU", "SMEM")}
    f32 = name == "fp32"
    valu_c, mfma_c = VALU_CYCLES * n["VALU"] / SIMDS, (MFMA_F32_CYCLES if f32 else MFMA_F16_CYCLES) * n["MFMA"] / SIMDS
    cycles_min = valu_c + mfma_c if f32 else max(valu_c, mfma_c)
    meas_cycles = k["GRBM_GUI_ACTIVE"] / 8
    return {
        "kernel": kernel, "instructions_per_launch": n, "valu_per_mfma": n["VALU"] / n["MFMA"],
        "valu_issue_cycles_per_simd": valu_c, "mfma_pipe_cycles_per_simd": mfma_c,
        "bound": "f32 mfma + valu, serial (the f32 MFMA does not overlap with VALU work: shadow probe)" if f32 else
                 "valu issue (5 cycles per wave64 VALU and SIMD; the f16 MFMAs fit beside them: shadow probe)",
        "cycles_min_per_simd": cycles_min, "time_min_us_at_2p4_ghz": cycles_min / (NOMINAL_GHZ * 1e3),
        "counter_pass": {"launch_us": k["mean_duration_us"], "cycles_per_launch": meas_cycles, "shader_clock_ghz": meas_cycles / k["mean_duration_us"] / 1e3,
                         "frac_at_measured_clock": cycles_min / meas_cycles, "mfma_busy_cycles_per_simd": k["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS,
                         "note": "counter passes run ~10 % slower than unprofiled launches: bench.py divides time_min by ITS live launch time"},
    }


def main():
    out_dir = sys.argv[1]
    res = {"constants": {"valu_cycles": VALU_CYCLES, "mfma_f16_cycles": MFMA_F16_CYCLES, "mfma_f32_cycles": MFMA_F32_CYCLES, "simds": SIMDS,
                         "source": "tools/shadow_probe.hip on the MI355X: profiles/round6/r6b_shadow_probe_one_wave_vs_two_waves_per_simd.txt"}}
    for name, kernel in (("fp32", "k_head_phase"), ("split", "k_head_phase_split"), ("fast", "k_head_phase_fast")):
        p = os.path.join(out_dir, f"pmci{name}_summary.json")
        if os.path.exists(p):
            res[name] = tier(p, name, kernel)
            res["_source_digest"] = json.load(open(p)).get("_source_digest")
    if len(sys.argv) > 2:      # evaluated samples per frame of the frames the counter passes rendered (bench.py scales the mix to its own fixture)
        res["samples_per_frame"] = float(sys.argv[2])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
