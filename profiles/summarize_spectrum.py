#!/usr/bin/env python3
"""profiles/<tag>_spectrum_pmc.json from the raw output of tests/tools/prof_spectrum.sh (gpurun_out/prof_spec/): counters of the
128-energy job of tests/tools/bench_spectrum.py (the middle third of the launches of each pass), the kernel time of the same launches
from the kernel trace, executed FP64 operations.   python3 profiles/summarize_spectrum.py r05 <job_ms of the same build>"""
import csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
job_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
grid = sys.argv[3] if len(sys.argv) > 3 else "log"             # "uniform": the raw output of prof_spectrum.sh uniform
src = os.path.join(root, "gpurun_out", "prof_spec_uniform" if grid == "uniform" else "prof_spec")
pm = json.load(open(os.path.join(src, "spectrum_pmc.json")))
c = {k: v["mean_128_energies"] for k, v in pm.items()}
traces = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(traces[-1])))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main = [dur(r) for r in rows if "disk_spectrum_fast" in r["Kernel_Name"]]
n = len(main) // 3
mid = main[n:2 * n]
mid = mid[len(mid) // 2:]                                 # at the working clock
sums = [dur(r) for r in rows if "spectrum_sum" in r["Kernel_Name"]]
fl = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]) * 64
out = {"what": "disk_spectrum_fast_kernel<true>, 1024^2 pixels x 128 energies (tests/tools/prof_spectrum.sh -> bench_spectrum.py; counters of "
               "the 350 launches of the 128-energy job, separate --pmc passes); summarised by profiles/summarize_spectrum.py",
       "kernel_time_us_rocprof_128_energies_second_half_of_its_launches": sum(mid) / len(mid) / 1e3,
       "sum_kernel_us": sum(sums) / len(sums) / 1e3,
       "job_ms_hip_events_128_energies": job_ms,
       "counters_per_launch": c,
       "executed_fp64_flop_per_launch": fl,
       "executed_fp64_flop_per_pixel_energy_pair_upper_bound": fl / (1024 * 1024 * 128),
       "valu_wave_instr_per_launch": c["SQ_INSTS_VALU"],
       "history": {"round_4": "full-precision exp + Newton division per pair, unpaired trace at 2 waves per SIMD: 0.214 ms",
                   "round_5_first_form": "pairs, closed-form frame: 65.3 M VALU wave-instructions per launch",
                   "round_5_18_slots": "51.6 M, 0.1116 ms, busy 82.6 %",
                   "round_5_16_slots": "42.2 M, 0.0960 ms: n from the low word of t + 1.5 2^52, degree-6 minimax 2^f, fused sum, (x, amplitude) "
                                       "pairs read at immediate offsets with the loop counter on the scalar unit",
                   "round_5_frame_without_quotients": "41.5 M, 0.0949 ms: the local frame in four square roots and two reciprocals per pixel",
                   "round_5_eight_terms_per_reciprocal": "this record: u = e^-x, term = a u / (1 - u), eight terms over one reciprocal seed through a tree of (N, D) pairs: 15.25 slots per pair"}}
if grid == "uniform":
    out["what"] = out["what"].replace("128 energies", "128 energies in EQUAL steps 0.1 ... 30 keV (the recurrence along the energies: k_spectrum.hip planck_runs_uniform)")
    out["history"]["round_6_uniform_grid"] = ("this record: lanes own runs of eight consecutive energies, e^-x of a pixel advances by one multiplication per energy, "
                                              "eight pixels share a reciprocal: ~7.5 issue slots per pair (15.25 on an arbitrary grid)")
json.dump(out, open(os.path.join(root, "profiles", tag + ("_spectrum_uniform_pmc.json" if grid == "uniform" else "_spectrum_pmc.json")), "w"), indent=1)
print("kernel %.2f us, sum %.2f us, VALU %.1f M, busy %.1f %%, executed flop %.3e" % (
    out["kernel_time_us_rocprof_128_energies_second_half_of_its_launches"], out["sum_kernel_us"], c["SQ_INSTS_VALU"] / 1e6,
    c.get("VALUBusy", float("nan")), fl))
