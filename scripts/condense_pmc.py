#!/usr/bin/env python3
"""Condenses the counters of the distance sweep from a scripts/profile_bench.sh
output directory into profiles/pmc_knn2_mfma.json, stamped with the hash of the
kernel source they were collected on (bench.py only quotes them while that
source is unchanged).  usage: condense_pmc.py gpurun_out/prof_<tag> <source tag>"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(out, tag):
    pmc = json.load(open(os.path.join(out, "pmc.json")))
    name = [k for k in pmc if k.startswith("knn2_mfma16_kernel")][0]
    c = {k: v["mean_per_launch"] for k, v in pmc[name].items()}
    pairs, n = 64, 8192
    fetch_kb, write_kb = c["FETCH_SIZE"], c["WRITE_SIZE"]
    hbm = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    records = pairs * n * 4 * 16                       # one 16-byte record per sweep lane of a query
    algo = pairs * (2 * n * 128 + n * 4) + records     # int8 rows of both images + cin of the train image + records
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in c else c["SQ_BUSY_CYCLES"] / 32.0
    src = os.path.join(ROOT, "sfm_amd", "csrc", "knn_kernels.hip")
    res = {
        "kernel": name, "pairs_per_launch": pairs, "n_desc": n,
        "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
        "fetch_size_kb_reported": fetch_kb, "write_size_kb_reported": write_kb,
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported; KB = 1024 B",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo,
        "mfma_instructions_per_launch": c["SQ_INSTS_MFMA"],
        "valu_instructions_per_launch_incl_mfma": c["SQ_INSTS_VALU"],
        "mfma_busy_cycles_per_launch": c["SQ_VALU_MFMA_BUSY_CYCLES"],
        "kernel_cycles_per_launch": cycles,
        "mfma_pipe_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles,
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
        "cycles_note": "kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; MFMA busy cycles are summed over the 1024 SIMDs "
                       "(16 cycles per v_mfma_i32_16x16x64_i8)",
        "source": "rocprofv3 --pmc passes (one counter group per run, scripts/profile_bench.sh) over `python3 bench.py "
                  "--steps 3 --warmup 1` with the extra blocks off, profiles/%s_bench_pmc.json" % tag,
    }
    json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_knn2_mfma.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
