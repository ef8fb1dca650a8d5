#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC tables of a profile collection (scripts/rocpd_pmc_summary.py output):
   scripts/make_pmc_traffic.py TAG COMMIT SWITCHES > profiles/pmc_traffic.json
reads profiles/TAG_pmc_fetch_write.md and profiles/TAG_pmc_sq.md.  bench.py quotes these figures as roofline.traffic / mfma_busy
(rocprofv3 --pmc cannot run inside the benchmark process) together with where they came from (roofline.traffic_source).
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section, and profiles/*_pmc_fetch_calibration.md): FETCH_SIZE / WRITE_SIZE are KiB
per dispatch; 16-B-per-lane full-line reads are tallied at half (recurrence kernels: 2 x FETCH + WRITE); the GEMMs' 64-contiguous-
byte row pattern reads at 0.63 of its bytes (1.59 x FETCH + WRITE)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eesen_amd.build import csrc_digest  # noqa: E402


def table(path):
    rows, cols = {}, None
    for line in open(path):
        if not line.startswith("|"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if cols is None:
            cols = cells
            continue
        if set(cells[0]) <= set("-:"):
            continue
        name = cells[0].strip("`")
        rows[name] = {c: (float(v) if v else None) for c, v in zip(cols[1:], cells[1:])}
    return rows


def main(tag, commit, switches):
    fw = table(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write.md"))
    sq = table(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.md"))

    def find(rows, prefix):
        hits = [(k, v) for k, v in rows.items() if k.startswith(prefix)]
        return max(hits, key=lambda kv: kv[1].get("avg us (profiled)", 0) * kv[1].get("dispatches", 0)) if hits else (None, None)

    out = {"source": f"profiles/{tag}_pmc_fetch_write.md, profiles/{tag}_pmc_sq.md (rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes, cfg2, T=1000, S=32)",
           "commit": commit, "csrc_sha": csrc_digest(), "switches": switches,
           "correction": "recurrence kernels: traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 (16-B-per-lane full-line reads are tallied at half, "
                         "MI355X_MICROARCH.md); GEMM: (1.59*FETCH_SIZE + WRITE_SIZE) per the FETCH calibration run of the same collection",
           "config": {"config": "cfg2", "T": 1000, "S": 32}, "bytes_per_launch": {}, "mfma_busy": {}, "kernels": {},
           "mfma_busy_source": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per dispatch (profiled launches: rocprofv3 --pmc lets ONE kernel run at a time)"}
    for key, prefix, fmul in (("lstm_bwd_persistent_q4_kernel", "lstm_bwd_persistent_q4_kernel", 2.0), ("lstm_fwd_persistent_kernel", "lstm_fwd_persistent_kernel", 2.0),
                              ("lstm_fwd_persistent_bf_kernel<AP=3, WP=3>", "lstm_fwd_persistent_bf_kernel<2, 2, 3, 3>", 2.0),
                              ("gemm_f32_split_bf16_big_kernel(input->gates)", "gemm_f32_split_bf16_big_kernel<true, false>", 1.59)):
        name, r = find(fw, prefix)
        if r:
            out["bytes_per_launch"][key] = (fmul * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
            out["kernels"][key] = name
        name, r = find(sq, prefix)
        if r and r.get("GRBM_GUI_ACTIVE"):
            out["mfma_busy"][key] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    out["bytes_per_launch"]["lstm_bwd_persistent_kernel"] = out["bytes_per_launch"].get("lstm_bwd_persistent_q4_kernel")
    out["bytes_per_launch"]["gemm_f32_mfma_kernel(input->gates)"] = out["bytes_per_launch"].get("gemm_f32_split_bf16_big_kernel(input->gates)")
    w = {k: v for k, v in fw.items() if k.startswith("wait_for_word")}
    out["milestone_waiter_rows"] = {k: v.get("avg us (profiled)") for k, v in w.items()}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else "")
