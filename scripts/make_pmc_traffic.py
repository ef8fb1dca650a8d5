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
                              ("lstm_fwd_persistent_bf_kernel<AP=3, WP=3> (bf16 planes)", "lstm_fwd_persistent_bf_kernel<2, 2, 3, 3, false>", 2.0),
                              ("lstm_fwd_persistent_bf_kernel<AP=2, WP=2, F16> (fp16 planes)", "lstm_fwd_persistent_bf_kernel<2, 2, 2, 2, true>", 2.0),
                              ("gemm_f32_split_f16_big_kernel(input->gates)", "gemm_f32_split_f16_big_kernel<true, true>", 1.59),
                              ("gemm_f32_split_f16_big_kernel(input gradient)", "gemm_f32_split_f16_big_kernel<true, false>", 1.59),
                              ("amax_kernel(gate gradients)", "amax_kernel<true, true, false, true>", 2.0),
                              ("gemm_f32_split_bf16_big_kernel(input->gates)", "gemm_f32_split_bf16_big_kernel<true, true>", 1.59)):
        name, r = find(fw, prefix)
        if r:
            out["bytes_per_launch"][key] = (fmul * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
            out["kernels"][key] = name
        name, r = find(sq, prefix)
        if r and r.get("GRBM_GUI_ACTIVE"):
            out["mfma_busy"][key] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    out["bytes_per_launch"]["lstm_bwd_persistent_kernel"] = out["bytes_per_launch"].get("lstm_bwd_persistent_q4_kernel")
    out["bytes_per_launch"]["gemm_f32_mfma_kernel(input->gates)"] = out["bytes_per_launch"].get("gemm_f32_split_f16_big_kernel(input->gates)")
    # the WIDE configurations (round 6: scripts/collect_profiles_wide.sh): per configuration the recurrence kernels' traffic per launch
    # against their algorithmic bytes, and their matrix-pipe occupancy -- quoted by bench.py's secondary legs
    wide = {}
    for name, cfgname, S, T, nl in (("cfg4", "cfg4_f32", 32, 1000, 5), ("cfg4_bf16_forward", "cfg4_bf16", 32, 1000, 5), ("cfg5", "cfg5", 64, 3000, 6)):
        f1, f2 = os.path.join(ROOT, "profiles", f"{tag}_{cfgname}_pmc_fetch_write.md"), os.path.join(ROOT, "profiles", f"{tag}_{cfgname}_pmc_sq.md")
        if not (os.path.exists(f1) and os.path.exists(f2)):
            continue
        fw2, sq2 = table(f1), table(f2)
        H, nd = 1024, 2
        rows = float(T) * S
        # algorithmic bytes per launch: backward reads G (4H), C (H), dY (H) and writes DG (4H) per frame and direction;
        # forward reads G (4H) and writes C, Y (H each) [+ the exchange copy of m_t]
        alg = {"lstm_bwd": rows * nd * (4 * H + H + H + 4 * H) * 4.0, "lstm_fwd": rows * nd * (4 * H + H + H) * 4.0}
        flops = 2.0 * S * 4 * H * H * nd * T
        ent = {"config": {"config": name.split("_")[0], "T": T, "S": S}, "kernels": {}}
        for k, r in fw2.items():
            if not k.startswith(("lstm_bwd_persistent", "lstm_fwd_persistent")):
                continue
            kind = "lstm_bwd" if k.startswith("lstm_bwd") else "lstm_fwd"
            b = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
            a_bytes = alg[kind]
            if "ksplit_h" in k:   # round 6: the gate gradients are written a second time as fp16 planes (4H x 4 B per frame and direction) and read back as such by the next step
                a_bytes = rows * nd * (4 * H + H + H + 4 * H + 4 * H + 4 * H) * 4.0
            e = {"bytes_per_launch": b, "algorithmic_bytes_per_launch": a_bytes, "traffic_over_algorithmic": b / a_bytes,
                 "profiled_avg_us": r["avg us (profiled)"], "flops_per_launch": flops}
            q = sq2.get(k)
            if q and q.get("GRBM_GUI_ACTIVE"):
                e["mfma_busy"] = q["SQ_VALU_MFMA_BUSY_CYCLES"] / (q["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            ent["kernels"][k] = e
        wide[name] = ent
    out["wide"] = wide
    out["wide_source"] = f"profiles/{tag}_cfg4_f32_*, {tag}_cfg4_bf16_*, {tag}_cfg5_* (scripts/collect_profiles_wide.sh)"
    w = {k: v for k, v in fw.items() if k.startswith("wait_for_word")}
    out["milestone_waiter_rows"] = {k: v.get("avg us (profiled)") for k, v in w.items()}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else "")
