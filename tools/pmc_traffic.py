#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC passes -> profiles/<round>_traffic.json (read by bench.py for
`roofline.traffic`).

MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE come from the L2's memory-side request counters; on gfx950
FETCH_SIZE reports half the bytes of a 16 B/lane coalesced streaming read, other access widths and WRITE_SIZE are
uncalibrated: "calibrate on a known byte count in your own access pattern".  So four separate passes (a counter file never
mixes with tracing):
  rocprofv3 --pmc FETCH_SIZE -d cal_f -o cal -- tools/pmc_calib/pmc_calib        known-byte kernels, reads
  rocprofv3 --pmc WRITE_SIZE -d cal_w -o cal -- tools/pmc_calib/pmc_calib        known-byte kernels, writes
  rocprofv3 --pmc FETCH_SIZE -d run_f -o run -- python bench.py ...              the workload
  rocprofv3 --pmc WRITE_SIZE -d run_w -o run -- python bench.py ...
and  tools/pmc_traffic.py --calib-fetch cal_f/cal_results.db --calib-write cal_w/cal_results.db --fetch run_f/run_results.db
                          --write run_w/run_results.db --calib-bytes <bytes_per_kernel printed by pmc_calib> -o profiles/r02_traffic.json
The correction factor of a pattern = bytes the calibration kernel moved / (counter value * 1024).  A library kernel is
corrected with the factor of the pattern its dominant access has (PATTERN below; default: 16 B/lane streams), and the
uncorrected counters are kept next to the result."""
import argparse
import json
import re
import sqlite3
import sys

# dominant access pattern of the library's heavy kernels: (reads, writes); see DESIGN.md section 3 for what each one moves
PATTERN = {
    "k_range_code": ("read_b128", "write_b32"),           # 16-byte triples, 1 KiB coalesced per wave step; bytes out
    "k_evolve_small": ("read_b64", "scatter_b128"),       # sorted (key, index) in, triples scattered to stream order
    "k_evolve_large": ("read_b64", "scatter_b128"),
    "k_dna_evolve": ("read_b64", "scatter_b128"),
    "k_long_apply": ("read_b64", "scatter_b128"),
    "k_sort_scatter": ("read_b64", "write_b64"),
    "k_sort_hist": ("read_b64", "write_b32"),
    "k_table_insert": ("read_b64", "write_b128"),          # regions of 16-byte slots written out whole, coalesced
    "k_match": ("gather_b64x8", "write_b64"),
    "k_align_wave": ("read_b64", "write_b64"),             # history words, lane-contiguous
    "k_align_quad_rows": ("read_b128", "write_b128"),  # 16-byte history pairs, block-major
    "k_align_small": ("read_b64", "write_b64"),
    "k_emit_write": ("read_b32", "write_b64"),
    "k_emit_count": ("read_b32", "write_b32"),
    "k_dna_walk": ("read_b32", "write_b64"),
    "k_qual_symbols": ("read_b128", "write_b64"),
    "k_kmer_scan": ("read_b64", "write_b64"),
    "k_found_mask": ("gather_b64x8", "write_b32"),
    "k_lis_anchors": ("read_b64", "write_b32"),
}
CALIB = {"read_b128": "calib_read<HIP_vector_type<unsigned int, 4", "read_b64": "calib_read<HIP_vector_type<unsigned int, 2", "read_b32": "calib_read<unsigned int",
         "gather_b64x8": "calib_gather_b64x8", "write_b128": "calib_write<HIP_vector_type<unsigned int, 4", "write_b64": "calib_write<HIP_vector_type<unsigned int, 2",
         "write_b32": "calib_write<unsigned int", "scatter_b128": "calib_scatter_b128"}


def per_kernel(db_path, counter):
    """kernel name -> (launches, sum of the counter over launches); a launch may carry several rows (one per XCD / instance)."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    did = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
    out = {}
    if did:
        q = f"select kernel_name, {did}, sum(value) from counters_collection where counter_name = ? group by kernel_name, {did}"
        for name, _, v in db.execute(q, (counter,)):
            n, s = out.get(name, (0, 0.0))
            out[name] = (n + 1, s + (v or 0.0))
    else:
        for name, n, v in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
            out[name] = (n, v or 0.0)
    return out


def base_name(full):
    n = re.sub(r"^void\s+", "", full).replace("(anonymous namespace)::", "")
    b = re.split(r"[<(]", n, 1)[0]
    if b == "k_sort_scatter":                          # (with and without a value array: two different byte counts per key)
        m = re.search(r"k_sort_scatter<[^,]+,\s*(true|false)", n)
        if m:
            return f"k_sort_scatter<{m.group(1)}>"
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calib-fetch", required=True); ap.add_argument("--calib-write", required=True)
    ap.add_argument("--fetch", required=True); ap.add_argument("--write", required=True)
    ap.add_argument("--calib-bytes", type=float, required=True)
    ap.add_argument("--command", default="", help="the profiled command, for the record")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    cf, cw = per_kernel(a.calib_fetch, "FETCH_SIZE"), per_kernel(a.calib_write, "WRITE_SIZE")
    factors, raw = {}, {}
    for pat, needle in CALIB.items():
        src = cf if not pat.startswith(("write", "scatter")) else cw
        hit = [(k, v) for k, v in src.items() if needle in k]
        if not hit:
            print("calibration kernel missing:", pat, file=sys.stderr)
            continue
        n, s = hit[0][1]
        counter_bytes = s / n * 1024.0
        factors[pat] = a.calib_bytes / counter_bytes if counter_bytes else None
        raw[pat] = {"counter_KB_per_launch": s / n, "true_bytes": a.calib_bytes}
    rf, rw = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
    kernels = {}
    for full in sorted(set(rf) | set(rw)):
        b = base_name(full)
        if not b.startswith("k_"):
            continue
        e = kernels.setdefault(b, {"launches": 0, "fetch_KB": 0.0, "write_KB": 0.0})
        if full in rf:
            e["launches"] = max(e["launches"], 0) + rf[full][0]; e["fetch_KB"] += rf[full][1]
        if full in rw:
            e["write_KB"] += rw[full][1]
    for b, e in kernels.items():
        pr, pw = PATTERN.get(b.split("<")[0], ("read_b128", "write_b128"))
        # FETCH_SIZE under-reports coalesced reads (x2 on gfx950): corrected.  A WRITE_SIZE above the payload (16-byte records to
        # scattered slots: 2x) is real traffic — partial lines go out as 32-byte writes — so it is kept and reported as amplification.
        fr, cw_ = factors.get(pr) or 1.0, factors.get(pw) or 1.0
        fw = max(cw_, 1.0)
        n = max(e["launches"], 1)
        e.update({"read_pattern": pr, "write_pattern": pw, "read_factor": fr, "write_factor": fw, "write_amplification_of_pattern": round(1.0 / cw_, 3) if cw_ else None,
                  "fetch_bytes_per_launch_uncorrected": e["fetch_KB"] * 1024 / n, "write_bytes_per_launch_uncorrected": e["write_KB"] * 1024 / n,
                  "hbm_bytes_per_launch": (e["fetch_KB"] * fr + e["write_KB"] * fw) * 1024 / n})
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `{a.command}`, corrected per access pattern with tools/pmc_calib "
                         f"(MI355X_MICROARCH.md, HBM: counters are uncalibrated but for 16 B/lane reads = 1/2); see profiles/ for the raw passes",
               "calibration_factors": factors, "calibration_raw": raw, "kernels": kernels}, open(a.out, "w"), indent=1)
    print("factors:", {k: (round(v, 3) if v else None) for k, v in factors.items()})
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for b, e in top:
        print(f"{b:24s} launches {e['launches']:5d}  HBM/launch {e['hbm_bytes_per_launch'] / 1e9:8.3f} GB (fetch {e['fetch_bytes_per_launch_uncorrected'] / 1e9:.3f} x{e['read_factor']:.2f}, write {e['write_bytes_per_launch_uncorrected'] / 1e9:.3f} x{e['write_factor']:.2f})")


if __name__ == "__main__":
    main()
