#!/usr/bin/env python3
"""profiles/traffic_<config>.json from the two PMC passes over tools/spmm_traffic_run.py:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/spmm_traffic_run.py c4 > gpurun_out/pmc_fetch.log
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/spmm_traffic_run.py c4 > gpurun_out/pmc_write.log
    python tools/make_traffic_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch.log <commit>

(`c4`: writes traffic_c4.json — the R-MAT graph — and traffic_c4_uniform.json — the cache-hostile
uniform graph of bench.py's roofline_uniform; `c5`: traffic_c5.json, bf16 F = 128 through
spmm_narrow_kernel with ITS OWN calibration.)  Each file is stamped with the sha256 of
pygcn_amd/csrc/gcn_spmm.hip: bench.py reports `roofline.traffic` only while that source is unchanged
(MI355X_MICROARCH.md §HBM: FETCH_SIZE reads about half of the bytes of wide coalesced reads on gfx950
— calibrated here on a permutation-matrix launch of known byte count, in the kernel's own access
pattern and row width)."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d, name, kernel):
    """grid size -> list of counter values (KiB) of the SpMM kernel's dispatches."""
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != name or kernel not in r["Kernel_Name"]:
                continue
            out.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return out


def main():
    d_fetch, d_write, log, commit = sys.argv[1:5]
    info = None
    for ln in open(log):
        if ln.startswith("TRAFFIC_INFO "):
            info = json.loads(ln[len("TRAFFIC_INFO "):])
    assert info, "TRAFFIC_INFO line not found"
    kernel = info.get("kernel", "spmm_wide_kernel")
    fetch, write = counters(d_fetch, "FETCH_SIZE", kernel), counters(d_write, "WRITE_SIZE", kernel)

    def mean_bytes(tab, grid_x):
        v = tab.get(grid_x * 256)
        assert v, f"no dispatch with grid {grid_x * 256}: have {sorted(tab)}"
        return sum(v) / len(v) * 1024.0, v
    cal_f, cal_f_raw = mean_bytes(fetch, info["calib"]["grid_x"])
    cal_w, _ = mean_bytes(write, info["calib"]["grid_x"])
    factor = info["calib"]["fetch_bytes_expected"] / cal_f
    w_factor = info["calib"]["write_bytes_expected"] / cal_w
    src = open(os.path.join(ROOT, "pygcn_amd", "csrc", "gcn_spmm.hip"), "rb").read()
    common = {
        "dtype": info.get("dtype", "f32"), "F": info.get("F", 256), "kernel": kernel,
        "fetch_size_calibration_factor": round(factor, 4),
        "write_size_calibration_factor": round(w_factor, 4),
        "calibration": f"random permutation matrix, n={info['calib']['n']}, F={info.get('F', 256)} "
                       f"{info.get('dtype', 'f32')}: expected {info['calib']['fetch_bytes_expected']} B fetched / "
                       f"{info['calib']['write_bytes_expected']} B written; FETCH_SIZE read "
                       f"{cal_f / 1024:.0f} KiB (x1024 = {cal_f / info['calib']['fetch_bytes_expected']:.4f} "
                       f"of expected), WRITE_SIZE read {cal_w / 1024:.0f} KiB "
                       f"({cal_w / info['calib']['write_bytes_expected']:.4f} of expected)",
        "kernel_source_sha256": hashlib.sha256(src).hexdigest(), "commit": commit,
        "source": ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, over "
                   f"tools/spmm_traffic_run.py {info.get('config', 'c4')}; tools/make_traffic_json.py")}
    outputs = {"c4": [("traffic_c4", "c4_fwd", "c4_bwd"), ("traffic_c4_uniform", "c4_uniform_fwd", None)],
               "c5": [("traffic_c5", "c5_fwd", None)]}[info.get("config", "c4")]
    for fname, fwd, bwd in outputs:
        if fwd not in info:
            continue
        res = {"workload": f"{fwd}: forward SpMM ({kernel}), nnz {info[fwd]['nnz']}, n {info[fwd]['n']}, "
                           f"F {info.get('F', 256)} {info.get('dtype', 'f32')}", **common}
        for key, tag in ((fwd, ""), (bwd, "bwd_")):
            if key is None:
                continue
            f, f_raw = mean_bytes(fetch, info[key]["grid_x"])
            w, w_raw = mean_bytes(write, info[key]["grid_x"])
            res[tag + "fetch_bytes_per_launch"] = int(f * factor)
            res[tag + "write_bytes_per_launch"] = int(w * w_factor)
            res[tag + "hbm_bytes_per_launch"] = int(f * factor + w * w_factor)
            res[tag + "raw_kib"] = {"FETCH_SIZE": f_raw, "WRITE_SIZE": w_raw}
        res["algorithmic_bytes_per_launch"] = info[fwd]["algorithmic_bytes"]
        res["ratio_to_algorithmic"] = round(res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"], 4)
        json.dump(res, open(os.path.join(ROOT, "profiles", fname + ".json"), "w"), indent=1)
        print(fname, json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
