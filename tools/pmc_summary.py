#!/usr/bin/env python3
"""Diagnostic: condense rocprofv3 --pmc counter_collection CSVs (one directory per pass) into the JSON summaries kept under
profiles/ (HBM traffic per launch with the gfx950 FETCH_SIZE correction; MFMA-pipe / VALU / wait shares).

    python tools/pmc_summary.py traffic <fetch_dir> <write_dir>  > profiles/roundN/rN_traffic_pmc.json
    python tools/pmc_summary.py busy <dir>                        > profiles/roundN/rN_pmc_busy.json
"""
import collections
import csv
import glob
import json
import os
import sys

ENTRY = [("core_bwd_mfma_kernel", "rat_attn_core_bwd_map"), ("core_fwd_mfma_kernel", "rat_attn_core_fwd_map"), ("core_bwd_kernel", "rat_attn_core_bwd_map:valu"),
         ("attn_bwd_wide_kernel", "rat_attn_bwd_groups"), ("attn_fwd_wide_kernel", "rat_attn_fwd_groups"), ("attn_bwd3_kernel", "rat_attn_bwd_ex"), ("attn_fwd3_kernel", "rat_attn_fwd_ex"), ("ffn_bwd_t4_kernel", "rat_ffn_bwd_res"), ("ffn_bwd_t3_kernel", "rat_ffn_bwd_res:t3"),
         ("ffn_fwd_t3_kernel", "rat_ffn_fwd_res"), ("attn_bwd_kernel", "rat_attn_bwd"), ("attn_fwd_kernel", "rat_attn_fwd"),
         ("ffn_bwd_t_kernel", "rat_ffn_bwd"), ("ffn_fwd_t_kernel", "rat_ffn_fwd")]


def entry_of(kernel):
    for frag, name in ENTRY:
        if frag in kernel:
            return name
    return None


def read(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    names = {}
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            e = entry_of(r["Kernel_Name"])
            if e:
                out[e][r["Counter_Name"]].append(float(r["Counter_Value"]))
                names[e] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    return out, names


def avg(v):
    return sum(v) / len(v)


def main():
    if sys.argv[1] == "traffic":
        fetch, names = read(sys.argv[2])
        write, _ = read(sys.argv[3])
        res = {"_comment": "HBM bytes per launch from two separate rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; counters in KiB) over "
                           "tools/kbench.py at the north-star shapes (B=4096, T=11, S=21, d=64), both attention phases pooled.  gfx950 correction "
                           "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads -> read bytes = "
                           "2 x FETCH_SIZE; WRITE_SIZE is exact.", "kernels": {}}
        for e in fetch:
            f, w = avg(fetch[e]["FETCH_SIZE"]), avg(write[e]["WRITE_SIZE"])
            res["kernels"][e] = dict(kernel=names[e], FETCH_SIZE_KiB_avg=round(f, 1), WRITE_SIZE_KiB_avg=round(w, 1),
                                     read_bytes_corrected=int(2 * f * 1024), write_bytes=int(w * 1024),
                                     hbm_bytes_per_launch=int(2 * f * 1024 + w * 1024), launches=len(fetch[e]["FETCH_SIZE"]))
        for alias, src in (("rat_attn_bwd_ex:f32", "rat_attn_bwd"),):
            pass
        print(json.dumps(res, indent=1))
    elif sys.argv[1] == "coexec":
        data, names = read(sys.argv[2])
        res = {"_comment": "rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY "
                           "GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 "
                           "(north-star shapes; the program directly after --).  coexec_frac = cycles in which a VALU and an MFMA instruction "
                           "execute together / (1024 SIMDs x kernel cycles); the other fractions as in the pmc_busy file.", "kernels": {}}
        for e, c in data.items():
            cyc = avg(c["GRBM_GUI_ACTIVE"]) / 8.0
            busy, co = avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(c["SQ_VALU_MFMA_COEXEC_CYCLES"])
            res["kernels"][e] = dict(kernel=names[e], launches=len(c["GRBM_GUI_ACTIVE"]), kernel_cycles=int(cyc),
                                     mfma_busy_frac=round(busy / (1024 * cyc), 3), coexec_frac=round(co / (1024 * cyc), 3),
                                     coexec_over_mfma_busy=round(co / busy, 3) if busy else None,
                                     valu_active_frac=round(4 * avg(c["SQ_ACTIVE_INST_VALU"]) / (1024 * cyc), 3),
                                     wait_any_frac=round(avg(c["SQ_WAIT_ANY"]) / avg(c["SQ_WAVE_CYCLES"]), 3))
        print(json.dumps(res, indent=1))
    else:
        data, names = read(sys.argv[2])
        res = {"_comment": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY "
                           "GRBM_GUI_ACTIVE over tools/kbench.py (north-star shapes).  SQ_* wave counters are in quad-cycles, "
                           "SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs.  mfma_busy_frac = "
                           "MFMA busy cycles / (1024 SIMDs x kernel cycles); valu_active_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 x kernel cycles); "
                           "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", "kernels": {}}
        for e, c in data.items():
            cyc = avg(c["GRBM_GUI_ACTIVE"]) / 8.0
            res["kernels"][e] = dict(kernel=names[e], kernel_cycles=int(cyc), **{k: float("%.4g" % avg(v)) for k, v in c.items()},
                                     mfma_busy_frac=round(avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * cyc), 3),
                                     valu_active_frac=round(4 * avg(c["SQ_ACTIVE_INST_VALU"]) / (1024 * cyc), 3),
                                     wait_frac=round(avg(c["SQ_WAIT_ANY"]) / avg(c["SQ_WAVE_CYCLES"]), 3))
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
