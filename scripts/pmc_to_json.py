#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of a build into the two JSON files bench.py quotes
(profiles/istft_pqmf_pmc.json, profiles/conv_mfma_pmc.json), keyed by the kernel source hash and the
launch shape so that bench.py only quotes them for the kernel / shape they were taken on.
usage: pmc_to_json.py <tag> <fetch_dir> <write_dir> <mfma_dir> <mfma_infer_dir>"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d_fetch, d_write, d_mfma, d_mfma_inf = sys.argv[1:6]


def sha(name):
    return hashlib.sha256(open(os.path.join(ROOT, "mb-istft-vits_amd", "csrc", name), "rb").read()).hexdigest()[:16]


def means(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mbv::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


B, Tp = int(os.environ.get("PROF_B", "64")), int(os.environ.get("PROF_TP", "566"))
fk = [k for k in means(d_fetch) if k.startswith("istft_pqmf_kernel")][0]
fetch_kib, nf = means(d_fetch)[fk]["FETCH_SIZE"]
write_kib, nw = means(d_write)[fk]["WRITE_SIZE"]
fetch_b, write_b = fetch_kib * 1024 * 2, write_kib * 1024      # gfx950: FETCH_SIZE reports half the bytes of a wide streaming read
json.dump({
    "kernel": fk + " waveform-only", "B": B, "Tp": Tp, "kernel_source_sha16": sha("istft_pqmf.hip"), "profile": "profiles/%s_istft_pmc_*.csv" % tag,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 scripts/prof_kernels.py istft 5 (separate passes, mean of %d dispatches)" % nf,
    "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib, "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
    "hbm_bytes_per_launch": fetch_b + write_b, "algorithmic_bytes_per_launch": 5632 * B * Tp,
}, open(os.path.join(ROOT, "profiles", "istft_pqmf_pmc.json"), "w"), indent=1)
m1, m2 = means(d_mfma), means(d_mfma_inf)
util = {"isolated launches (scripts/prof_kernels.py conv 3): " + k: round(v["MfmaUtil"][0], 2) for k, v in m1.items() if "MfmaUtil" in v and v["MfmaUtil"][0] > 1}
util.update({"inside infer (scripts/run_infer.py, B=64): " + k: round(v["MfmaUtil"][0], 2) for k, v in m2.items() if "MfmaUtil" in v and v["MfmaUtil"][0] > 1})
json.dump({
    "kernel_source_sha16": sha("conv1d.hip"), "profile": "profiles/%s_mfma_util_*.csv" % tag,
    "source": "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 scripts/prof_kernels.py conv 3 / scripts/run_infer.py ljs_mb_istft_vits 64 2 (own passes, %s build)" % tag,
    "counter": "MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100",
    "mfma_util_percent": util,
}, open(os.path.join(ROOT, "profiles", "conv_mfma_pmc.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "istft_pqmf_pmc.json")).read())
print(json.dumps(util, indent=1))
