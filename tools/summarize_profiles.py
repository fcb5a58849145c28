"""Copy the outputs of tools/profile_round.sh (gpurun_out/prof_<round>/) into profiles/ and build the PMC summaries.
usage: python tools/summarize_profiles.py gpurun_out/prof_r02c r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

O, TAG = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)


def load(d):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return acc, n


def one(pattern):
    return glob.glob(f"{O}/{pattern}", recursive=True)[0]


for src, dst in [("bench_r50.json", "bench_n1.json"), ("bench_vit.json", "bench_vit.json"),
                 ("bench_cnx.json", "bench_convnext.json"), ("layers.txt", "per_layer_table.txt"),
                 ("r02_pmc_traffic.json", "pmc_traffic.json")]:
    shutil.copy(f"{O}/{src}", f"{P}/{TAG}_{dst}")
for d, dst in [("stats_r50", "bench_n1_kernel_stats.csv"), ("stats_r50_1s", "bench_n1_kernel_stats_single_stream.csv"),
               ("stats_vit", "vit_b16_kernel_stats.csv"), ("stats_cnx", "convnext_t_mixup_ema_kernel_stats.csv")]:
    shutil.copy(one(f"{d}/**/*kernel_stats.csv"), f"{P}/{TAG}_{dst}")

acc, n = load("pmc_mfma")
out = {}
for k, v in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        frac = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
        if frac > 0.005:
            out[k] = {"launches": len(n[k]), "mfma_busy_frac": round(frac, 4)}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_frac"]))
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on `bench.py --steps 3 --warmup 1` "
                     "(single stream); busy fraction = MFMA busy cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": out}, open(f"{P}/{TAG}_pmc_mfma_busy.json", "w"), indent=1)
print("MFMA busy:", {k: v["mfma_busy_frac"] for k, v in list(out.items())[:12]})

acc, n = load("pmc_tcc")
out = {}
for k, v in acc.items():
    if "TCC_BUSY_avr" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        out[k] = {"launches": len(n[k]), "tcc_busy_frac": round(v["TCC_BUSY_avr"] / (v["GRBM_GUI_ACTIVE"] / 8.0), 4),
                  "tcc_req_per_launch": round(v.get("TCC_REQ_sum", 0) / max(len(n[k]), 1))}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["tcc_busy_frac"]))
json.dump({"source": "rocprofv3 --pmc TCC_BUSY_avr TCC_REQ_sum GRBM_GUI_ACTIVE on `bench.py --steps 3 --warmup 1` (single stream); "
                     "busy fraction = TCC_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs)", "kernels": out},
          open(f"{P}/{TAG}_pmc_l2_busy.json", "w"), indent=1)

res = {}
for v in (0, 1):
    acc, n = load(f"pmc_l2_c64_{v}")
    for k, c in acc.items():
        if "conv" in k or "slab" in k:
            L = len(n[k])
            res[f"{'register-resident / halo' if v else 'implicit GEMM'}: {k}"] = {
                "launches": L, "tcc_req_per_launch": round(c["TCC_REQ_sum"] / L), "tcc_hit_per_launch": round(c["TCC_HIT_sum"] / L),
                "tcc_miss_per_launch": round(c["TCC_MISS_sum"] / L), "cycles_per_launch": round(c["GRBM_GUI_ACTIVE"] / 8 / L)}
json.dump({"source": "rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE on `tools/one_layer.py 64 64 3 1 56 3 "
                     "fwd,dgrad,wgrad` (batch 256) with ICAMD_CONV3X3_RESIDENT / ICAMD_WGRAD_HALO = 0 and 1; the layer's operands: "
                     "103 MB in + 103 MB out = 1.6 M lines of 128 B", "kernels": res},
          open(f"{P}/{TAG}_pmc_l2_requests_64x64_3x3.json", "w"), indent=1)
d = json.loads(open(f"{P}/{TAG}_bench_n1.json").readline())
print("bench:", d["value"], d["ms_per_step"], d["roofline"])
