"""Copy the outputs of tools/profile_round.sh (gpurun_out/prof_<round>/, parts A and B) into profiles/ and build the PMC summaries.
usage: python tools/summarize_profiles.py gpurun_out/prof_r04 r04"""
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


def copy(src, dst):
    if os.path.exists(f"{O}/{src}"):
        shutil.copy(f"{O}/{src}", f"{P}/{TAG}_{dst}")
        print("copied", dst)
    else:
        print("MISSING", src)


for src, dst in [("bench_r50.json", "bench_n1.json"), ("bench_vit.json", "bench_vit.json"), ("bench_cnx.json", "bench_convnext.json"),
                 ("bench_eval.json", "bench_eval.json"), ("pmc_traffic_eval.json", "pmc_traffic_eval.json"),
                 ("layers.txt", "per_layer_table.txt"), ("gemm_shapes.txt", "gemm_shapes.txt"), ("dwconv.txt", "dwconv.txt"), ("cnx_layers.txt", "convnext_linear_layers.txt"), ("attention.txt", "attention.txt"),
                 ("fused_bwd.txt", "fused_conv_bn_bwd.txt"), ("fused_fwd.txt", "fused_bn_apply_conv.txt"), ("ls_fuse.txt", "convnext_layerscale_fold.txt"), ("wgrad_shapes.txt", "wgrad_shapes.txt"),
                 ("pmc_traffic_r50.json", "pmc_traffic.json"), ("pmc_traffic_vit.json", "pmc_traffic_vit.json"),
                 ("pmc_traffic_cnx.json", "pmc_traffic_convnext.json")]:
    copy(src, dst)
for d, dst in [("stats_r50", "bench_n1_kernel_stats.csv"), ("stats_r50_1s", "bench_n1_kernel_stats_single_stream.csv"),
               ("stats_vit", "vit_b16_kernel_stats.csv"), ("stats_cnx", "convnext_t_mixup_ema_kernel_stats.csv"),
               ("stats_eval", "eval_kernel_stats.csv")]:
    try:
        shutil.copy(one(f"{d}/**/*kernel_stats.csv"), f"{P}/{TAG}_{dst}")
        print("copied", dst)
    except IndexError:
        print("MISSING", d)

for d, dst, what in [("pmc_mfma", "pmc_mfma_busy.json", "bench.py"), ("pmc_mfma_vit", "pmc_mfma_busy_vit.json", "bench.py --arch vit_base_patch16_224"),
                     ("pmc_mfma_cnx", "pmc_mfma_busy_convnext.json", "bench.py --arch convnext_tiny --mixup"),
                     ("pmc_mfma_eval", "pmc_mfma_busy_eval.json", "bench.py --mode eval")]:
    try:
        acc, n = load(d)
    except IndexError:
        print("MISSING", d)
        continue
    out = {}
    for k, v in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            frac = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
            if frac > 0.005:
                out[k] = {"launches": len(n[k]), "mfma_busy_frac": round(frac, 4)}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_frac"]))
    json.dump({"source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on `{what} --steps 3 --warmup 1` "
                         "(single stream); busy fraction = MFMA busy cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
               "kernels": out}, open(f"{P}/{TAG}_{dst}", "w"), indent=1)
    print(dst, {k: v["mfma_busy_frac"] for k, v in list(out.items())[:10]})
for f in ("bench_n1.json", "bench_vit.json", "bench_convnext.json", "bench_eval.json"):
    try:
        d = json.loads(open(f"{P}/{TAG}_{f}").readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"])
    except (OSError, ValueError) as e:
        print(f, "unreadable", e)
