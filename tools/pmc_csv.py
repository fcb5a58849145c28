"""Per-kernel averages of the counters in the rocprofv3 --pmc CSVs under a directory (tools/pmc_probe.sh).
Usage: pmc_csv.py <dir> [kernel-name substring]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k[:110])
    for c, (s, n) in sorted(cs.items()):
        print("   %-28s launches %4d  avg %.5g" % (c, n, s / n))
    g = cs.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
        cyc = g[0] / g[1] / 8.0
        m = cs["SQ_VALU_MFMA_BUSY_CYCLES"]
        print("   -> kernel cycles %.0f, MFMA busy %.1f %%" % (cyc, 100.0 * (m[0] / m[1]) / 1024.0 / cyc))
        w = cs["SQ_WAVE_CYCLES"][0] / cs["SQ_WAVE_CYCLES"][1]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c not in cs:
                continue
            print("   -> %-20s %.1f %% of wave cycles" % (c, 100.0 * cs[c][0] / cs[c][1] / w))
        if "SQ_LDS_IDX_ACTIVE" in cs:
            print("   -> LDS bank-conflict share %.1f %%" % (100.0 * cs["SQ_LDS_BANK_CONFLICT"][0] / max(cs["SQ_LDS_IDX_ACTIVE"][0], 1)))
    if "TCC_HIT_sum" in cs:
        h, m = cs["TCC_HIT_sum"][0], cs["TCC_MISS_sum"][0]
        print("   -> L2 hit rate %.1f %%" % (100.0 * h / max(h + m, 1)))
    if "FETCH_SIZE" in cs:
        print("   -> HBM fetch per launch %.1f MB (FETCH_SIZE x 2 x 1 KiB... raw avg %.5g)" % (2 * cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] / 1024.0, cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1]))
