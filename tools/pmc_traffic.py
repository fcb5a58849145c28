"""Builds profiles/<name>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes: the two do not fit one TCC pass) of the same bench command.  rocprofv3 reports both
derived counters in KiB; on gfx950 FETCH_SIZE additionally counts 128 B requests as 64 B, so it is doubled (guide, HBM
section).  Sanity anchor: the BatchNorm kernels' result must equal their algorithmic bytes (profiles/README.md).

The summary is stamped with the workload and with bench.kernel_source_hash() of the tree it is run from (run it from the
snapshot that was profiled): bench.py reports `roofline.traffic` from it only while both match.

Usage: python tools/pmc_traffic.py <fetch.db> <write.db> <steps_in_run> <out.json> [arch] [batch] [train|eval]"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection"
    assert view in tabs, tabs
    cols = [d[0] for d in db.execute("select * from %s limit 1" % view).description]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    q = "select %s, count(distinct dispatch_id), sum(value) from %s where counter_name = ? group by 1" % (kn, view)
    for name, n, total in db.execute(q, (counter,)):
        short = re.sub(r"^void ", "", name)
        short = re.sub(r"\(anonymous namespace\)::", "", short)
        short = re.sub(r"\(.*$", "", short)
        out[short] = (n, total)
    return out


def main():
    fetch_db, write_db, steps, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    arch = sys.argv[5] if len(sys.argv) > 5 else "resnet50"
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 256
    mode = sys.argv[7] if len(sys.argv) > 7 else "train"
    from bench import kernel_source_hash
    # units: rocprofv3's derived FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 correction x2 on FETCH_SIZE (guide, HBM section)
    fetch = per_kernel(fetch_db, "FETCH_SIZE")
    write = per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    tf = tw = 0.0
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, (0, 0.0))
        nw, w = write.get(k, (0, 0.0))
        n = max(nf, nw)
        fb, wb = f * 1024.0 * 2.0, w * 1024.0
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1),
                      "fetch_GB_per_step": fb / steps / 1e9, "write_GB_per_step": wb / steps / 1e9}
        tf += fb / steps / 1e9
        tw += wb / steps / 1e9
    kernels = dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_GB_per_step"] + kv[1]["write_GB_per_step"])))
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) via tools/pmc_traffic.py; %d steps in the "
                     "profiled run; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
                     "bytes" % steps,
           "workload": {"arch": arch, "batch": batch, "mode": mode}, "kernel_source_hash": kernel_source_hash(),
           "kernels": kernels, "total_fetch_GB_per_step": tf, "total_write_GB_per_step": tw}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print("fetch %.1f GB/step, write %.1f GB/step, %d kernels" % (tf, tw, len(kernels)))


if __name__ == "__main__":
    main()
