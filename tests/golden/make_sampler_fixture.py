"""Generates tests/golden/ra_sampler.json by RUNNING the reference's RASampler (/root/reference/utils.py:17-63) in this
container (same stand-in modules for the absent third-party imports as make_engine_fixture.py).  Only the resulting
index lists travel; run from the repo root:  python tests/golden/make_sampler_fixture.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_engine_fixture import REF, install_standins  # noqa: E402


def main():
    install_standins()
    sys.path.insert(0, REF)
    import utils   # the reference's utils.py, unmodified
    cases = []
    for (n, replicas, shuffle) in [(600, 1, True), (600, 2, True), (1000, 4, True), (777, 2, False), (5000, 8, True)]:
        for rank in sorted({0, replicas - 1}):
            for epoch in (0, 3):
                s = utils.RASampler(list(range(n)), num_replicas=replicas, rank=rank, shuffle=shuffle)
                s.set_epoch(epoch)
                idx = list(iter(s))
                cases.append({"n": n, "replicas": replicas, "rank": rank, "epoch": epoch, "shuffle": shuffle,
                              "len": len(s), "indices": idx})
    with open(os.path.join(HERE, "ra_sampler.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_sampler_fixture.py", "cases": cases}, f)
    print("wrote ra_sampler.json:", [(c["n"], c["replicas"], c["rank"], c["len"]) for c in cases])


if __name__ == "__main__":
    main()
