#!/bin/bash
# PK32-OPSEL lint (tools/isa_lint.py rule) over the code objects that co-run with this library's VGPR-accumulator MFMA kernels but
# are not built here: the RCCL library csrc/collective.hip maps (torch's bundled copy first, then ROCm's) and
# the PyTorch kernels a step launches (fill / random / flip kernels: names from profiles/r04_*_kernel_stats.csv).
# Build container, no GPU; ~10 minutes.  Writes profiles/r05_foreign_codeobj_lint.json (bench.py reads it) and a text log.
#   objcopy --only-section=.hip_fatbin -> split the CCOB bundles -> clang-offload-bundler --unbundle (gfx950) -> llvm-objdump -d -> rule
cd "$(dirname "$0")/.."
T=$(python3 -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
O=gpurun_out/lint; mkdir -p $O
J=${JOBS:-4}
python3 tools/lint_foreign_codeobj.py $T/librccl.so --jobs $J --json $O/rccl_torch.json > $O/rccl_torch.log 2>&1
python3 tools/lint_foreign_codeobj.py /opt/rocm/lib/librccl.so.1 --jobs $J --json $O/rccl_rocm.json > $O/rccl_rocm.log 2>&1
python3 tools/lint_foreign_codeobj.py $T/libtorch_hip.so --jobs $J \
  --symbols FillFunctor,normal_and_transform,random_from_to,flip_kernel_impl,bernoulli,uniform_and_transform \
  --json $O/torch_hip_step_kernels.json > $O/torch_hip_step_kernels.log 2>&1
python3 - <<'PY'
import hashlib, json, os
out = {"rule": "PK32-OPSEL (tools/isa_lint.py)", "libraries": []}
for tag in ("rccl_torch", "rccl_rocm", "torch_hip_step_kernels"):
    d = json.load(open("gpurun_out/lint/%s.json" % tag))
    h = hashlib.sha256()
    with open(d["library"], "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    d["sha256"] = h.hexdigest()
    d["tag"] = tag
    out["libraries"].append(d)
json.dump(out, open("profiles/r05_foreign_codeobj_lint.json", "w"), indent=1)
for d in out["libraries"]:
    print(d["tag"], os.path.basename(d["library"]), d["bytes"], "bytes:", d["packed_fp32_instructions"], "packed-fp32 instructions,",
          d["with_op_sel"], "with an op_sel swizzle", ("in " + ", ".join(x[:80] for x in d["hit_functions"][:3])) if d["hit_functions"] else "")
PY
