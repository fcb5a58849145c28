"""Headline benchmark: images/sec (whole node) of ResNet-50 bf16 training at 224x224, batch 256 per GPU
(BASELINE.json configs[1]; N>1 = configs[2], one process per GPU over RCCL/xGMI, weak scaling).

A "step" is one pass of the hot path (engine.train_one_epoch body: input pack, forward, label-smoothed loss,
backward, gradient all-reduce, fused AdamW, device-side metrics) over one synthetic batch already resident in
HBM.  Prints ONE JSON line on rank 0.  `roofline` is for the kernel class that takes the most time of ALL classes,
timed live with HIP events on the launch stream; `roofline_classes` carries every class (time, calls, algorithmic
bytes and flops booked by the C ABI per call, both roof fractions), `roofline_step` the blended whole step,
`host_enqueue_ms` the Python / launch time one step needs with no device wait; `cpu_baseline` is the CPU oracle of
the same step (torch-CPU fp32 restatement of the reference loop) timed on this box's host cores on a bounded sample.

`--mode eval` measures the other half of the named hot path the same way: engine.evaluate (reference
engine.py:145-225) over device-resident batches of int(1.5 * 256) = 384 images (the reference's eval batch,
train.py:167), BatchNorm folded into the filters.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def conv_flops_per_image(net, hw):
    """Algorithmic MACs*2 of every conv/FC forward at true (unpadded) channel counts."""
    h = w = hw
    layers = []
    layer_bytes = []

    def add(conv, ih, iw):
        oh = (ih + 2 * conv.pad - conv.k) // conv.stride + 1
        ow = (iw + 2 * conv.pad - conv.k) // conv.stride + 1
        layers.append((conv.name, 2 * oh * ow * conv.cout * conv.cin * conv.k * conv.k))
        # algorithmic HBM bytes: bf16 input (stem: RGB stored zero-padded to 8 channels), bf16 output, filter elements
        cin_mem = getattr(conv, "cin_p", max(conv.cin, 8))   # channels as stored (stem: rgb4 layout)
        layer_bytes.append((2 * ih * iw * cin_mem, 2 * oh * ow * conv.cout, conv.cout * cin_mem * conv.k * conv.k))
        return oh, ow

    h, w = add(net.stem_conv, h, w)
    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dgrad_extra = 0   # bytes per image the data-gradient class moves beyond in + out + filter: the BatchNorm input y read by the
                      # launches that carry the previous block's BatchNorm-backward reduction (icamd_conv2d_dgrad_bnred)
    for bi, blk in enumerate(net.blocks):
        ih, iw = h, w
        if "down_conv" in blk:
            add(blk["down_conv"], h, w)
        c1 = blk["convs"][0]
        if (net.block == "bottleneck" and bi > 0 and "down_conv" not in net.blocks[bi - 1] and c1.k == 1 and c1.stride == 1
                and c1.cout in (64, 128, 256) and c1.cin % 256 == 0):
            dgrad_extra += 2 * h * w * c1.cin
        for conv in blk["convs"]:
            ih, iw = add(conv, ih, iw)
        h, w = ih, iw
    add(net.fc, 1, 1)
    conv_flops_per_image.layer_bytes = layer_bytes
    conv_flops_per_image.dgrad_extra = dgrad_extra
    return layers


def vit_flops_per_image(net):
    """Algorithmic forward FLOPs of every Linear (as [name, flops]) plus the attention products."""
    T, D, Hd = net.T, net.dim, net.hidden
    layers = [("patch_embed", 2 * (T - 1) * D * 3 * net.patch * net.patch)]
    for i in range(net.depth):
        layers += [(f"blocks.{i}.qkv", 2 * T * D * 3 * D), (f"blocks.{i}.proj", 2 * T * D * D),
                   (f"blocks.{i}.fc1", 2 * T * D * Hd), (f"blocks.{i}.fc2", 2 * T * Hd * D),
                   (f"blocks.{i}.attention", 4 * net.heads * T * T * 64)]
    layers.append(("head", 2 * D * net.num_classes))
    # algorithmic bytes per GEMM (bf16 input rows + output rows, filter elements); attention: qkv in, out
    lb = [(2 * 224 * 224 * 8, 2 * (T - 1) * D, D * 8 * net.patch * net.patch)]
    for _ in range(net.depth):
        lb += [(2 * T * D, 2 * T * 3 * D, 3 * D * D), (2 * T * D, 2 * T * D, D * D), (2 * T * D, 2 * T * Hd, D * Hd),
               (2 * T * Hd, 2 * T * D, D * Hd), (2 * T * 3 * D, 2 * T * D, 0)]
    lb.append((2 * D, 2 * net.num_classes, D * net.num_classes))
    conv_flops_per_image.layer_bytes = lb
    return layers


def convnext_flops_per_image(net, hw):
    """Algorithmic forward FLOPs of the GEMM-shaped layers (stem, downsample, pointwise MLPs, head); the depthwise
    stencils (2*49 flop per output element) are listed separately as `dwconv`."""
    layers, dw, lb = [], 0, []
    h = w = hw // 4
    layers.append(("stem", 2 * h * w * net.dims[0] * 3 * 16))
    lb.append((2 * hw * hw * 8, 2 * h * w * net.dims[0], net.dims[0] * 8 * 16))
    for si, st in enumerate(net.stages):
        dim = st["dim"]
        if si > 0:
            lb.append((2 * h * w * net.dims[si - 1], 2 * (h // 2) * (w // 2) * dim, dim * net.dims[si - 1] * 4))
            h, w = h // 2, w // 2
            layers.append((f"stages.{si}.downsample", 2 * h * w * dim * net.dims[si - 1] * 4))
        for blk in st["blocks"]:
            layers.append((blk["name"] + ".mlp", 2 * 2 * h * w * dim * 4 * dim))
            lb += [(2 * h * w * dim, 2 * h * w * 4 * dim, 4 * dim * dim), (2 * h * w * 4 * dim, 2 * h * w * dim, 4 * dim * dim)]
            dw += 2 * 49 * h * w * dim
    layers.append(("head", 2 * net.dims[-1] * net.num_classes))
    lb.append((2 * net.dims[-1], 2 * net.num_classes, net.dims[-1] * net.num_classes))
    layers.append(("dwconv", dw))
    conv_flops_per_image.layer_bytes = lb   # algorithmic bytes of the GEMM-shaped layers (input, output, filter elements)
    return layers


# every kernel a C-ABI call of the class can launch, by name prefix (csrc/capi.hip routing), for the PMC traffic figures.
# Kernels shared between classes (conv_igemm serves forward and data gradient; the fp32 slab fold serves every weight
# gradient) are listed under every class they serve and the traffic is then reported for the GROUP of classes that share.
_FD = ("conv_igemm_kernel", "conv1x1_resident_kernel", "conv3x3_halo_kernel", "conv3x3_c64_resident_kernel",
       "stem7x7s2_resident_kernel", "gemm_nt_kernel", "gemm_nt_8phase_kernel")
CLASS_KERNELS = {
    "conv_fwd": _FD, "conv_dgrad": _FD,
    "conv_wgrad": ("conv_wgrad_kernel", "conv_wgrad_ring_kernel", "conv3x3_wgrad_halo_kernel", "stem7x7s2_wgrad_resident_kernel",
                   "slab_reduce_kernel"),
    "bn_finalize": ("bn_reduce_finalize_kernel", "bn_eval_coeffs", "bn_fold"),
    "bn_apply": ("bn_apply_kernel", "bn_relu_maxpool"),
    "bn_bwd": ("bn_bwd_",),
    "pool": ("maxpool3x3s2_", "avgpool_"),
    "pack": ("pack_input",),
    "loss": ("softmax_xent_kernel", "step_metrics_kernel"),
    "optimizer": ("adamw_ema_kernel", "optim_ema_kernel", "filter_transpose", "sumsq_partial_kernel", "gradnorm_finalize_kernel",
                  "lerp_kernel", "f32_to_bf16_kernel", "grad_guard"),
    "misc": ("colsum_kernel", "colsum_partial_kernel", "vit_tokens_fwd_kernel", "batch_sum_kernel", "strided_rows_copy_kernel"),
    "attn_fwd": ("attn_fwd_kernel",), "attn_bwd": ("attn_bwd_",),
    "ln_fwd": ("layernorm_fwd_kernel",), "ln_bwd": ("layernorm_bwd_kernel",),
    "elementwise": ("layerscale_", "gelu_fwd_kernel", "gelu_bwd_kernel"),
    "dwconv": ("dwconv7_",),
    # round 5: BatchNorm-backward apply + data gradient + weight gradient of a bottleneck's conv3 in one launch (its finalize and
    # slab fold are launched by the same C-ABI call and counted in its time; they carry the bn_bwd_ / slab_reduce names)
    "conv_bn_bwd_fused": ("conv1x1_bn_bwd_fused_kernel",),
    # round 5: BatchNorm apply + shortcut + ReLU of a block's end and the next block's 1x1 convolution (+ statistics) in one launch
    "bn_apply_conv_fused": ("bn_apply_conv1x1_fused_kernel",),
}
MFMA_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "attn_fwd", "attn_bwd", "conv_bn_bwd_fused", "bn_apply_conv_fused")


def kernel_source_hash():
    """sha256 over the kernel sources WITHOUT comments and whitespace: PMC summaries under profiles/ are stamped with it
    (tools/pmc_traffic.py) so that a traffic figure measured on older kernels is never reported for newer ones, while an edit
    of a comment does not orphan the measurement."""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "imageclassification_amd", "csrc")
    strip = re.compile(r'//[^\n]*|/\*.*?\*/|("(?:\\.|[^"\\])*")', re.S)
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(csrc, f), encoding="utf-8", errors="replace").read()
            text = strip.sub(lambda m: m.group(1) or " ", text)      # comments out, string literals kept
            h.update(f.encode())
            h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def foreign_codeobj_guard(lib):
    """The op_sel hazard guard carried to what co-runs with the step (DESIGN.md section 5, round-4 finding 1):
    the RCCL library this process has MAPPED (the gradient all-reduce's kernels share SIMDs with the MFMA kernels at N > 1) is
    compared -- path, size, sha256 -- with the libraries tools/lint_foreign_codeobj.sh disassembled and linted in the build
    container (profiles/r05_foreign_codeobj_lint.json).  Returns {rccl_version, rccl_library, verdict}: "clean" only for a byte-
    identical library with 0 op_sel hits; anything else says why it is unverified (a different RCCL / torch build on the box)."""
    import hashlib
    info = {"rccl_version": int(lib.icamd_rccl_version()), "rccl_library": None, "verdict": "unverified: no librccl mapped"}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r05_foreign_codeobj_lint.json")))
    except (OSError, ValueError):
        info["verdict"] = "unverified: no lint record (tools/lint_foreign_codeobj.sh)"
        return info
    mapped = []
    try:
        for line in open("/proc/self/maps"):
            path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
            if "librccl" in os.path.basename(path) and path not in mapped:
                mapped.append(path)
    except OSError:
        pass
    if not mapped:
        return info
    path = os.path.realpath(mapped[0])
    info["rccl_library"] = path
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    for d in rec.get("libraries", []):
        if d.get("sha256") == h.hexdigest():
            info["verdict"] = ("clean: %d packed-fp32 instructions, 0 with an op_sel swizzle (linted copy, sha256 match)" % d["packed_fp32_instructions"]
                               if d.get("with_op_sel") == 0 else "HAZARD: %d packed-fp32 instructions with an op_sel swizzle" % d["with_op_sel"])
            return info
    info["verdict"] = "unverified: the mapped librccl differs from the linted copies (re-run tools/lint_foreign_codeobj.sh on this image)"
    return info


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N rank processes ourselves, the way the reference's
    README starts multi-GPU training (`torchrun --nproc_per_node=N train.py`, /root/reference/README.md:21; rendezvous
    variables read as in utils.py:339-375).  The ranks are FRESH child processes of torch.distributed.run, started before this
    process has imported torch or touched the GPU; this process never does either: it relays the children's output (rank 0
    prints the JSON line) and exits with the launcher's return code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"no WORLD_SIZE in the environment: launching {n} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU and step (default 256; 384 with --mode eval)")
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--mode", default="train", choices=["train", "eval"],
                    help="train: engine.train_one_epoch (the headline); eval: engine.evaluate on batches of 384")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mixup", action="store_true", help="mixup 0.8 + cutmix 1.0 + EMA (BASELINE configs[4] recipe)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--pmc-file", default=None, help="PMC traffic summary under profiles/ (default: this round's file for --arch)")
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic batches resident in HBM (SURVEY 8d: K >= 8)")
    ap.add_argument("--transport", default=None, choices=["torch", "rccl"],
                    help="gradient transport at N > 1: torch.distributed's RCCL process group (default) or the library's own "
                         "RCCL communicator (icamd_allreduce_bucket_launch)")
    args = ap.parse_args()
    is_eval = args.mode == "eval"
    if args.batch is None:
        args.batch = 384 if is_eval else 256     # reference train.py:167: the eval loader uses int(1.5 * batch_size)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    global torch, dist
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback for the product path")
    if os.environ.get("ICAMD_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm); ICAMD_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
        backend = os.environ.get("ICAMD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from imageclassification_amd import hip
    from imageclassification_amd.ddp import DistributedDataParallel
    from imageclassification_amd import engine
    from imageclassification_amd.engine import evaluate, train_one_epoch
    from imageclassification_amd.mixup import LabelSmoothingCrossEntropy
    from imageclassification_amd.nets import ResNet
    from imageclassification_amd.vit import CONFIGS as VIT_CONFIGS, VisionTransformer
    from imageclassification_amd.convnext import CONFIGS as CNX_CONFIGS, ConvNeXt
    from imageclassification_amd.ema import ModelEmaV3
    from imageclassification_amd.mixup import Mixup, SoftTargetCrossEntropy
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount, cosine_scheduler

    lib = hip.load()
    C, B, HW = 1000, args.batch, args.hw
    is_vit = args.arch in VIT_CONFIGS
    is_cnx = args.arch in CNX_CONFIGS
    if is_vit:
        net = VisionTransformer(args.arch, C, device=str(device), img_size=HW, seed=88)
    elif is_cnx:
        net = ConvNeXt(args.arch, C, device=str(device), drop_path_rate=0.05, seed=88)   # reference --drop_path default
    else:
        net = ResNet(args.arch, C, device=str(device), seed=88)
    model = DistributedDataParallel(net, transport=args.transport) if world > 1 else net
    # ranks in the communicator that carries the gradients, asked of the live RCCL communicator (ncclCommCount) at N > 1
    ranks_seen = model.reducer.ranks_seen() if world > 1 else 1
    guard = None
    if world > 1:
        log(f"gradient transport: {model.reducer.transport}, communicator reports {ranks_seen} ranks, "
            f"{len(model.reducer.buckets)} buckets")
        if rank == 0:     # the packed-fp32 op_sel hazard guard for the RCCL kernels that co-run with the step
            guard = foreign_codeobj_guard(lib)
            log(f"RCCL version {guard['rccl_version']} ({guard['rccl_library']}): op_sel lint {guard['verdict']}")
    opt = create_optimizer("adamw", 1e-3, 5e-4, net)
    crit = LabelSmoothingCrossEntropy(0.1)
    mixup_fn, model_ema = None, None
    if args.mixup and not is_eval:   # BASELINE configs[4]: mixup 0.8 + cutmix 1.0 (soft targets) + model EMA 0.9995
        import numpy as np
        np.random.seed(88 + rank)
        mixup_fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
        crit = SoftTargetCrossEntropy()
        model_ema = ModelEmaV3(net, decay=0.9995)
    total_steps = args.warmup + args.steps
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        lr = cosine_scheduler(1e-3, 1e-6, 1, total_steps, warmup_epochs=0)
        wd = cosine_scheduler(5e-4, 5e-6, 1, total_steps)
    # synthetic pool, device resident (reference seed convention: 88 + rank, train.py:83,116)
    g = torch.Generator(device=device).manual_seed(88 + rank)
    pool = [(torch.randn(B, 3, HW, HW, generator=g, device=device),
             torch.randint(0, C, (B,), generator=g, device=device)) for _ in range(max(1, args.pool))]

    def run(nsteps, start):
        loader = [pool[i % len(pool)] for i in range(nsteps)]
        with contextlib.redirect_stdout(sink):
            if is_eval:
                return evaluate(loader, model, device, C, use_amp=True)
            return train_one_epoch(model, crit, loader, opt, device, 0, NativeScalerWithGradNormCount(), None, model_ema, mixup_fn,
                                   start_steps=start, lr_schedule_values=lr, wd_schedule_values=wd,
                                   num_training_steps_per_epoch=nsteps, update_freq=1, use_amp=True, num_classes=C)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model + {len(pool)} synthetic batches resident; warm-up {args.warmup} steps")
    if args.warmup:
        run(args.warmup, 0)
    log("warm-up done; timing")
    barrier()
    t0 = time.perf_counter()
    stats = run(args.steps, args.warmup)
    barrier()
    dt = time.perf_counter() - t0
    # host side of the same K steps: the loop enqueues them back to back and waits once at its end (engine.py).  Over 20 steps
    # this figure is NOT the host's own cost: the HIP queue takes ~2 700 outstanding launches (tools/host_launch_probe.py), a
    # host that is faster than the device runs ~5 steps ahead and then blocks inside the launch calls at the device's pace.
    enqueue_timed_s = (evaluate if is_eval else train_one_epoch).last_enqueue_s
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = world * B * args.steps / dt
    log(f"timed {args.steps} steps: {1e3 * dt / args.steps:.2f} ms/step, {value:.1f} img/s")

    # host_enqueue_ms: a burst of 3 steps from an idle GPU (fewer launches than the queue holds, so no launch call blocks): the
    # time until the loop reaches its single wait = Python + ctypes + HIP launch cost of a step, with the device never waited on
    nburst = min(3, args.steps)
    barrier()
    run(nburst, 0)
    enqueue_s = (evaluate if is_eval else train_one_epoch).last_enqueue_s / nburst
    barrier()
    log(f"host enqueue {1e3 * enqueue_s:.2f} ms/step (burst of {nburst} from an idle GPU)")

    # Second pass over the same workload with HIP events recorded around every C-ABI call on the launch stream
    # (icamd_prof_*): per-kernel-class durations and the algorithmic bytes / flops the C ABI books per call.  Kept out of
    # the headline timing because the ~1500 event records per step cost ~10 % of the step.
    prof_steps = min(args.steps, 10)
    # The timed region above runs the weight gradients on a second stream beside the main chain; co-running kernels
    # stretch one another, so for per-kernel rates this pass keeps everything on ONE stream (the rocprofv3 summaries of
    # both modes are in profiles/).
    net.wgrad_side_stream = False
    hip.prof_collect()
    lib.icamd_prof_enable(1)
    barrier()
    t1 = time.perf_counter()
    run(prof_steps, 0)
    barrier()
    dt_prof = time.perf_counter() - t1
    lib.icamd_prof_enable(0)
    net.wgrad_side_stream = True
    prof = hip.prof_collect(work=True)
    psteps = prof_steps

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        # ---- every class: time, calls, algorithmic work (booked per call by the C ABI), both roof fractions
        classes = {}
        for k, (ms, calls, nbytes, flops) in prof.items():
            if not calls:
                continue
            sec = ms * 1e-3 / psteps
            c = {"ms_per_step": round(ms / psteps, 3), "calls_per_step": calls / psteps,
                 "algorithmic_GB_per_step": round(nbytes / psteps / 1e9, 3),
                 "GBps": round(nbytes / psteps / sec / 1e9, 1) if sec > 0 else None,
                 "hbm_frac": round(nbytes / psteps / sec / 1e9 / PEAK_HBM_GBS, 4) if sec > 0 else None}
            if flops:
                c["algorithmic_TFLOP_per_step"] = round(flops / psteps / 1e12, 4)
                c["tflops"] = round(flops / psteps / sec / 1e12, 1)
                c["mfma_frac"] = round(flops / psteps / sec / 1e12 / PEAK_BF16_TFLOPS, 4)
            c["bound"] = "mfma" if (k in MFMA_CLASSES and c.get("mfma_frac", 0) > (c["hbm_frac"] or 0)) else "hbm"
            classes[k] = c
        dom = max(classes, key=lambda k: classes[k]["ms_per_step"])

        # ---- HBM bytes from the committed rocprofv3 --pmc passes (profiles/): only used when the summary was collected for THIS
        # workload on THESE kernel sources (hash stamp), else null + a note
        pmc, traffic_note = None, None
        default_pmc = {"resnet50": "r05_pmc_traffic.json", "vit_base_patch16_224": "r05_pmc_traffic_vit.json",
                       "convnext_tiny": "r05_pmc_traffic_convnext.json"}.get(args.arch, "r05_pmc_traffic.json")
        if is_eval:
            default_pmc = "r05_pmc_traffic_eval.json"
        pmc_file = args.pmc_file or default_pmc
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            wl = doc.get("workload", {})
            if wl.get("arch") != args.arch or wl.get("batch") != B or wl.get("mode", "train") != args.mode:
                traffic_note = f"{pmc_file} was collected for another workload"
            elif doc.get("kernel_source_hash") != kernel_source_hash():
                traffic_note = (f"stale: {pmc_file} was collected on kernel sources {doc.get('kernel_source_hash')}, "
                                f"this build is {kernel_source_hash()}")
            else:
                pmc = doc
        except (OSError, KeyError, ValueError) as e:
            traffic_note = f"no usable PMC summary ({type(e).__name__})"

        def pmc_bytes(names):
            return sum(v["fetch_GB_per_step"] + v["write_GB_per_step"] for k, v in pmc["kernels"].items() if k.startswith(names)) * 1e9

        # the classes whose kernels the dominant class shares (forward + data gradient): PMC bytes are per kernel NAME, so the
        # per-launch traffic is the average over the launches of the whole group -- said in traffic_scope, not hidden
        group = [k for k in classes if CLASS_KERNELS.get(k) == CLASS_KERNELS.get(dom)]
        traffic = None
        if pmc is not None:
            per_step = pmc_bytes(CLASS_KERNELS[dom])
            calls = sum(classes[k]["calls_per_step"] for k in group)
            traffic = round(per_step / calls) if calls else None
            for k, c in classes.items():
                if [g_ for g_ in classes if CLASS_KERNELS.get(g_) == CLASS_KERNELS.get(k)] == [k]:
                    c["pmc_GB_per_step"] = round(pmc_bytes(CLASS_KERNELS[k]) / 1e9, 3)
        d = classes[dom]
        by_flops = d["bound"] == "mfma"
        per_launch = (prof[dom][3] if by_flops else prof[dom][2]) / prof[dom][1]
        roofline = {"kernel": dom, "bound": d["bound"],
                    "achieved": d["tflops"] if by_flops else d["GBps"], "peak": PEAK_BF16_TFLOPS if by_flops else PEAK_HBM_GBS,
                    "unit": "TFLOP/s" if by_flops else "GB/s", "frac": d["mfma_frac"] if by_flops else d["hbm_frac"],
                    "traffic": traffic, "traffic_scope": "+".join(group),
                    "avg_launch_us": round(1e3 * d["ms_per_step"] / d["calls_per_step"], 2),
                    ("algorithmic_gflop_per_launch" if by_flops else "algorithmic_mbytes_per_launch"):
                        round(per_launch / (1e9 if by_flops else 1e6), 3),
                    "hbm_frac": d["hbm_frac"], "mfma_frac": d.get("mfma_frac")}
        if traffic_note:
            roofline["traffic_note"] = traffic_note

        # ---- the whole step against both roofs (SURVEY 8d "blended"): algorithmic work of all classes over the TIMED step
        tot_bytes = sum(v[2] for v in prof.values()) / psteps
        tot_flops = sum(v[3] for v in prof.values()) / psteps
        step = {"ms_per_step": round(ms_step, 3), "algorithmic_GB": round(tot_bytes / 1e9, 2),
                "GBps": round(tot_bytes / (ms_step * 1e-3) / 1e9, 1), "hbm_frac": round(tot_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "algorithmic_TFLOP": round(tot_flops / 1e12, 3), "tflops": round(tot_flops / (ms_step * 1e-3) / 1e12, 1),
                "mfma_frac": round(tot_flops / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "pmc_GB": None, "kernel_ms_single_stream": round(sum(c["ms_per_step"] for c in classes.values()), 3),
                "note": "algorithmic_GB = every operand of every C-ABI call once (as the step is built: unfused passes count); "
                        "pmc_GB = rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of all kernels of a step (profiles/)"}
        if pmc is not None:
            step["pmc_GB"] = round(pmc["total_fetch_GB_per_step"] + pmc["total_write_GB_per_step"], 2)
            step["pmc_over_algorithmic"] = round(step["pmc_GB"] / step["algorithmic_GB"], 3)
        if not (is_vit or is_cnx or is_eval):
            # SURVEY 8(d)'s ideal-fusion figure for ResNet: 24 B per conv-output element + 36 B per parameter
            layers = conv_flops_per_image(net, HW)
            elems = sum(o for _, o, _ in conv_flops_per_image.layer_bytes[:-1]) / 2
            step["survey_ideal_GB"] = round((24 * elems * B + 36 * net.n_params) / 1e9, 2)

        cpu = None
        if not args.no_cpu_baseline and world == 1 and not (is_vit or is_cnx):
            from oracle.engine_ref import time_cpu_eval, time_cpu_training
            ncpu = host_cores()
            log(f"cpu baseline on {ncpu} threads ...")
            with contextlib.redirect_stdout(sink):
                ips, threads, sps = (time_cpu_eval if is_eval else time_cpu_training)(
                    args.arch, 32, HW, C, warmup=args.cpu_warmup, steps=args.cpu_steps, threads=ncpu)
            what = ("engine.py evaluate" if is_eval else "engine.py train_one_epoch")
            cpu = {"value": round(ips, 2), "unit": "images/sec", "cores": threads, "kind": "port",
                   "cpu_model": cpu_model_string(),
                   "sample": f"{args.cpu_steps} timed steps of batch 32 after {args.cpu_warmup} warm-up steps, torch-CPU fp32 "
                             f"restatement of {what}, {args.arch} {HW}x{HW}" + ("" if is_eval else ", AdamW+label smoothing")}
        label = {"resnet50": "ResNet-50", "vit_base_patch16_224": "ViT-B/16", "convnext_tiny": "ConvNeXt-T"}.get(args.arch, args.arch)
        if args.mixup and not is_eval:
            label += " + mixup/cutmix + EMA"
        cfg_i = 3 if is_vit else (4 if is_cnx else (1 if world == 1 else 2))
        if is_eval:
            workload = (f"{args.arch} evaluate (engine.py:145-225), synthetic 3x{HW}x{HW}, batch {B}/GPU = int(1.5 x 256) "
                        f"(train.py:167), 1000 classes, BatchNorm folded (the eval half of BASELINE north_star's hot path)")
        else:
            workload = (f"{args.arch} train step, synthetic 3x{HW}x{HW}, batch {B}/GPU, AdamW, label smoothing 0.1, 1000 classes "
                        f"(BASELINE configs[{cfg_i}])")
        out = {"metric": f"images/sec (whole node) {label} bf16 {HW}^2 " + ("evaluation" if is_eval else "training"),
               "value": round(value, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": workload, "global_batch": B * world, "parallelism": f"dp{world}"},
               "n_ranks_seen": ranks_seen, "foreign_codeobj_lint": guard, "roofline": roofline, "roofline_classes": classes,
               "roofline_step": step,
               "host_enqueue_ms": round(1e3 * enqueue_s, 3),
               "host_enqueue_frac": round(1e3 * enqueue_s / ms_step, 3),
               "host_enqueue_note": f"Python + ctypes + HIP launch time of one step, burst of {nburst} steps from an idle GPU (no launch "
                                    f"blocks); over the {args.steps} timed steps the loop reached its wait after "
                                    f"{round(1e3 * enqueue_timed_s / args.steps, 2)} ms/step, which includes blocking on the full HIP queue",
               "cpu_baseline": cpu,
               "kernels": {k: {"ms_per_step": c["ms_per_step"], "calls_per_step": c["calls_per_step"]} for k, c in classes.items()},
               "kernel_source_hash": kernel_source_hash(),
               "kernel_timing": {"method": "HIP events on the launch stream around every C-ABI call; separate single-stream pass "
                                           "(the timed region overlaps weight gradients on a second stream)",
                                 "steps": psteps, "ms_per_step_with_events": round(1e3 * dt_prof / psteps, 3)},
               ("eval_stats" if is_eval else "train_stats"): {k: round(v, 5) for k, v in stats.items()
                                                              if not k.startswith(("precision_", "recall_"))}}
        print(json.dumps(out))
    if world > 1:
        model.shutdown()            # destroys the library's own communicator when one was made
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
