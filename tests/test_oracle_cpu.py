"""CPU suite (-m "not gpu"): the oracle against the golden vectors produced by the reference's own loop code,
the oracle's arithmetic against torch primitives, the host-side logic, and the C-ABI export table."""
import json
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

from oracle import engine_ref as E
from oracle import ops_ref as R
from oracle.resnet_ref import ResNetRef

GOLD = json.load(open(os.path.join(HERE, "golden", "engine_trace.json")))


def _tiny(C):
    from make_engine_fixture import TinyNet
    return TinyNet(C)


def _data(seed, nbatch, B, C, hw=8):
    from make_engine_fixture import make_data
    return make_data(seed, nbatch, B, C, hw)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_oracle_loop_matches_reference_engine_trace(case):
    """oracle/engine_ref.py reproduces what /root/reference/engine.py returned for the same seeded run."""
    C, B, nbatch, uf = case["C"], case["B"], case["nbatch"], case["update_freq"]
    torch.manual_seed(1234)
    np.random.seed(1234)
    model = _tiny(C)
    assert model.state_dict()["fc.bias"].tolist() == case["init_fc_bias"]
    data = _data(77, nbatch, B, C)
    if case["nan_step"] is not None:
        data[case["nan_step"]][0][0, 0, 0, 0] = float("nan")
    steps = nbatch // uf
    lr = E.cosine_scheduler_ref(1e-2, 1e-5, 2, steps, warmup_epochs=1)
    wd = E.cosine_scheduler_ref(5e-2, 5e-3, 2, steps)
    assert np.allclose(lr, case["lr"], rtol=0, atol=0) and np.allclose(wd, case["wd"], rtol=0, atol=0)
    opt = torch.optim.AdamW([{"params": list(model.parameters()), "weight_decay": 5e-2}], lr=1e-2, weight_decay=0.0)
    mix = E.MixupRef(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C) if case["mixup"] else None
    crit = E.SoftTargetCrossEntropyRef() if case["mixup"] else E.LabelSmoothingCrossEntropyRef(0.1)
    ema = E.ModelEmaRef(model, decay=0.9) if case["use_ema"] else None
    loader = [(x.clone(), y.clone()) for x, y in data]
    stats = E.train_one_epoch_ref(model, crit, loader, opt, 0, None, ema, mix, 0, lr, wd, steps, uf, C, cpu_alias=True)
    assert set(stats) == set(case["train_stats"])
    for k, v in case["train_stats"].items():
        assert abs(stats[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, stats[k], v)
    assert np.allclose(model.state_dict()["fc.bias"].tolist(), case["final_fc_bias"], rtol=1e-6, atol=1e-7)
    assert np.allclose(model.state_dict()["bn.running_mean"].tolist(), case["final_bn_running_mean"], rtol=1e-6, atol=1e-7, equal_nan=True)  # a NaN batch poisons BN stats in the reference too
    if ema:
        assert np.allclose(ema.module.state_dict()["fc.bias"].tolist(), case["ema_fc_bias"], rtol=1e-6, atol=1e-7)
    ev = E.evaluate_ref(_data(78, 2, B + B // 2, C), model, C)
    assert list(ev) == list(case["eval_stats"])   # same keys in the same order
    for k, v in case["eval_stats"].items():
        if v != v:   # the poisoned running statistics make the eval loss NaN in the reference as well
            assert ev[k] != ev[k], k
        else:
            assert abs(ev[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, ev[k], v)


def test_cosine_scheduler_matches_reference_values():
    from imageclassification_amd.utils import cosine_scheduler
    for (base, final, ep, n, wu), vals in zip(GOLD["cosine_scheduler"]["args"], GOLD["cosine_scheduler"]["values"]):
        for fn in (cosine_scheduler, E.cosine_scheduler_ref):
            got = fn(base, final, ep, n, warmup_epochs=wu)
            assert len(got) == len(vals) and np.array_equal(np.asarray(got), np.asarray(vals))
    assert cosine_scheduler(1e-3, 1e-6, 3, 7, warmup_epochs=1)[0] == 0.0   # first step lr is 0
    with pytest.raises(AssertionError):
        cosine_scheduler(1e-3, 1e-6, 2, 5, warmup_epochs=0, warmup_steps=3)   # SURVEY Appx C.9


def test_resnet_oracle_parameter_counts_and_names():
    for arch, n in (("resnet18", 11689512), ("resnet50", 25557032)):
        m = ResNetRef(arch)
        assert sum(p.numel() for p in m.parameters()) == n
    sd = ResNetRef("resnet50", 2).state_dict()
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 23512130
    for k in ("conv1.weight", "layer1.0.downsample.0.weight", "layer4.2.bn3.running_var", "fc.bias"):
        assert k in sd
    # zero-init of each block's last BN weight
    assert float(ResNetRef("resnet18").layer1[0].bn2.weight.abs().sum()) == 0.0


def test_vit_oracle_parameter_count():
    from oracle.vit_ref import ViTRef
    m = ViTRef("vit_tiny_test", 10, img_size=64)
    out = m(torch.randn(2, 3, 64, 64))
    assert out.shape == (2, 10)
    names = [n for n, _ in m.named_parameters()]
    for k in ("cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.bias",
              "norm.weight", "head.bias"):
        assert k in names


def test_convnext_oracle_parameter_count():
    from oracle.convnext_ref import ConvNeXtRef
    assert sum(p.numel() for p in ConvNeXtRef("convnext_tiny").parameters()) == 28589128


def test_oracle_ops_agree_with_torch_primitives():
    g = torch.Generator().manual_seed(0)
    y = R.bf16_round(torch.randn(4, 5, 5, 16, generator=g))
    gamma, beta = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g)
    mean, invstd, scale, shift, rm, rv = R.bn_train_coeffs(y, gamma, beta, torch.zeros(16), torch.ones(16), 0.1, 1e-5)
    tm, tv = torch.zeros(16), torch.ones(16)
    ref = torch.nn.functional.batch_norm(y.reshape(-1, 16), tm, tv, gamma, beta, True, 0.1, 1e-5)
    assert torch.allclose(rm, tm, atol=1e-6) and torch.allclose(rv, tv, rtol=1e-5)
    out = torch.addcmul(shift, y.reshape(-1, 16), scale)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    logits = R.bf16_round(torch.randn(6, 10, generator=g) * 3)
    t = torch.randint(0, 10, (6,), generator=g)
    loss, pred, _ = R.softmax_xent(logits, t, None, 1.0, 0.1, 1.0)
    assert torch.allclose(loss, torch.nn.functional.cross_entropy(logits, t, label_smoothing=0.1, reduction="none"),
                          rtol=1e-5, atol=1e-6)
    # soft targets == timm-style lam-mix of smoothed one-hots, and the LS criterion == torch's label_smoothing
    ls = E.LabelSmoothingCrossEntropyRef(0.1)(logits, t)
    assert abs(float(ls) - float(loss.mean())) < 1e-6
    st = E.SoftTargetCrossEntropyRef()(logits, R.soft_targets(t, t.flip(0), 0.3, 0.1, 10))
    l2, _, _ = R.softmax_xent(logits, t, t.flip(0), 0.3, 0.1, 1.0)
    assert abs(float(st) - float(l2.mean())) < 1e-6


def test_product_mixup_draws_equal_oracle_mixup():
    from imageclassification_amd.mixup import Mixup
    for kw in ({"mixup_alpha": 0.8, "cutmix_alpha": 0.0}, {"mixup_alpha": 0.0, "cutmix_alpha": 1.0},
               {"mixup_alpha": 0.8, "cutmix_alpha": 1.0, "prob": 0.7}):
        np.random.seed(5)
        a = Mixup(label_smoothing=0.1, num_classes=7, **kw)
        draws = [a.sample((4, 3, 32, 40)) for _ in range(20)]
        np.random.seed(5)
        b = E.MixupRef(label_smoothing=0.1, num_classes=7, **kw)
        for d in draws:
            x = torch.randn(4, 3, 32, 40)
            x0 = x.clone()
            _, soft = b(x, torch.arange(4) % 7)
            mode, lam, box = b.last
            assert d[0] == mode and abs(d[1] - lam) < 1e-12 and tuple(d[2]) == tuple(box)
            assert torch.allclose(R.pack_input(x0, mode, lam, box)[..., :3].permute(0, 3, 1, 2), R.bf16_round(x), atol=1e-2)
            assert torch.allclose(soft, R.soft_targets(torch.arange(4) % 7, (torch.arange(4) % 7).flip(0), lam, 0.1, 7))


def test_reference_style_mixup_object_draws_like_the_product_sampler():
    """engine.train_one_epoch accepts the reference's `mixup_fn` (timm.data.Mixup, train.py:176-185): an object that only
    carries timm's public fields draws exactly what this package's Mixup draws; live `mixup_enabled` flips are honoured;
    unsupported modes / foreign objects raise."""
    from types import SimpleNamespace
    from imageclassification_amd.mixup import Mixup, sample_params
    kw = dict(mixup_alpha=0.8, cutmix_alpha=1.0, cutmix_minmax=None, mix_prob=0.9, switch_prob=0.5, mode="batch",
              correct_lam=True, label_smoothing=0.1, num_classes=7, mixup_enabled=True)
    ref_style = SimpleNamespace(**kw)
    np.random.seed(11)
    a = [sample_params(ref_style, (4, 3, 32, 40)) for _ in range(30)]
    np.random.seed(11)
    own = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=0.9, label_smoothing=0.1, num_classes=7)
    b = [own.sample((4, 3, 32, 40)) for _ in range(30)]
    assert a == b and {m for m, _, _ in a} == {0, 1, 2}
    ref_style.mixup_enabled = False          # timm's mixup_off_epoch
    assert sample_params(ref_style, (4, 3, 32, 40)) == (0, 1.0, (0, 0, 0, 0))
    with pytest.raises(ValueError, match="elem"):
        sample_params(SimpleNamespace(**dict(kw, mode="elem")), (4, 3, 8, 8))
    with pytest.raises(TypeError, match="switch_prob"):
        sample_params(SimpleNamespace(mixup_alpha=1.0, cutmix_alpha=0.0, mix_prob=1.0, label_smoothing=0.1, num_classes=3),
                      (4, 3, 8, 8))


def test_bench_self_launches_its_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE (the way the driver starts N = 1) must start N rank processes itself:
    fresh children of torch.distributed.run, before torch is imported in the parent (reference entry:
    `torchrun --nproc_per_node=N train.py`, README.md:21).  Here the launch command is captured instead of run."""
    import importlib
    import subprocess
    import sys as _sys
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench = importlib.import_module("bench")
    assert not hasattr(bench, "torch"), "bench.py must not import torch at module level (the launcher parent stays GPU-free)"
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the launcher's return code is the parent's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert cmd[-7].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in vars(bench), "the parent imported torch before launching"


def test_meters_match_reference_semantics():
    from imageclassification_amd.utils import MetricLogger, SmoothedValue
    m = SmoothedValue(window_size=4)
    for v, n in ((1.0, 1), (5.0, 3), (2.0, 1), (4.0, 1), (3.0, 2)):
        m.update(v, n)
    assert m.count == 8 and abs(m.total - 28.0) < 1e-12 and abs(m.global_avg - 3.5) < 1e-12
    assert m.median == torch.tensor([5.0, 2.0, 4.0, 3.0]).median().item() and m.max == 5.0 and m.value == 3.0
    ml = MetricLogger(delimiter="  ")
    ml.update(loss=1.5, class_acc=torch.tensor(0.25), skipped=None)
    assert list(ml.meters) == ["loss", "class_acc"] and str(ml).startswith("loss: 1.5000 (1.5000)")
    with pytest.raises(AttributeError):
        ml.nope


def test_c_abi_library_exports_every_declared_symbol():
    """include/icamd.h <-> libicamd.so <-> the ctypes table agree (no compute call: there is no GPU here)."""
    from imageclassification_amd import hip
    header = open(os.path.join(ROOT, "include", "icamd.h")).read()
    declared = set(re.findall(r"\b(icamd_[a-z0-9_]+)\s*\(", header))
    lib = hip.load()
    assert declared == set(hip.EXPORTED_SYMBOLS), declared ^ set(hip.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.icamd_abi_version() == hip.ABI_VERSION
    d = hip.conv_desc(256, 56, 56, 64, 64, 3, 3, 1, 1)
    import ctypes
    assert lib.icamd_conv2d_stats_rows(ctypes.byref(d)) == 256 * 56 * 56 // 128
    assert lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d)) > 0
    assert lib.icamd_bn_workspace_bytes(64) >= 64 * 2 * 64 * 8 + 4


def test_product_fails_loudly_without_gpu():
    from imageclassification_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.IcamdError):
        from imageclassification_amd.nets import ResNet
        ResNet("resnet18", 10)


def test_optimizer_factory_surface():
    from imageclassification_amd.optim_factory import create_optimizer
    with pytest.raises(ValueError):
        create_optimizer("lamb", 1e-3, 0.05, None)


def test_imagefolder_split_and_eval_transform(tmp_path, monkeypatch):
    """Generated 2-class ImageFolder (the reference's cat/dog set is an external download): class order, per-class
    equal validation counts (reference datasets.py:12-53) and the eval transform's arithmetic (:139-144)."""
    from types import SimpleNamespace
    from PIL import Image
    from imageclassification_amd import datasets as D
    rng = np.random.RandomState(0)
    work = tmp_path / "work"
    os.makedirs(work)
    monkeypatch.chdir(work)                          # class_indices.json goes to ./train_cls/output like the reference's
    data = tmp_path / "data"
    for cls, n in (("cat", 11), ("dog", 17)):
        os.makedirs(data / cls)
        for i in range(n):
            Image.fromarray(rng.randint(0, 255, (20, 24, 3), dtype=np.uint8)).save(data / cls / f"{i:03d}.png")
    args = SimpleNamespace(data_path=str(data), train_split_rato=0.8, input_size=16, color_jitter=0.3, reprob=0.25, aa="")
    train, val, C = D.build_dataset(args)
    assert C == 2 and len(train) + len(val) == 28
    vt = [val[i][1] for i in range(len(val))]
    assert vt.count(0) == vt.count(1) == 3          # 11 - int(11 * 0.8) = 3 of EACH class (reference datasets.py:25)
    assert json.load(open(work / "train_cls" / "output" / "class_indices.json")) == {"0": "cat", "1": "dog"}
    # the reference's count rule on its own example sizes: min 25, ratio 0.9 -> 25 - int(22.5) = 3 (not round(2.5) = 2)
    fake = SimpleNamespace(targets=[0] * 25 + [1] * 40)
    tr_i, va_i = D.split_dataset(fake, 0.9)
    assert len(va_i) == 6 and len(tr_i) == 59 and sorted(tr_i + va_i) == list(range(65))
    # train transform: non-square images are centre-cropped to min(W, H) before the bicubic resize (timm's
    # RandomResizedCrop fallback for scale=(1,1), ratio=(1,1)): an image whose left/right margins are white and whose
    # central square is black must come out all black
    a = np.full((20, 40, 3), 255, dtype=np.uint8)
    a[:, 10:30] = 0
    tt = D.TrainTransform(16, color_jitter=0.0, reprob=0.0)
    out = tt(Image.fromarray(a))
    black = (0.0 - np.array(D.IMAGENET_DEFAULT_MEAN, dtype=np.float32)) / np.array(D.IMAGENET_DEFAULT_STD, dtype=np.float32)
    assert np.allclose(out.numpy(), black[:, None, None] * np.ones((3, 16, 16), dtype=np.float32), atol=1e-6)
    x, y = val[0]
    assert x.shape == (3, 16, 16) and x.dtype == torch.float32
    path, _ = val.base.samples[val.indices[0]]
    img = Image.open(path).convert("RGB").resize((16, 16), Image.BILINEAR)
    ref = (np.asarray(img, dtype=np.float32) / 255.0 - np.array(D.IMAGENET_DEFAULT_MEAN, dtype=np.float32)) / \
        np.array(D.IMAGENET_DEFAULT_STD, dtype=np.float32)
    assert np.allclose(x.numpy(), ref.transpose(2, 0, 1), atol=1e-6)
    xt, _ = train[0]
    assert xt.shape == (3, 16, 16) and torch.isfinite(xt).all()
    with pytest.raises(NotImplementedError):
        D.TrainTransform(16, auto_augment="rand-m9-mstd0.5-inc1")


def test_ra_sampler_matches_reference_index_streams():
    """RASampler against index lists produced by the reference's own class (tests/golden/make_sampler_fixture.py)."""
    from imageclassification_amd.utils import RASampler
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ra_sampler.json")))
    assert len(fx["cases"]) >= 10
    for c in fx["cases"]:
        s = RASampler(list(range(c["n"])), num_replicas=c["replicas"], rank=c["rank"], shuffle=c["shuffle"])
        s.set_epoch(c["epoch"])
        assert len(s) == c["len"]
        assert list(iter(s)) == c["indices"], (c["n"], c["replicas"], c["rank"], c["epoch"])


def test_create_optimizer_names_follow_the_reference_factory():
    """Name parsing of reference optim_factory.py:50-122 for the fused kinds; everything else is refused loudly."""
    from imageclassification_amd import optim_factory as OF

    class FakeModel:
        n_params = 64
        param_arena = torch.zeros(64)

        def parameters(self):
            return [self.param_arena]

    m = FakeModel()
    for name, kind in [("adamw", OF.OPT_ADAMW), ("AdamW", OF.OPT_ADAMW), ("sgd", OF.OPT_SGD_NESTEROV),
                       ("nesterov", OF.OPT_SGD_NESTEROV), ("momentum", OF.OPT_SGD_MOMENTUM), ("adam", OF.OPT_ADAM),
                       ("lion", OF.OPT_LION)]:
        opt = OF.create_optimizer(name, 1e-3, 0.05, m)
        assert getattr(opt, "kind", OF.OPT_ADAMW) == kind
        g = opt.param_groups[0]
        assert g["weight_decay"] == 0.05 and g["name"] == "decay"
        if kind in (OF.OPT_SGD_MOMENTUM, OF.OPT_SGD_NESTEROV):
            assert g["momentum"] == 0.9 and g["nesterov"] == (kind == OF.OPT_SGD_NESTEROV)
            assert opt.exp_avg_sq is None
    for bad in ("radam", "lookahead_adamw", "fusedlamb", "rmsprop"):
        with pytest.raises(ValueError):
            OF.create_optimizer(bad, 1e-3, 0.05, m)


def test_train_cli_surface_matches_the_reference_parser():
    """Every flag of the reference's train.py exists here with the same default (flag table extracted from the reference
    source by tests/golden/make_cli_fixture.py); the deliberate differences are listed."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("icamd_train_cli", os.path.join(root, "train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mine = {o: a for a in mod.get_args_parser()._actions for o in a.option_strings if o.startswith("--")}
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_cli_flags.json")))["flags"]
    assert len(ref) >= 50
    assert [k for k in ref if k not in mine] == []
    deliberate = {"--model": "resnet50",      # reference default efficientvit_m0 is outside the built model families
                  "--pretrained": False}      # no network: weights are never downloaded
    for k, spec_ in ref.items():
        if "default" not in spec_ or spec_["default"] == "<expr>":
            continue
        want = deliberate.get(k, spec_["default"])
        assert mine[k].default == want, (k, mine[k].default, want)
    assert sorted(k for k in mine if k not in ref and k != "--help") == ["--gpu_aug", "--num_classes", "--synthetic"]


# ---- ConvNeXt oracle pinned to the reference's own definition (tests/golden/make_convnext_fixture.py) ----------------
from _convnext_pin import _bf16_bits_to_f32, convnext_ref_to_timm_name  # noqa: E402


def test_convnext_oracle_matches_reference_vectors():
    """oracle/convnext_ref.py (fp32, no rounding points) against vectors computed by the reference's own Block / LayerNorm /
    ConvNeXt classes: stage outputs, input gradient and every parameter gradient of the 4-stage backbone; single blocks at
    dims 96 and 192 with the reference's 1e-6 layer scale; LayerNorm in both data formats."""
    import numpy as np
    from oracle.convnext_ref import ConvNeXtRef, _Block, _ln2d
    from oracle import ops_ref as R
    v = np.load(os.path.join(ROOT, "tests", "golden", "convnext_ref_vectors.npz"))
    # backbone
    net = ConvNeXtRef("convnext_pin", 10, bf16_points=False)
    sd = net.state_dict()
    names = [k[len("net/param/"):] for k in v.files if k.startswith("net/param/")]
    assert len(names) == 4 + 3 * 4 + 4 * 9      # stem, three downsample layers, four blocks
    for n in names:
        t = convnext_ref_to_timm_name(n)
        val = _bf16_bits_to_f32(v["net/param/" + n]).reshape(sd[t].shape)
        sd[t].copy_(val)
    x = _bf16_bits_to_f32(v["net/x"]).reshape(2, 3, 64, 64).requires_grad_(True)
    feats = net.forward_features(x)
    for i in range(4):
        assert R.rel_l2(feats[i].detach(), torch.from_numpy(v[f"net/feat{i}"])) <= 2e-6, i
    (feats[3] * torch.from_numpy(v["net/r"])).sum().backward()
    assert R.rel_l2(x.grad, torch.from_numpy(v["net/dx"])) <= 1e-5
    params = dict(net.named_parameters())
    for n in names:
        g = params[convnext_ref_to_timm_name(n)].grad
        assert R.rel_l2(g, torch.from_numpy(v["net/grad/" + n]).reshape(g.shape)) <= 1e-5, n
    # single blocks, layer scale 1e-6
    for dim in (96, 192):
        k = f"blk{dim}"
        blk = _Block(dim, False)
        m = {"dwconv": blk.conv_dw, "norm": blk.norm, "pwconv1": blk.mlp.fc1, "pwconv2": blk.mlp.fc2}
        with torch.no_grad():
            blk.gamma.copy_(_bf16_bits_to_f32(v[f"{k}/param/gamma"]))
            for rn, mod in m.items():
                mod.weight.copy_(_bf16_bits_to_f32(v[f"{k}/param/{rn}.weight"]).reshape(mod.weight.shape))
                mod.bias.copy_(_bf16_bits_to_f32(v[f"{k}/param/{rn}.bias"]))
        xb = _bf16_bits_to_f32(v[f"{k}/x"]).reshape(2, dim, 5, 7).requires_grad_(True)
        yb = blk(xb)
        assert R.rel_l2(yb.detach(), torch.from_numpy(v[f"{k}/y"])) <= 1e-6
        (yb * torch.from_numpy(v[f"{k}/r"])).sum().backward()
        assert R.rel_l2(xb.grad, torch.from_numpy(v[f"{k}/dx"])) <= 1e-6
        assert R.rel_l2(blk.gamma.grad, torch.from_numpy(v[f"{k}/grad/gamma"])) <= 1e-5
        for rn, mod in m.items():
            gw = mod.weight.grad[:4] if rn.startswith("pwconv") else mod.weight.grad
            assert R.rel_l2(gw, torch.from_numpy(v[f"{k}/grad/{rn}.weight"]).reshape(gw.shape)) <= 1e-5, (dim, rn)
            assert R.rel_l2(mod.bias.grad, torch.from_numpy(v[f"{k}/grad/{rn}.bias"])) <= 1e-5, (dim, rn)
    # LayerNorm: the oracle's one form (_ln2d / nn.LayerNorm over the channel axis) against BOTH reference data formats
    C = 96
    ln = torch.nn.LayerNorm(C, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(_bf16_bits_to_f32(v["ln/w"]))
        ln.bias.copy_(_bf16_bits_to_f32(v["ln/b"]))
    r = torch.from_numpy(v["ln/r"])
    for fmt in ("channels_last", "channels_first"):
        ln.zero_grad()
        xl = _bf16_bits_to_f32(v["ln/x"]).reshape(2, 5, 7, C).requires_grad_(True)
        y = ln(xl) if fmt == "channels_last" else _ln2d(xl.permute(0, 3, 1, 2), ln).permute(0, 2, 3, 1)
        (y * r).sum().backward()
        assert R.rel_l2(y.detach(), torch.from_numpy(v[f"ln/{fmt}/y"])) <= 1e-6
        assert R.rel_l2(xl.grad, torch.from_numpy(v[f"ln/{fmt}/dx"])) <= 1e-5
        assert R.rel_l2(ln.weight.grad, torch.from_numpy(v[f"ln/{fmt}/dw"])) <= 1e-5
        assert R.rel_l2(ln.bias.grad, torch.from_numpy(v[f"ln/{fmt}/db"])) <= 1e-5


def test_checkpoint_model_object_serves_the_reference_consumers_without_a_gpu():
    """checkpoint["model"] after torch.load (imageclassification_amd.checkpoint.DeferredModel): state_dict / load_state_dict /
    eval / to / re-pickle / deepcopy work on the CPU copy -- what modelchange.py:155-162 needs with map_location="cpu" --
    and using it as a model without a GPU fails loudly instead of computing anything on the CPU."""
    import copy
    import io
    from imageclassification_amd import hip
    from imageclassification_amd.checkpoint import DeferredModel
    ref = ResNetRef("resnet18", 2)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    m = DeferredModel("imageclassification_amd.nets", "ResNet", {"arch": "resnet18", "num_classes": 2}, sd, True)
    buf = io.BytesIO()
    torch.save({"model": m, "epoch": 3}, buf)
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu", weights_only=False)
    got = ck["model"]
    assert isinstance(got, DeferredModel) and got.arch == "resnet18" and got.num_classes == 2 and got.training
    assert all(torch.equal(got.state_dict()[k], sd[k]) for k in sd)
    ema = {k: (v + 1 if v.is_floating_point() else v) for k, v in sd.items()}
    got.load_state_dict(ema)
    assert torch.equal(got.state_dict()["fc.weight"], sd["fc.weight"] + 1)
    with pytest.raises(KeyError):
        got.load_state_dict({"fc.weight": sd["fc.weight"]})
    assert got.eval() is got and not got.training and got.to("cpu") is got
    assert torch.equal(copy.deepcopy(got).state_dict()["conv1.weight"], got.state_dict()["conv1.weight"])
    if not torch.cuda.is_available():
        with pytest.raises(hip.IcamdError):
            got(torch.zeros(1, 3, 32, 32))


def test_image_oracle_is_pillow_bit_for_bit():
    """oracle/image_ref.py (the checker of the GPU input pipeline) against Pillow itself, which the reference's
    transforms delegate to (datasets.py:121-144 through timm / torchvision on PIL images): two-pass 8-bit resampling (bicubic
    and bilinear, up- and down-scaling, odd sizes), centre crop, and the three ImageEnhance operations of ColorJitter."""
    from PIL import Image, ImageEnhance
    from oracle import image_ref as I
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in [(37, 53, 16, 16), (120, 90, 64, 64), (20, 20, 64, 64), (17, 61, 32, 48), (64, 64, 64, 64), (100, 80, 7, 9)]:
        a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(a)
        for filt, pf in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
            assert np.array_equal(np.asarray(im.resize((ow, oh), pf)), I.resize_u8(a, oh, ow, filt)), (h, w, oh, ow, filt)
    a = rng.randint(0, 256, (23, 31, 3), dtype=np.uint8)
    im = Image.fromarray(a)
    for f in (0.7, 0.85, 1.0, 1.13, 1.3):
        assert np.array_equal(np.asarray(ImageEnhance.Brightness(im).enhance(f)), I.enhance_brightness(a, f))
        assert np.array_equal(np.asarray(ImageEnhance.Contrast(im).enhance(f)), I.enhance_contrast(a, f))
        assert np.array_equal(np.asarray(ImageEnhance.Color(im).enhance(f)), I.enhance_color(a, f))
    chained = ImageEnhance.Brightness(ImageEnhance.Color(ImageEnhance.Contrast(im).enhance(1.21)).enhance(0.77)).enhance(1.08)
    assert np.array_equal(np.asarray(chained), I.color_jitter(a, (1, 2, 0), (1.08, 1.21, 0.77)))
    b = rng.randint(0, 256, (20, 37, 3), dtype=np.uint8)
    assert np.array_equal(I.center_square(b), np.asarray(Image.fromarray(b).crop((8, 0, 28, 20))))
    assert np.array_equal(I.center_square(b.transpose(1, 0, 2)), np.asarray(Image.fromarray(b.transpose(1, 0, 2)).crop((0, 8, 20, 28))))


def test_gelu_constants_of_the_hip_kernels_hold_their_stated_error():
    """csrc/common.h evaluates GELU as max(z, 0) - u * 2^P(u), u = min(|z|, 9.5), P = a degree-8 polynomial (the constants are
    read from the header, so the test follows them).  The same arithmetic in numpy float32 over EVERY bf16 value in
    [-9.5, 9.5] against the erf definition in fp64: the error bounds the header states (1.2e-6 absolute, 2.1e-5 relative) --
    two orders below the bf16 rounding of the result -- and the derivative form Phi(z) + z phi(z)."""
    from scipy.special import ndtr
    src = open(os.path.join(ROOT, "imageclassification_amd", "csrc", "common.h")).read()
    body = src[src.index("gelu_tail2(const f32x2 z)"):src.index("return {u, f32x2{")]
    coef = [float(m) for m in re.findall(r"splat2\((-?[0-9.e+-]+)f\)", body)]
    assert len(coef) == 9, coef                                   # highest degree first
    bits = np.arange(0, 1 << 16, dtype=np.uint32)
    z = (bits << 16).view(np.float32)
    z = z[np.isfinite(z) & (np.abs(z) <= 9.5)]
    u = np.minimum(np.abs(z), np.float32(9.5)).astype(np.float32)
    p = np.full_like(u, np.float32(coef[0]))
    for c in coef[1:]:
        p = (p * u + np.float32(c)).astype(np.float32)            # (the kernel fuses the multiply-add: one rounding fewer)
    q = np.exp2(p.astype(np.float64))                             # Phi(-u)
    gelu = np.maximum(z, 0).astype(np.float64) - u.astype(np.float64) * q
    ref = z.astype(np.float64) * ndtr(z.astype(np.float64))
    err = np.abs(gelu - ref)
    assert err.max() <= 1.5e-6, err.max()
    nz = np.abs(ref) > 0
    assert (err[nz] / np.abs(ref[nz])).max() <= 3e-5
    cdf = np.where(z < 0, q, 1.0 - q)
    phi = np.exp2(z.astype(np.float64) ** 2 * -0.72134752044448170 - 1.3257480647361593)
    grad_ref = ndtr(z.astype(np.float64)) + z.astype(np.float64) * np.exp(-0.5 * z.astype(np.float64) ** 2) / math.sqrt(2 * math.pi)
    assert np.abs(cdf + z * phi - grad_ref).max() <= 3e-6


def test_bf16_close_rejects_non_finite_values():
    """the elementwise comparator of the kernel tests must fail on a NaN / Inf in the tested tensor (a comparison
    `|a-b| > bound` is False for NaN and let it through), with and without an allowed outlier share."""
    from oracle import ops_ref as R
    b = R.bf16_round(torch.randn(4096))
    assert R.bf16_close(b.clone(), b) and R.bf16_close(b.clone(), b, max_frac=1e-3)
    for poison in (float("nan"), float("inf"), -float("inf")):
        a = b.clone()
        a[17] = poison
        assert not R.bf16_close(a, b)
        assert not R.bf16_close(a, b, max_frac=1e-6)
        assert not R.bf16_close(b, a)            # a poisoned oracle value must not pass either
    a = b.clone()
    a[5] += 1.0                                  # one finite outlier: rejected at max_frac 0, accepted within the share
    assert not R.bf16_close(a, b) and R.bf16_close(a, b, max_frac=1e-3)


def test_isa_lint_flags_the_packed_fp32_op_sel_form(tmp_path):
    """tools/isa_lint.py (run by csrc/build.sh): the instruction form behind round 3's wrong LayerNorm column sums is reported,
    its harmless relatives (no op_sel, op_sel_hi only, v_pk_mov_b32) are not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    s = tmp_path / "k.s"
    s.write_text("_Z1kv:\n\tv_pk_add_f32 v[68:69], v[68:69], v[30:31] op_sel:[0,1] op_sel_hi:[1,0]\n"
                 "\tv_pk_fma_f32 v[6:7], v[20:21], v[18:19], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
                 "\tv_pk_add_f32 v[2:3], v[2:3], v[4:5]\n\tv_pk_mul_f32 v[2:3], v[2:3], v[4:5] op_sel_hi:[1,0]\n"
                 "\tv_pk_mov_b32 v[24:25], v[20:21], v[20:21] op_sel:[1,0]\n")
    hits, n = lint.lint(str(s))
    assert n == 4 and [h[0] for h in hits] == [2, 3] and all(h[1] == "_Z1kv" for h in hits)
    assert lint.main([str(s)]) == 1


def test_loss_curve_fixture_is_the_live_oracles_curve():
    """tests/golden/loss_curve_resnet50.json (the oracle trajectories the GPU loss-curve test compares with) against the live
    oracle: the first two steps re-run here (ResNet-50, batch 32, 128 x 128: ~20 s) must reproduce the stored losses -- step 0
    depends on the forward only (1e-5), step 1 on one full backward + AdamW step with lr = 1e-3 / 24 (1e-4: a different host CPU
    re-associates the convolutions).  A changed oracle, init, seed or recipe shows here, on the CPU, not as a GPU mystery."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_loss_curve_fixture", os.path.join(HERE, "golden", "make_loss_curve_fixture.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    doc = json.load(open(os.path.join(HERE, "golden", "loss_curve_resnet50.json")))
    assert len(doc["oracle"]) == gen.STEPS == 24 and len(doc["oracle_fp64"]) == 24
    assert doc["oracle"][-1] < 0.35 * doc["oracle"][0]
    ref = gen.reference_model()
    loader = gen.batches()[:2]
    lr, wd = gen.schedules()
    opt = torch.optim.AdamW([{"params": list(ref.parameters()), "weight_decay": 5e-4}], lr=1e-3, weight_decay=0.0)
    from oracle import engine_ref as E
    tr = []
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        E.train_one_epoch_ref(ref, E.LabelSmoothingCrossEntropyRef(0.1), [(x.clone(), t.clone()) for x, t in loader], opt,
                              lr_schedule_values=lr, wd_schedule_values=wd, num_training_steps_per_epoch=2, num_classes=gen.C, trace=tr)
    assert abs(tr[0]["loss"] - doc["oracle"][0]) <= 1e-5 * doc["oracle"][0], (tr[0]["loss"], doc["oracle"][0])
    assert abs(tr[1]["loss"] - doc["oracle"][1]) <= 1e-4 * doc["oracle"][1], (tr[1]["loss"], doc["oracle"][1])
    assert abs(doc["oracle_fp64"][0] - doc["oracle"][0]) <= 1e-3 * doc["oracle"][0]


@pytest.mark.parametrize("name,batch", [("r04_bench_n1.json", 256), ("r04_bench_eval.json", 384), ("r04_bench_vit.json", 256),
                                        ("r04_bench_convnext.json", 256)])
def test_committed_bench_lines_keep_the_contract(name, batch):
    """The bench lines committed under profiles/ (what bench.py printed on the MI355X box) carry every field of the driver's
    contract and are self-consistent: value = batch / ms_per_step, roofline.frac = achieved / peak, roofline.kernel is the class
    with the largest time of ALL classes, the blended step is the sum of the classes' algorithmic work over the timed step, and
    the PMC summary the traffic came from is stamped with the same kernel-source hash as the line."""
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_classes", "roofline_step", "host_enqueue_ms", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - batch * d["n_gpus"] * 1e3 / d["ms_per_step"]) <= 2e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 2500.0) and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    cls = d["roofline_classes"]
    assert r["kernel"] == max(cls, key=lambda k: cls[k]["ms_per_step"])
    assert isinstance(r["traffic"], int) and r["traffic"] > 0      # committed lines were printed with the PMC summary in place
    step = d["roofline_step"]
    assert abs(step["algorithmic_GB"] - sum(c["algorithmic_GB_per_step"] for c in cls.values())) <= 0.05
    assert abs(step["hbm_frac"] - step["algorithmic_GB"] / (d["ms_per_step"] * 1e-3) / 8000.0) <= 2e-3
    assert 0.0 < d["host_enqueue_ms"] < d["ms_per_step"]
    if name in ("r04_bench_n1.json", "r04_bench_eval.json"):
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    pmc = {"r04_bench_n1.json": "r04_pmc_traffic.json", "r04_bench_eval.json": "r04_pmc_traffic_eval.json",
           "r04_bench_vit.json": "r04_pmc_traffic_vit.json", "r04_bench_convnext.json": "r04_pmc_traffic_convnext.json"}[name]
    assert json.load(open(os.path.join(ROOT, "profiles", pmc)))["kernel_source_hash"] == d["kernel_source_hash"]


def test_foreign_codeobj_guard_reports_the_mapped_rccl():
    """bench.py's op_sel hazard guard for code objects this repository does not build: the RCCL library the
    process maps is identified by sha256 against the lint record of tools/lint_foreign_codeobj.sh; only a byte-identical, hit-free
    library reads "clean", anything else says it is unverified (never silently fine)."""
    import bench
    from imageclassification_amd import hip
    lib = hip.load()
    g = bench.foreign_codeobj_guard(lib)
    assert set(g) == {"rccl_version", "rccl_library", "verdict"}
    assert g["verdict"].startswith(("clean:", "unverified:", "HAZARD:"))
    if lib.icamd_rccl_available():
        assert g["rccl_version"] > 0 and g["rccl_library"] and "librccl" in g["rccl_library"]
