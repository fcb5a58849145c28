"""BASELINE.json configs[0] analogue on the GPU path: ResNet-18 on a generated 2-class ImageFolder through the
reference's command line (train.py), including checkpoint keys and auto-resume (reference utils.py:536-615)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_folder(root, n_per_class=48, hw=48):
    from PIL import Image
    rng = np.random.RandomState(1)
    for ci, cls in enumerate(("cat", "dog")):
        os.makedirs(os.path.join(root, cls))
        for i in range(n_per_class):
            a = rng.randint(0, 90, (hw, hw, 3)).astype(np.uint8)
            a[..., ci] += 150                      # class = dominant colour channel: learnable in a few steps
            Image.fromarray(a).save(os.path.join(root, cls, f"{i:03d}.png"))


def test_train_cli_two_class_imagefolder(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import train as T
    data = tmp_path / "data"
    os.makedirs(data)
    _make_folder(str(data))
    work = tmp_path / "work"
    os.makedirs(work / "train_cls" / "output")
    monkeypatch.chdir(work)
    argv = ["--model", "resnet18", "--data_path", str(data), "--batch_size", "16", "--epochs", "3", "--input_size", "48",
            "--num_workers", "0", "--mixup", "0", "--warmup_epochs", "1", "--lr", "2e-3", "--model_ema", "true",
            "--use_amp", "true", "--clip_grad", "5.0", "--reprob", "0"]
    args = T.get_args_parser().parse_args(argv)
    stats = T.main(args)
    assert {"train_loss", "train_class_acc", "test_loss", "test_acc1", "test_avg_precision", "test_recall_1",
            "test_acc1_ema", "epoch", "n_parameters"} <= set(stats)
    assert stats["n_parameters"] == "11.18M"                      # ResNet-18 at 2 classes (SURVEY Appx A.2)
    lines = [json.loads(l) for l in open(work / "train_cls" / "log.txt")]
    assert [l["epoch"] for l in lines] == [0, 1, 2]
    assert lines[-1]["train_loss"] < lines[0]["train_loss"]        # it learns the colour rule
    assert lines[-1]["test_acc1"] >= 75.0
    ck = torch.load(work / "train_cls" / "output" / "checkpoint-2.pth", map_location="cpu", weights_only=False)
    assert {"model", "optimizer", "epoch", "scaler", "input_shape", "num_classes", "args", "model_ema"} <= set(ck)
    sd = ck["model"].state_dict()                                   # the reference's consumers call exactly this
    assert sd["conv1.weight"].shape == (64, 3, 7, 7) and sd["fc.weight"].shape == (2, 512)
    assert os.path.exists(work / "train_cls" / "output" / "checkpoint-best.pth")
    # checkpoint["model"] is usable the way the reference's tools use it:
    #  * modelchange.py:155-162 convert_model_ema_to_model: load on the CPU, model.load_state_dict(model_ema), re-save
    best = work / "train_cls" / "output" / "checkpoint-best.pth"
    conv = work / "train_cls" / "output" / "checkpoint-best-converted.pth"
    c = torch.load(best, map_location="cpu", weights_only=False)
    c["model"].load_state_dict(c["model_ema"])
    c.pop("model_ema", None); c.pop("optimizer", None); c.pop("scaler", None)
    torch.save(c, conv)
    c2 = torch.load(conv, map_location="cpu", weights_only=False)
    ema_sd = torch.load(best, map_location="cpu", weights_only=False)["model_ema"]
    assert all(torch.equal(c2["model"].state_dict()[k], ema_sd[k]) for k in ema_sd if ema_sd[k].is_floating_point())
    #  * val.py:14-28 initialize_model + inference: model = checkpoint["model"]; model.eval(); model(img)
    model = c2["model"]
    model.to(torch.device("cuda")).eval()
    from PIL import Image
    from imageclassification_amd.datasets import build_transform
    img = build_transform(False, args)(Image.open(data / "dog" / "000.png").convert("RGB")).unsqueeze(0)
    with torch.no_grad():
        out = model(img.to("cuda"))
    assert tuple(out.shape) == (1, 2) and int(out.float().argmax(1)) == 1        # class index 1 = "dog"
    import copy
    clone = copy.deepcopy(model)                                                   # timm ModelEmaV3(model) deep-copies it
    assert torch.equal(clone.state_dict()["fc.weight"], model.state_dict()["fc.weight"])
    # the same recipe with the transforms on the GPU: workers only decode, icamd_image_pipeline does the rest
    work2 = tmp_path / "work_gpu_aug"
    os.makedirs(work2 / "train_cls" / "output")
    monkeypatch.chdir(work2)
    args_g = T.get_args_parser().parse_args(argv[:-2] + ["--reprob", "0.25", "--gpu_aug", "true", "--auto_resume", "false"])
    stats_g = T.main(args_g)
    lines_g = [json.loads(l) for l in open(work2 / "train_cls" / "log.txt")]
    assert lines_g[-1]["train_loss"] < lines_g[0]["train_loss"] and stats_g["test_acc1"] >= 75.0
    monkeypatch.chdir(work)
    # auto-resume continues from the latest numbered checkpoint with optimizer state
    args2 = T.get_args_parser().parse_args(argv[:-6] + ["--epochs", "4", "--use_amp", "true", "--reprob", "0"])
    args2.epochs = 4
    stats2 = T.main(args2)
    assert stats2["epoch"] == 3 and args2.start_epoch == 3
