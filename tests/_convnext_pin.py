"""Shared by the CPU and GPU ConvNeXt pin tests: decoding of tests/golden/convnext_ref_vectors.npz."""
import torch


def _bf16_bits_to_f32(a):
    return torch.from_numpy(a.astype("int32") << 16).view(torch.float32)


def convnext_ref_to_timm_name(name):
    """Parameter names of the reference backbone (semantic_segmentation/backbone/convnext.py:79-99) -> timm's
    classification model names used by the product and the oracle."""
    parts = name.split(".")
    if parts[0] == "downsample_layers":
        i, j = int(parts[1]), int(parts[2])
        head = f"stem.{j}" if i == 0 else f"stages.{i}.downsample.{j}"
        return ".".join([head] + parts[3:])
    if parts[0] == "stages":
        sub = {"dwconv": "conv_dw", "norm": "norm", "pwconv1": "mlp.fc1", "pwconv2": "mlp.fc2", "gamma": "gamma"}[parts[3]]
        return ".".join([f"stages.{parts[1]}.blocks.{parts[2]}", sub] + parts[4:])
    raise KeyError(name)
