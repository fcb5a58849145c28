"""ViT on the HIP kernels vs the CPU oracle (oracle/vit_ref.py, same bf16 rounding points).  Whole-network tolerances
use the oracle's own re-association noise (fp64 vs fp32 accumulation) as the yardstick, as in tests/test_model_gpu.py;
transformers have no ReLU masks, so the noise is an order of magnitude smaller than for ResNets."""
import copy
import ctypes

import pytest
import torch

from oracle import ops_ref as R
from oracle.vit_ref import ViTRef

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(arch, C, img, seed=0):
    from imageclassification_amd.vit import VisionTransformer
    torch.manual_seed(seed)
    ref = ViTRef(arch, C, img_size=img, bf16_points=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # non-trivial biases / LayerNorm affine / cls token so every gradient path is exercised
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
        ref.cls_token.copy_(0.02 * torch.randn(ref.cls_token.shape, generator=g))
    net = VisionTransformer(arch, C, img_size=img)
    net.load_state_dict(ref.state_dict())
    return ref, net


def test_token_plumbing_kernels():
    from imageclassification_amd import hip
    lib = hip.load()
    B, T, C = 3, 5, 16
    g = torch.Generator().manual_seed(3)
    patches = R.bf16_round(torch.randn(B, T - 1, C, generator=g))
    cls, pos = torch.randn(C, generator=g), torch.randn(T, C, generator=g)
    pd, cd, posd = patches.to(torch.bfloat16).to(DEV), cls.to(DEV), pos.to(DEV)
    tok = torch.empty(B, T, C, dtype=torch.bfloat16, device=DEV)
    assert lib.icamd_vit_tokens_fwd(hip.ptr(pd), hip.ptr(cd), hip.ptr(posd), hip.ptr(tok), B, T, C, hip.stream_ptr()) == 0
    ref = torch.cat([cls.expand(B, 1, C), patches], 1) + pos
    torch.cuda.synchronize()
    assert R.max_bf16_ulp(tok.float().cpu(), R.bf16_round(ref)) <= 1.0
    out = torch.ones(T * C, device=DEV)
    assert lib.icamd_batch_sum(hip.ptr(tok), T * C, B, T * C, hip.ptr(out), 1, hip.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), 1 + tok.float().cpu().sum(0).flatten(), rtol=1e-5, atol=1e-5)
    dst = torch.zeros(B, C, dtype=torch.bfloat16, device=DEV)
    assert lib.icamd_strided_rows_copy(hip.ptr(tok), T * C, hip.ptr(dst), C, B, C, hip.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), tok[:, 0].cpu())


@pytest.mark.parametrize("arch,img,B", [("vit_tiny_test", 64, 6), ("vit_tiny_test", 224, 2),
                                        # ViT-B/16 at its true width and depth (D 768, 12 heads, MLP 3072, 12 blocks, T 197):
                                        # every gradient tensor against the oracle's, not only the loss
                                        ("vit_base_patch16_224", 224, 2)])
def test_vit_forward_backward_matches_oracle(arch, img, B):
    from imageclassification_amd import hip
    C = 10
    ref, net = _pair(arch, C, img)
    ref64 = copy.deepcopy(ref).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, img, img, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1)
    loss.backward()
    out64 = ref64(x.double())
    torch.nn.functional.cross_entropy(out64, y, label_smoothing=0.1).backward()
    net.train()
    ws = net.pack(x.cuda())
    logits = net.forward_packed(ws)
    yd = y.cuda()
    hip.check(net.lib.icamd_softmax_xent(ws["logits"].data_ptr(), net.ncls_p, B, C, yd.data_ptr(), None, 1.0, 0.1, 1.0 / B,
                                         ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(), ws["dlogits"].data_ptr(),
                                         hip.stream_ptr()), "xent")
    net.backward_packed(ws)
    torch.cuda.synchronize()
    got = logits[:, :C].float().cpu()
    noise = R.rel_l2(out64.detach().float(), out.detach())
    err = R.rel_l2(got, out.detach())
    assert err <= 2.0 * max(noise, 2e-3), (err, noise)
    assert abs(float(ws["loss_rows"].mean()) - float(loss)) <= 5e-3 * float(loss)
    p64 = dict(ref64.named_parameters())
    worst = ("", 0.0)
    for name, p in ref.named_parameters():
        e = R.rel_l2(net.grad_of(name), p.grad)
        n = R.rel_l2(p64[name].grad.float(), p.grad)
        if e > worst[1]:
            worst = (name, e)
        assert e <= 3.0 * max(n, 1e-2), (name, e, n)
    print(f"{arch} img{img}: logits err {err:.2e} (self-noise {noise:.2e}); worst grad err {worst[1]:.2e} at {worst[0]}")


def test_vit_base_parameter_layout_and_engine_step():
    """Full ViT-B/16 shapes through the engine for one step (loss vs the fp32 oracle on the same weights)."""
    from imageclassification_amd.engine import evaluate, train_one_epoch
    from imageclassification_amd.mixup import LabelSmoothingCrossEntropy
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    from imageclassification_amd.vit import VisionTransformer
    C, B = 1000, 4
    torch.manual_seed(0)
    ref = ViTRef("vit_base_patch16_224", C, bf16_points=True)
    assert sum(p.numel() for p in ref.parameters()) == 86567656
    net = VisionTransformer("vit_base_patch16_224", C)
    net.load_state_dict(ref.state_dict())
    sd = net.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in ref.state_dict().items())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    with torch.no_grad():
        rl = float(torch.nn.functional.cross_entropy(ref(x), y, label_smoothing=0.1))
    opt = create_optimizer("adamw", 1e-3, 5e-2, net)
    stats = train_one_epoch(net, LabelSmoothingCrossEntropy(0.1), [(x, y)], opt, torch.device("cuda"), 0,
                            NativeScalerWithGradNormCount(), None, None, None, start_steps=0, lr_schedule_values=[1e-4],
                            wd_schedule_values=[5e-2], num_training_steps_per_epoch=1, update_freq=1, use_amp=True,
                            num_classes=C)
    assert abs(stats["loss"] - rl) <= 5e-3 * rl, (stats, rl)
    assert opt.step_count == 1 and float(opt.norm_clip[0]) > 0
    ev = evaluate([(x, y)], net, torch.device("cuda"), C)
    assert "acc1" in ev and ev["loss"] > 0
