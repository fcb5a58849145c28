"""Parity at BASELINE.json's FULL sizes (batch 256, 3x224x224, 1000 classes), where the CPU oracle would take minutes per
step: size-independent properties of the step instead of element-wise comparison.

* reproducibility: the same step from the same state twice gives bit-identical loss rows and gradients (the build's
  claim of fixed-order reductions, incl. the weight-gradient stream running beside the main stream);
* softmax-cross-entropy: every row of dlogits sums to zero (sum_c (p_c - t_c) = 0 for smoothed / mixed targets);
* eval forward: a sample's logits do not depend on which batch it sits in (ResNet-50 with BatchNorm folded, batch 256 vs
  its first 32 images) -- different tile decompositions and kernel routings, same numbers to bf16 rounding;
* ViT-B/16 (no batch statistics): the gradient of the 256-image batch is the mean of the gradients of its two halves
  (1/B is a power of two, so the only difference is fp32 summation order) -- the identity data parallelism rests on;
* BatchNorm in train mode: the batch statistics written by the step are those of the stored conv output (fp64 check on
  the stem: 256 x 112 x 112 x 64 values).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
B, HW, C = 256, 224, 1000


def _step(net, x, y, smoothing=0.1):
    from imageclassification_amd import hip
    lib = hip.load()
    ws = net.pack(x)
    logits = net.forward_packed(ws)
    hip.check(lib.icamd_softmax_xent(logits.data_ptr(), net.ncls_p, x.shape[0], C, y.data_ptr(), None, 1.0, smoothing,
                                     1.0 / x.shape[0], ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(),
                                     ws["dlogits"].data_ptr(), hip.stream_ptr()), "xent")
    net.backward_packed(ws)
    torch.cuda.synchronize()
    return ws


def _data(n=B, seed=88):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, HW, HW, generator=g).to(DEV), torch.randint(0, C, (n,), generator=g).to(DEV)


def test_resnet50_full_batch_step_is_reproducible_and_consistent():
    from imageclassification_amd.nets import ResNet
    net = ResNet("resnet50", C, seed=3, zero_init_last=False)
    x, y = _data()
    buf0 = net.buffer_arena.clone()
    ws = _step(net, x, y)
    loss1, grad1, dl1 = ws["loss_rows"].clone(), net.grad_arena.clone(), ws["dlogits"].clone()
    assert torch.isfinite(loss1).all() and torch.isfinite(grad1).all()
    # (1) every row of dlogits sums to zero (over the real classes; the 24 padding columns are zero)
    rows = dl1.float()[:, :C].double().sum(1)
    assert float(rows.abs().max()) <= 2e-3 * float(dl1.float().abs().max()) * 8     # bf16 rounding of 1000 terms
    assert float(dl1.float()[:, C:].abs().max()) == 0.0
    # (2) BatchNorm statistics of the stem equal those of the stored conv output (fp64)
    y0 = ws["y0"].double().reshape(-1, 64)
    st = net.stat_arena[net.stem_bn.stat_offset: net.stem_bn.stat_offset + 4 * 64].double()
    mean, var = y0.mean(0), y0.var(0, unbiased=False)
    assert torch.allclose(st[:64], mean.to(DEV), rtol=1e-5, atol=1e-6)
    assert torch.allclose(st[64:128], 1.0 / torch.sqrt(var + 1e-5).to(DEV), rtol=1e-5)
    # (3) same state, same batch again: bit-identical
    net.buffer_arena.copy_(buf0)
    ws = _step(net, x, y)
    assert torch.equal(ws["loss_rows"], loss1) and torch.equal(ws["dlogits"], dl1)
    assert torch.equal(net.grad_arena, grad1)
    # (4) eval forward: sample i's logits are the same in a batch of 256 and in a batch of 32
    net.eval()
    full = net(x).float().clone()
    part = net(x[:32]).float()
    err = (full[:32] - part).abs().max().item()
    assert err <= 2.0 ** -7 * full.abs().max().item(), err      # 1 bf16 ulp of the largest logit
    # and folded vs unfolded eval paths agree at the full size
    net.fold_eval = False
    plain = net(x).float()
    rel = ((plain - full).norm() / full.norm()).item()
    assert rel <= 2e-2, rel


def test_vit_base_full_batch_gradient_is_the_mean_of_its_halves():
    from imageclassification_amd.vit import VisionTransformer
    net = VisionTransformer("vit_base_patch16_224", C, seed=4)
    x, y = _data(seed=89)
    _step(net, x, y)
    g_full = net.grad_arena.double().clone()
    _step(net, x[:128].contiguous(), y[:128].contiguous())
    g_a = net.grad_arena.double().clone()
    _step(net, x[128:].contiguous(), y[128:].contiguous())
    g_b = net.grad_arena.double().clone()
    mean = 0.5 * (g_a + g_b)
    rel = ((g_full - mean).norm() / mean.norm()).item()
    assert rel <= 1e-5, rel
    # twice the same step: bit-identical
    _step(net, x, y)
    assert torch.equal(net.grad_arena.double(), g_full)
