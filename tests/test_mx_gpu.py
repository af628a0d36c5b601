"""MX fp8 weights (BASELINE config 5: "FasterViT-0 bf16/fp8 weights ... on CDNA4 fp8 MFMA"): csrc/dfd_mx.hip against
oracle/ops_ref.py's restatement of the OCP MX conversion rule.

  * quantisation is BYTE work: element bytes (e4m3fn) and scale bytes (e8m0) must equal the oracle's bit for bit, for
    weights and for activations (bf16 and f32 inputs, zero blocks, huge / tiny magnitudes, saturation);
  * the dequantised [K][N] weight copy (what the bf16 backward multiplies by) is exact;
  * the block-scaled MFMA GEMM: integer-valued operands (exact in e4m3, asymmetric) must give the exact integer result —
    this pins the operand lane layout and the A/B roles — and random operands must match deq(a) @ deq(w)^T to f32
    accumulation accuracy; ragged M and N edges;
  * network level: FasterViT-0 with fp8 weights against the oracle holding THE SAME quantised weights dequantised to f32:
    eval logits and a training step within the bf16 tolerances of tests/test_fastervit_gpu.py.
"""

from __future__ import annotations

import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _K():
    from deepfakedetection_amd import kernels as K

    return K


def _nasty(rows: int, cols: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(rows, cols, generator=g)
    v *= torch.logspace(-6, 6, rows)[:, None]             # every row its own magnitude
    v[0, :32] = 0.0                                        # an all-zero block
    v[1, 32:64] = 1e-30                                    # tiny block
    v[2, 5] = 3.0e38                                       # near f32 max
    v[3, 64:96] = torch.tensor([448.0, 449.0, 464.0, 480.0, 500.0, 511.0, -500.0, 256.0] * 4)   # saturation inside a block
    return v


@pytest.mark.parametrize("shape", [(64, 256), (96, 1024), (1536, 512)])
def test_weight_quantisation_is_bit_exact(shape):
    K = _K()
    w = _nasty(*shape, seed=1)
    mw, kn = K.mx_quant_weight(w.cuda(), BF)
    torch.cuda.synchronize()
    q, s = R.mx_quant(w)
    assert torch.equal(mw.q.cpu(), q), int((mw.q.cpu() != q).sum())
    assert torch.equal(mw.scale.cpu(), s)
    deq = R.mx_dequant(q, s)
    finite = torch.isfinite(deq.to(BF).float())
    assert torch.equal(kn.float().cpu().t()[finite], deq.to(BF).float()[finite])
    assert torch.equal(deq.to(BF).float()[finite], deq[finite]), "a dequantised e4m3 value times 2^e is exact in bf16"


def test_non_finite_values_survive_the_quantiser():
    """fmaxf / fminf drop NaNs, so a NaN or infinite activation used to come out of the quantiser as +-448 * 2^e and
    a diverged run kept producing finite logits.  A block with a non-finite value is now the OCP MX NaN block (scale 0xff, elements
    0x7f), byte for byte as the oracle says, and the scaled MFMA turns it into NaN outputs for exactly the rows that held it."""
    K = _K()
    a = _nasty(64, 256, seed=5).float()
    a[3, 40] = float("nan")
    a[17, 200] = float("inf")
    a[18, 0] = -float("inf")
    q, s = K.mx_quant_rows(a.cuda().view(64, 1, 1, 256))
    torch.cuda.synchronize()
    qr, sr = R.mx_quant(a)
    assert torch.equal(q.cpu(), qr) and torch.equal(s.cpu(), sr)
    assert int(sr[3, 1]) == 0xFF and int(sr[17, 6]) == 0xFF and int(sr[18, 0]) == 0xFF and int((sr == 0xFF).sum()) == 3
    w = _nasty(256, 256, seed=6)
    mw, kn = K.mx_quant_weight(w.cuda(), BF)
    out = K.mx_gemm(q, s, mw, torch.float32).cpu()
    bad_rows = torch.tensor([3, 17, 18])
    assert torch.isnan(out[bad_rows]).all(), "a NaN block did not reach the output"
    good = torch.ones(64, dtype=torch.bool)
    good[bad_rows] = False
    clean = _nasty(64, 256, seed=5).float()
    qc, sc_ = K.mx_quant_rows(clean.cuda().view(64, 1, 1, 256))
    ref = K.mx_gemm(qc, sc_, mw, torch.float32).cpu()
    assert torch.equal(out[good].nan_to_num(), ref[good].nan_to_num()), "the other rows changed"
    # the weight side too, and its dequantised [K][N] copy for the bf16 backward
    w2 = w.clone()
    w2[5, 33] = float("nan")
    mw2, kn2 = K.mx_quant_weight(w2.cuda(), BF)
    q2, s2 = R.mx_quant(w2)
    assert torch.equal(mw2.q.cpu(), q2) and torch.equal(mw2.scale.cpu(), s2)
    assert torch.isnan(kn2.float().cpu()[32:64, 5]).all() and torch.isfinite(kn2.float().cpu()[:32, 5]).all()


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_activation_quantisation_is_bit_exact(dtype):
    K = _K()
    a = _nasty(301, 256, seed=2).to(dtype)
    q, s = K.mx_quant_rows(a.cuda().view(301, 1, 1, 256))
    torch.cuda.synchronize()
    qr, sr = R.mx_quant(a.float())
    assert torch.equal(q.cpu(), qr) and torch.equal(s.cpu(), sr)


def test_activation_quantisation_behind_a_gelu_prologue():
    """fc2's operand: GELU(h + b1) applied while quantising.  The device GELU is an erfc approximation (1.5e-7 absolute),
    so bytes may differ by one rounding step in rare cases: judged on dequantised values."""
    K = _K()
    from deepfakedetection_amd._lib import ACT_GELU

    g = torch.Generator().manual_seed(3)
    h = torch.randn(257, 1024, generator=g).to(BF)
    st = torch.zeros(4, 1024)
    st[0] = 0.5 + torch.rand(1024, generator=g)
    st[1] = torch.randn(1024, generator=g) * 0.1
    q, s = K.mx_quant_rows(h.cuda().view(257, 1, 1, 1024), K.pro_bn_act(st.cuda(), ACT_GELU))
    torch.cuda.synchronize()
    want = torch.nn.functional.gelu(st[0] * h.float() + st[1]).to(BF).float()
    got = R.mx_dequant(q.cpu(), s.cpu())
    ref = R.mx_dequant(*R.mx_quant(want))
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 0.07 * scale       # one e4m3 step (2^-3 relative) of the largest block at most
    assert float((got != ref).float().mean()) < 2e-3            # and only where the two GELUs round differently


@pytest.mark.parametrize("M,K_,N", [(64, 128, 128), (100, 256, 36), (53 * 7, 1024, 256), (4096, 512, 1536)])
def test_scaled_mfma_gemm_exact_on_integer_data(M, K_, N):
    """Small integers and power-of-two block scales are exact in e4m3/e8m0 and their products sum exactly in f32: any
    mistake in the operand lane layout, the scale selection or the A/B roles changes the integer result."""
    K = _K()
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-8, 9, (M, K_), generator=g).float()
    w = torch.randint(-8, 9, (N, K_), generator=g).float()
    # asymmetric structure: row- and block-dependent power-of-two magnitudes
    a *= (2.0 ** torch.randint(-2, 3, (M, K_ // 32), generator=g)).repeat_interleave(32, dim=1)
    w *= (2.0 ** torch.randint(-2, 3, (N, K_ // 32), generator=g)).repeat_interleave(32, dim=1)
    mw, _ = K.mx_quant_weight(w.cuda(), torch.float32)
    aq, asc = K.mx_quant_rows(a.cuda().view(M, 1, 1, K_))
    out = K.mx_gemm(aq, asc, mw, torch.float32)
    torch.cuda.synchronize()
    want = (a.double() @ w.double().t()).float()
    assert torch.equal(R.mx_dequant(aq.cpu(), asc.cpu()), a), "integer test data must survive quantisation exactly"
    assert torch.equal(out.cpu(), want), float((out.cpu() - want).abs().max())


@pytest.mark.parametrize("M,K_,N", [(53 * 16, 256, 768), (49 * 8, 512, 2048), (777, 1024, 256)])
def test_scaled_mfma_gemm_random(M, K_, N):
    K = _K()
    g = torch.Generator().manual_seed(7)
    a = (torch.randn(M, K_, generator=g) * torch.logspace(-2, 2, M)[:, None]).to(BF)
    w = torch.randn(N, K_, generator=g) * 0.05
    mw, _ = K.mx_quant_weight(w.cuda(), BF)
    out, _, _ = K.pwconv(a.cuda().view(M, 1, 1, K_), None, mw)
    torch.cuda.synchronize()
    want = R.mx_linear(a.float(), w)
    got = out.float().cpu().view(M, N)
    row_scale = want.abs().amax(dim=1, keepdim=True).clamp_min(1e-20)
    assert float(((got - want).abs() / row_scale).max()) <= 1.0e-2          # bf16 output rounding (2^-8) + summation order
    # weights-only yardstick (what a bf16-activation x fp8-weight product would give): within bf16's two ulps of the row
    loose = R.mx_linear(a.float(), w, quantise_a=False)
    assert float(((got - loose).abs() / row_scale).max()) <= 6e-2


def _fv_pair(seed=0):
    """(oracle whose Linear weights were replaced by their MX-dequantised values, HIP model with fp8_weights and the
    ORIGINAL masters)."""
    from tests.test_fastervit_gpu import _imports, randomise

    Hip, Ref, _, _ = _imports()
    torch.manual_seed(seed)
    ref = Ref("0", 2, 224, drop_path_rate=0.0)
    randomise(ref, seed + 1)
    hip = Hip("0", 2, 224, drop_path_rate=0.0, fp8_weights=True)
    hip.load_state_dict(ref.state_dict(), strict=True)
    n = 0
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith(("qkv.weight", "proj.weight", "fc1.weight", "fc2.weight")) and p.dim() == 2 and p.shape[1] % 128 == 0:
                p.copy_(R.mx_dequant(*R.mx_quant(p)))
                n += 1
    assert n >= 4 * 11, n                                   # levels 2 and 3 of FasterViT-0: 6 + 5 blocks (+ carrier blocks)
    return ref, hip.cuda()


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)


def test_fastervit_fp8_weights_eval_logits_against_the_dequantised_oracle():
    ref, hip = _fv_pair()
    ref.eval(); hip.eval()
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = ref(x)
    with torch.inference_mode(), torch.autocast("cuda", dtype=BF):
        got = hip(x.cuda())
    hip.fp8_weights = False
    with torch.inference_mode(), torch.autocast("cuda", dtype=BF):
        plain = hip(x.cuda())                                # bf16 weights, ORIGINAL masters: differs by the weight rounding
    import copy

    with torch.no_grad(), torch.autocast("cpu", dtype=BF):
        yard = rel_err(copy.deepcopy(ref)(x).float(), want)     # the oracle's own bf16 autocast against its f32 self
    err, err_plain = rel_err(got, want), rel_err(plain, want)
    print(f"fp8-weight logits vs dequantised-weight f32 oracle: {err:.4f} (oracle's own bf16 autocast: {yard:.4f}); "
          f"bf16-weight engine with the unquantised masters vs the same oracle: {err_plain:.4f}")
    # W8A8: the activations of 44 Linear layers are rounded to e4m3 (3 mantissa bits) on top of the bf16 pipeline's own
    # rounding, so the bound is the bf16 yardstick plus the measured fp8 activation noise — and it must beat the engine
    # that ignores the weight quantisation altogether
    assert err <= max(3.0 * yard, 6e-2), (err, yard)
    assert err < err_plain
    top2 = want.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * err * float(want.abs().max())
    assert torch.equal(got.float().cpu().argmax(1)[clear], want.argmax(1)[clear])


def test_fastervit_fp8_weights_training_step():
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    ref, hip = _fv_pair(seed=2)
    ref.train(); hip.train()
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 2, (8,), generator=g)
    want = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(want, y, label_smoothing=0.1)
    loss_ref.backward()
    opt = HipAdamW(hip.parameters(), lr=1e-4)
    with torch.autocast("cuda", dtype=BF):
        got = hip(x.cuda())
        loss = HipCrossEntropyLoss(0.1)(got, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = rel_err(got, want)
    print(f"fp8-weight training-mode logits vs the dequantised-weight f32 oracle: {err:.4f} (largest logit {float(want.detach().abs().max()):.3f})")
    assert err <= 0.15, err                                   # e4m3 activations: ~5-9 % of the largest (tiny, random-init) logit
    assert abs(float(loss) - float(loss_ref)) <= 2e-2 * max(1.0, abs(float(loss_ref)))
    rp = dict(ref.named_parameters())
    ga = torch.cat([p.grad.float().cpu().flatten() for _, p in hip.named_parameters()])
    gb = torch.cat([rp[n].grad.flatten() for n, _ in hip.named_parameters()])
    assert torch.isfinite(ga).all()
    ga, gb = ga.double(), gb.double()                         # 31 M entries: f32 dot / norm accumulations drift above 1
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    print(f"fp8-weight training step: gradient cosine vs the dequantised-weight f32 oracle {cos:.4f}")
    assert cos >= 0.97, cos
    opt.step()                                               # masters stay f32; the next forward re-quantises them
    torch.cuda.synchronize()
