"""CPU checks of the EfficientFormerV2 oracle (oracle/efformer_ref.py) and of the host side of its HIP module.

timm (the package that carries the reference's arithmetic, requirements.txt:95) is absent, so the oracle is a
restatement; these are the known answers that pin it: the published parameter counts of all four variants
(timm model cards: 3.60 / 6.19 / 12.71 / 26.32 M with the distillation head), the S1 MAC count of the paper
(0.65 G), the state-dict key grammar the reference's freeze masks rely on, and an independent derivation of the
attention-bias index tables.
"""

from __future__ import annotations

import pytest
import torch

from oracle.efformer_ref import EfficientFormerV2Ref, bias_index, count_macs

PUBLISHED_PARAMS = {"s0": 3_600_256, "s1": 6_185_560, "s2": 12_710_112, "l": 26_322_288}      # 3.60 / 6.19 / 12.71 / 26.32 M


@pytest.mark.parametrize("variant", sorted(PUBLISHED_PARAMS))
def test_parameter_counts_match_the_published_model_cards(variant):
    m = EfficientFormerV2Ref(variant, 1000)
    n = sum(p.numel() for p in m.parameters())
    assert n == PUBLISHED_PARAMS[variant]
    assert round(n / 1e6, 2) == {"s0": 3.60, "s1": 6.19, "s2": 12.71, "l": 26.32}[variant]


def test_s1_macs_match_the_paper():
    macs = count_macs(EfficientFormerV2Ref("s1", 1000)) / 1e9
    assert 0.64 <= macs <= 0.67, macs          # Li et al. 2023, Table 1: 0.65 GMACs


def test_key_grammar_and_freeze_masks():
    m = EfficientFormerV2Ref("s1", 2)
    names = [n for n, _ in m.named_parameters()]
    keys = set(m.state_dict())
    for k in ("stem.conv1.conv.weight", "stem.conv1.conv.bias", "stem.conv2.bn.running_var", "stages.1.downsample.conv.conv.weight",
              "stages.3.downsample.attn.q.local.weight", "stages.3.downsample.attn.q.proj.bn.weight",
              "stages.3.downsample.attn.attention_biases", "stages.2.blocks.7.token_mixer.stride_conv.conv.weight",
              "stages.2.blocks.8.token_mixer.talking_head1.weight", "stages.3.blocks.5.token_mixer.v_local.bn.bias",
              "stages.0.blocks.0.mlp.mid.conv.weight", "stages.0.blocks.0.ls2.gamma", "stages.3.blocks.4.ls1.gamma",
              "norm.weight", "head.weight", "head_dist.bias"):
        assert k in keys, k
    assert not any("attention_bias_idxs" in k for k in keys)            # non-persistent buffer, as in timm
    assert "stages.2.blocks.6.token_mixer.q.conv.weight" not in keys    # attention only in the last num_vit = 2 blocks
    assert "stages.3.blocks.3.ls1.gamma" not in keys
    # reference trainers/efficientformer_v2.py:351-352: warm-up trains names with "classifier" or "head"
    warm = [n for n in names if "classifier" in n or "head" in n]
    assert {n.split(".")[-2] for n in warm} == {"head", "head_dist", "talking_head1", "talking_head2"}
    # :66-74, :389-393: fine-tune set; the earliest trainable tensor is in stages.2.blocks.3
    unfreeze = ("stages.3", "blocks.3", "layer4", "bneck", "features.6", "classifier", "head")
    ft = [n for n in names if any(k in n for k in unfreeze)]
    assert len(ft) == 182 and ft[0].startswith("stages.2.blocks.3.")
    assert m.head.in_features == 224 and m.head_dist.out_features == 2


def test_bias_index_tables():
    """timm builds rel_pos = |q_pos - k_pos| per axis, index = dy * W + dx; checked here by brute force."""
    idx = bias_index((7, 7), (7, 7), 1)
    assert idx.shape == (49, 49) and int(idx.max()) == 48 and int(idx[0, 0]) == 0
    for qi in (0, 10, 48):
        for kj in (0, 5, 33, 48):
            qy, qx, ky, kx = qi // 7, qi % 7, kj // 7, kj % 7
            assert int(idx[qi, kj]) == abs(qy - ky) * 7 + abs(qx - kx)
    idx2 = bias_index((7, 7), (14, 14), 2)                            # Attention2dDownsample: queries on the even grid
    assert idx2.shape == (49, 196) and int(idx2.max()) == 13 * 14 + 13
    for qi in (0, 8, 48):
        for kj in (0, 17, 195):
            qy, qx, ky, kx = 2 * (qi // 7), 2 * (qi % 7), kj // 14, kj % 14
            assert int(idx2[qi, kj]) == abs(qy - ky) * 14 + abs(qx - kx)


def test_forward_shapes_and_distillation_average():
    torch.manual_seed(0)
    m = EfficientFormerV2Ref("s0", 5, img_size=96).eval()
    x = torch.randn(2, 3, 96, 96)
    with torch.no_grad():
        feats = m.forward_features(x)
        out = m(x)
        pooled = feats.mean((2, 3))
        want = (m.head(pooled) + m.head_dist(pooled)) / 2
    assert feats.shape == (2, 176, 3, 3) and out.shape == (2, 5)
    assert torch.allclose(out, want, atol=1e-6)


@pytest.mark.parametrize("variant", ["s0", "s1", "s2", "l"])
def test_hip_module_has_the_oracles_state_dict(variant):
    """Host side of the HIP module (no kernel runs): same keys, same shapes, strict load."""
    from deepfakedetection_amd.efficientformer_v2 import HipEfficientFormerV2

    ref, hip = EfficientFormerV2Ref(variant, 3), HipEfficientFormerV2(variant, 3)
    a, b = ref.state_dict(), hip.state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape for k in a)
    hip.load_state_dict(a, strict=True)
    blk = hip.stages[2].blocks[-1].token_mixer
    assert torch.equal(blk.attention_bias_idxs, ref.stages[2].blocks[-1].token_mixer.attention_bias_idxs)
    assert blk._idx32.dtype == torch.int32 and blk._idx32.numel() == 49 * 49
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip(torch.zeros(1, 3, 224, 224))
