"""Host-side logic of the look-ahead pipeline that needs no GPU (the kernels behind it are covered by -m gpu)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))


def test_conv_planes_rule_matches_the_kernel_tiling():
    """slu_wconv_fwd_bf16(out_planes) needs pool 1 and a channel tiling (1, 2, 4, 5 or 8 tiles of 16) that covers
    round_up(c_out, 32) columns — the rule ops.wconv_bf16_planes_ok mirrors for the model's dispatch."""
    from slu_hip import ops
    assert ops.wconv_bf16_planes_ok(60, 1)          # 4 tiles = 64 columns = round_up(60, 32)
    assert ops.wconv_bf16_planes_ok(64, 1) and ops.wconv_bf16_planes_ok(128, 1)
    assert not ops.wconv_bf16_planes_ok(16, 1)      # 1 tile = 16 columns < 32
    assert not ops.wconv_bf16_planes_ok(80, 1)      # 5 tiles = 80 columns < 96
    assert not ops.wconv_bf16_planes_ok(60, 2)      # pooled outputs go through fp32
    assert not ops.wconv_bf16_planes_ok(200, 1)     # more than 8 tiles


def test_row_table_stands_in_for_the_concatenated_batch():
    from slu_hip import ops
    ptrs = torch.zeros(5, dtype=torch.int64)
    t = ops.RowTable(ptrs, 64, 48000)
    assert t.shape == (320, 48000) and t.dim() == 2 and not t.requires_grad
    assert t.float() is t and t.to("cpu") is t and t.device == ptrs.device


def test_lookahead_width_rules(monkeypatch):
    import training
    assert training._lookahead_width(7, 64) == 7                   # explicit SLU_LOOKAHEAD
    monkeypatch.setenv("SLU_CU_SPLIT", "96")
    w = training._lookahead_width(-1, 64)
    assert 2 <= w <= 32
    if not torch.cuda.is_available():
        assert w == 32 and training._lookahead_width(-1, 4096) == 2    # clamps without a device (256 CUs assumed)
