"""Structural checks of the U-Net restatement (oracle/unet_ref.py): the only cross-checks on
SURVEY.md Appendix A available without diffusers (parameter census, key table, topology)."""
import torch

from oracle import unet_ref


def test_param_census_sd15_and_sd2():
    assert unet_ref.param_count(unet_ref.SD15) == 859_532_484      # 859.53 M  
    assert unet_ref.param_count(unet_ref.SD2BASE) == 865_922_244   # 865.92 M


def test_key_table_has_diffusers_names():
    s = unet_ref.param_shapes(unet_ref.SD15)
    assert s["conv_in.weight"] == (320, 8, 3, 3)
    assert s["down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert s["down_blocks.2.downsamplers.0.conv.weight"] == (1280, 1280, 3, 3)
    assert s["mid_block.attentions.0.proj_in.weight"] == (1280, 1280, 1, 1)
    assert s["up_blocks.0.resnets.2.conv1.weight"] == (1280, 2560, 3, 3)
    assert s["up_blocks.1.resnets.2.conv_shortcut.weight"] == (1280, 1920, 1, 1)
    assert s["up_blocks.3.resnets.0.conv1.weight"] == (320, 960, 3, 3)
    assert s["up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.weight"] == (2560, 320)
    assert "down_blocks.3.attentions.0.norm.weight" not in s and "up_blocks.0.attentions.0.norm.weight" not in s
    assert "up_blocks.3.upsamplers.0.conv.weight" not in s and "down_blocks.3.downsamplers.0.conv.weight" not in s
    s2 = unet_ref.param_shapes(unet_ref.SD2BASE)
    assert s2["mid_block.attentions.0.proj_in.weight"] == (1280, 1280)


def test_forward_shapes_and_timestep_forms():
    cfg = unet_ref.TINY
    p = unet_ref.init_params(cfg, 0, affine_jitter=0.05)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 16, 16, generator=g)
    e = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    y0 = unet_ref.unet_forward(p, cfg, x, torch.tensor(481), e)          # 0-d tensor (sampling, difashion.py:520)
    y1 = unet_ref.unet_forward(p, cfg, x, 481, e)                        # python int
    y2 = unet_ref.unet_forward(p, cfg, x, torch.tensor([481, 481]), e)   # (B,) int64 (training, :251)
    assert y0.shape == (2, 4, 16, 16)
    torch.testing.assert_close(y0, y1, rtol=0, atol=0)
    torch.testing.assert_close(y0, y2, rtol=0, atol=0)


def test_batch_rows_are_independent():
    cfg = unet_ref.TINY
    p = unet_ref.init_params(cfg, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 8, 16, 16, generator=g)
    e = torch.randn(3, 77, cfg.cross_attention_dim, generator=g)
    t = torch.tensor([5, 300, 900])
    full = unet_ref.unet_forward(p, cfg, x, t, e)
    one = unet_ref.unet_forward(p, cfg, x[1:2], t[1:2], e[1:2])
    torch.testing.assert_close(full[1:2], one, rtol=1e-4, atol=1e-5)


def test_timestep_embedding_is_cos_then_sin():
    emb = unet_ref.timestep_embedding(torch.tensor([0, 7]), 320)
    assert emb.shape == (2, 320)
    torch.testing.assert_close(emb[0, :160], torch.ones(160))
    torch.testing.assert_close(emb[0, 160:], torch.zeros(160))
    torch.testing.assert_close(emb[1, 0], torch.cos(torch.tensor(7.0)))
    torch.testing.assert_close(emb[1, 160], torch.sin(torch.tensor(7.0)))
