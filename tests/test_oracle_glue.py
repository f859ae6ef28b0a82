"""Pins oracle/glue_ref.py to the golden vectors captured from the real reference glue
(/root/reference/DiFashion/models/difashion.py via tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import glue_ref, sched_ref, unet_ref
from tests.helpers import GLUE_CFG, GOLDEN, checksum, enc_params, glue_unet_params, load

SAMPLE_CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN, "sample_*.npz")))
TRAIN_CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "train_*.npz")))


@pytest.fixture(scope="module")
def unet():
    p = glue_unet_params()
    return p, (lambda x, t, e: unet_ref.unet_forward(p, GLUE_CFG, x, t, e))


def test_fixture_inventory():
    assert len(SAMPLE_CASES) >= 13 and len(TRAIN_CASES) >= 6


def test_seeded_weights_match_capture(unet):
    rec = load("sample_fitb_cate.npz")
    np.testing.assert_allclose(checksum(unet[0]), rec["unet_checksum"].numpy(), rtol=1e-12)


def test_mutual_encoder_full_size():
    """difashion.py:21-46 at the real size 16384 -> 256 -> 16384."""
    rec = load("mutual_encoder_full.npz")
    torch.manual_seed(3)
    # regenerate the seeded weights exactly as make_golden.cap_mutual_encoder did
    emb = torch.nn.Embedding(50, 64)
    l1 = torch.nn.Linear(16384, 256)
    l2 = torch.nn.Linear(256, 16384)
    for m in (emb,):
        torch.nn.init.xavier_normal_(m.weight.data)
    for m in (l1, l2):
        torch.nn.init.xavier_normal_(m.weight.data)
        torch.nn.init.constant_(m.bias.data, 0)
    with torch.no_grad():
        l1.bias.normal_(0, 0.05)
        l2.bias.normal_(0, 0.05)
    p = {"mlp.0.weight": l1.weight.data, "mlp.0.bias": l1.bias.data,
         "mlp.3.weight": l2.weight.data, "mlp.3.bias": l2.bias.data}
    if not np.allclose(checksum(p), rec["weight_checksum"].numpy(), rtol=1e-9):
        pytest.skip("torch RNG stream differs from the capture container; weights not reproducible")
    y = glue_ref.mutual_encoder(p, rec["x"])
    torch.testing.assert_close(y, rec["y"], rtol=0, atol=0)


def test_compute_snr():
    rec = load("snr_table.npz")
    s = sched_ref.DDIMRef()
    torch.testing.assert_close(glue_ref.compute_snr(s.alphas_cumprod, rec["timesteps"]), rec["snr"], rtol=0, atol=0)


def test_plain_dict_history_is_always_null():
    """with the reference's own dict the history branch is never taken."""
    rec = load("plain_dict_history.npz")
    torch.testing.assert_close(rec["hist_channels_branch0"],
                               rec["null_latent"][None].expand(2, -1, -1, -1), rtol=0, atol=0)


@pytest.mark.parametrize("case", SAMPLE_CASES)
def test_sampler_matches_reference_glue(case, unet):
    rec = load(f"sample_{case}.npz")
    sched = sched_ref.DDIMRef() if str(rec["sched"]) == "ddim" else sched_ref.PNDMRef()
    sc, sh, sm = (float(v) for v in rec["scales"])
    calls = []

    def rec_unet(x, t, e):
        out = unet[1](x, t, e)
        calls.append((x, torch.as_tensor(t), e, out))
        return out

    final = glue_ref.sample_outfits(
        rec_unet, enc_params(rec), sched, olists=rec["olists"], all_latents=rec["all_latents"],
        init_latents=rec["init_latents"], hist_latents=rec["hist_sel"], null_latent=rec["null_latent"],
        category_prompts=rec["category_prompts"], null_prompt=rec["null_prompt"],
        num_inference_steps=int(rec["steps"]), cate_scale=sc, hist_scale=sh, mutual_scale=sm, eta=0.1,
        use_history=bool(rec["use_history"]), use_mutual_guidance=bool(rec["use_mutual"]))
    assert len(calls) == int(rec["n_calls"])
    torch.testing.assert_close(torch.stack([c[1] for c in calls]), rec["timesteps"], rtol=0, atol=0)
    for tag, i in (("0", 0), ("1", 1), ("last", len(calls) - 1)):
        # inputs to the U-Net are pure glue arithmetic: bit-exact at step 0, and float-close later
        # (later steps pass through the U-Net whose conv/matmul reductions may reassociate).
        tol = dict(rtol=0, atol=0) if i == 0 else dict(rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(calls[i][0], rec[f"x_in_{tag}"], **tol)
        torch.testing.assert_close(calls[i][2][:, 0, :4], rec[f"ehs_rows_{tag}"], rtol=0, atol=0)
        torch.testing.assert_close(calls[i][3], rec[f"unet_out_{tag}"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(final, rec["final"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("case", TRAIN_CASES)
def test_training_forward_matches_reference_glue(case, unet):
    rec = load(f"train_{case}.npz")
    sched = sched_ref.DDIMRef(prediction_type=str(rec["pred_type"]))
    gamma = float(rec["snr_gamma"])
    taps = {}
    loss = glue_ref.train_forward(
        unet[1], enc_params(rec), sched, latents=rec["latents"], noise=rec["noise"],
        timesteps_outfit=rec["timesteps_outfit"], null_latent=rec["null_latent"], hist_latents=rec["hist_sel"],
        ehs=rec["ehs"], null_prompt=rec["null_prompt"], random_p=rec["random_p"],
        random_p_cate=rec["random_p_cate"], snr_gamma=None if np.isnan(gamma) else gamma,
        use_history=bool(rec["use_history"]), use_mutual_guidance=bool(rec["use_mutual"]),
        dropout_mask=rec.get("dropout_mask"), taps=taps)
    torch.testing.assert_close(taps["timesteps"], rec["timesteps"], rtol=0, atol=0)
    torch.testing.assert_close(taps["x_in"], rec["x_in"], rtol=0, atol=0)
    torch.testing.assert_close(taps["ehs"][:, 0, :4], rec["ehs_rows"], rtol=0, atol=0)
    torch.testing.assert_close(taps["pred"], rec["unet_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss, rec["loss"], rtol=1e-5, atol=1e-7)
