"""-m gpu: the sampler / training-loss callers on the HIP path vs (a) golden vectors captured from the
real reference glue and (b) the oracle run on identical inputs.

Tolerances: step-0 U-Net input differs from the reference only through the bf16 MutualEncoder GEMMs
scaled by eta=0.1 -> atol 2e-3.  Later quantities pass through the bf16 U-Net and guidance scales up
to 12, so they are compared in relative L2: raw U-Net output <= 3e-2 at EVERY step under teacher forcing (the oracle's
trajectory feeds both U-Nets), final latents of the free-running sampler <= 8e-2 after 6-10 steps AND after 50
(observed values are printed: ~1.6e-2 and ~0.01-0.06); the product sampler's combined guided epsilon is checked at step 0,
where the trajectories have not yet diverged (the guidance mix u + 4(a-cm) + 5(cm-c) + 12(c-u) amplifies the per-branch
error by its coefficients: bound 0.10, see test_sampler_vs_reference_golden)."""
import glob
import os

import numpy as np
import pytest
import torch

import difashion_amd as da
from oracle import glue_ref, sched_ref, unet_ref
from tests.gpu_util import DEV, rel_err
from tests.helpers import GLUE_CFG, GOLDEN, enc_params, glue_unet_params, load
from tests.test_gpu_unet import hip_unet

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN, "sample_*.npz")))
TRAIN = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "train_*.npz")))


@pytest.fixture(scope="module")
def unet():
    return hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32)


def encoder(rec):
    p = enc_params(rec)
    hid, flat = p["mlp.0.weight"].shape
    enc = da.MutualEncoder(cate_num=11, cate_emb_size=8, latent_channels=4, latent_size=GLUE_CFG.sample_size, hid_dim=hid)
    enc.load_state_dict({**p, "category_embedding.weight": enc.category_embedding.weight.data}, strict=True)
    return enc.to(DEV).eval()


def test_mutual_encoder_full_size_matches_reference():
    rec = load("mutual_encoder_full.npz")
    torch.manual_seed(3)
    ref_enc = da.MutualEncoder(cate_num=50, cate_emb_size=64, latent_channels=4, latent_size=64, hid_dim=256)
    torch.nn.init.xavier_normal_(ref_enc.category_embedding.weight.data)
    for i in (0, 3):
        torch.nn.init.xavier_normal_(ref_enc.mlp[i].weight.data)
        torch.nn.init.constant_(ref_enc.mlp[i].bias.data, 0)
    with torch.no_grad():
        ref_enc.mlp[0].bias.normal_(0, 0.05)
        ref_enc.mlp[3].bias.normal_(0, 0.05)
    from tests.helpers import checksum
    sd = {k: v for k, v in ref_enc.state_dict().items() if k.startswith("mlp.")}
    if not np.allclose(checksum(sd), rec["weight_checksum"].numpy(), rtol=1e-9):
        pytest.skip("torch RNG stream differs from the capture container")
    y = ref_enc.to(DEV).eval()(rec["x"].to(DEV))
    # bf16 operands, K = 16384 then 256, tanh output in [-1, 1]
    assert float((y.cpu() - rec["y"]).abs().max()) < 2e-2 and rel_err(y.cpu(), rec["y"]) < 1e-2


@pytest.mark.parametrize("case", CASES)
def test_sampler_vs_reference_golden(case, unet):
    rec = load(f"sample_{case}.npz")
    kind = str(rec["sched"])
    sched = da.DDIMScheduler() if kind == "ddim" else da.PNDMScheduler()
    sc, sh, sm = (float(v) for v in rec["scales"])
    taps = {}
    d = lambda k: rec[k].to(DEV)
    final = da.sample_outfits(unet, encoder(rec), sched, olists=rec["olists"], all_latents=d("all_latents"),
                              init_latents=d("init_latents"), hist_latents=d("hist_sel"), null_latent=d("null_latent"),
                              category_prompts=d("category_prompts"), null_prompt=d("null_prompt"),
                              num_inference_steps=int(rec["steps"]), cate_scale=sc, hist_scale=sh, mutual_scale=sm,
                              eta=0.1, use_history=bool(rec["use_history"]), use_mutual_guidance=bool(rec["use_mutual"]),
                              taps=taps)
    n = int(rec["n_calls"])
    assert len([k for k in taps if k.startswith("x_in_")]) == n
    x0 = taps["x_in_0"].cpu()
    assert x0.shape == rec["x_in_0"].shape
    torch.testing.assert_close(x0, rec["x_in_0"], rtol=0, atol=2e-3)
    assert torch.equal(x0[:, 4:], rec["x_in_0"][:, 4:])              # history channels: exact selection/stacking
    e0, el = rel_err(taps["unet_out_0"].cpu(), rec["unet_out_0"]), rel_err(taps[f"unet_out_{n-1}"].cpu(), rec["unet_out_last"])
    ef = rel_err(final.cpu(), rec["final"])
    print(case, f"unet_out_0 {e0:.2e} unet_out_last {el:.2e} final {ef:.2e}")
    assert e0 <= 3e-2
    assert ef <= 8e-2
    # the COMBINED guided epsilon of the product sampler at step 0 (same latents on both sides: no trajectory divergence yet), against
    # the oracle's combination of the golden per-branch predictions -- guards the guidance mix / scale plumbing inside sample_outfits
    mode, _ = glue_ref.cfg_plan(sc, sh, sm, bool(rec["use_history"]), bool(rec["use_mutual"]))
    eps_ref = glue_ref.cfg_combine(mode, rec["unet_out_0"], sc, sh, sm)
    eps_err = rel_err(taps["eps_0"].cpu(), eps_ref)
    print(case, f"eps_0 {eps_err:.2e}")
    # measured 1.6e-2 .. 7.9e-2 over the 14 cases (the guidance scales 12 / 5 / 4 amplify the per-branch error of <= 3e-2 by the
    # norm ratio of the weighted difference to the combination); 0.25 until round 3
    assert eps_err <= 0.10


@pytest.mark.parametrize("case", ["gor_full_ddim10", "fitb_full_ddim10"])
def test_fp8_sampler_vs_reference_golden(case):
    """BASELINE configs[4] end to end ("fp8 ... path, CFG batch, 4-item GOR sampling"): the product sampler with the U-Net in fp8 mode (every
    transformer linear in e4m3) against the golden run of the REAL reference glue.  Stated tolerances of the fp8 walk: raw U-Net output
    of the first step <= 6e-2, final latents after the 10 guided steps <= 0.12 (measured, round 5: 2.9e-2 / 8.9e-2 GOR, 3.1e-2 / 8.1e-2 FITB;
    the bound was 0.15; bf16 walk: 3e-2 / 8e-2)."""
    from difashion_amd import _lib
    rec = load(f"sample_{case}.npz")
    m = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32)
    m.enable_fp8()
    sc, sh, sm = (float(v) for v in rec["scales"])
    taps = {}
    d = lambda k: rec[k].to(DEV)
    _lib.census_reset()
    final = da.sample_outfits(m, encoder(rec), da.DDIMScheduler(), olists=rec["olists"], all_latents=d("all_latents"),
                              init_latents=d("init_latents"), hist_latents=d("hist_sel"), null_latent=d("null_latent"),
                              category_prompts=d("category_prompts"), null_prompt=d("null_prompt"),
                              num_inference_steps=int(rec["steps"]), cate_scale=sc, hist_scale=sh, mutual_scale=sm,
                              eta=0.1, use_history=bool(rec["use_history"]), use_mutual_guidance=bool(rec["use_mutual"]), taps=taps)
    torch.cuda.synchronize()
    assert _lib.census()["gemm_fp8"] > 0
    e0, ef = rel_err(taps["unet_out_0"].cpu(), rec["unet_out_0"]), rel_err(final.cpu(), rec["final"])
    print(case, f"fp8: unet_out_0 {e0:.2e} final {ef:.2e}")
    assert e0 <= 6e-2 and ef <= 0.12


@pytest.mark.parametrize("case", ["gor_full_ddim10", "mix_full_pndm10", "gor_full_ddim50"])
def test_sampler_teacher_forced_every_step(case, unet):
    """Teacher forcing: the fp32 oracle sampler runs the whole trajectory; at EVERY step the tensors it hands to its
    U-Net (assembled input, timestep, text states) go through the HIP U-Net too and the raw per-branch predictions are
    compared -- relative L2 <= 3e-2 at every step, not only at step 0.  (A free-running comparison of the COMBINED epsilon
    mixes this per-step error with the divergence of two trajectories and with the guidance amplification
    u + 4(a-cm) + 5(cm-c) + 12(c-u); the free-running check below therefore bounds only the latents it integrates to.)"""
    rec = load(f"sample_{case}.npz")
    p = glue_unet_params()
    steps = int(rec["steps"])
    sc, sh, sm = (float(v) for v in rec["scales"])
    osched = sched_ref.PNDMRef() if str(rec["sched"]) == "pndm" else sched_ref.DDIMRef()
    otaps = {}
    ref = glue_ref.sample_outfits(lambda x, t, e: unet_ref.unet_forward(p, GLUE_CFG, x, t, e), enc_params(rec), osched,
                                  olists=rec["olists"], all_latents=rec["all_latents"], init_latents=rec["init_latents"],
                                  hist_latents=rec["hist_sel"], null_latent=rec["null_latent"],
                                  category_prompts=rec["category_prompts"], null_prompt=rec["null_prompt"],
                                  num_inference_steps=steps, cate_scale=sc, hist_scale=sh, mutual_scale=sm, taps=otaps)
    n = len([k for k in otaps if k.startswith("x_in_")])
    ehs = otaps["ehs"].to(DEV)
    errs = []
    with torch.no_grad():
        for i in range(n):
            got = unet(otaps[f"x_in_{i}"].to(DEV), otaps[f"t_{i}"], ehs, return_dict=False)[0]
            errs.append(rel_err(got.cpu(), otaps[f"unet_out_{i}"]))
    print(case, "teacher-forced U-Net rel err per step", [f"{e:.2e}" for e in errs])
    assert max(errs) <= 3e-2
    # free-running product sampler on the same inputs: the latents it integrates to
    d = lambda k: rec[k].to(DEV)
    sched = da.PNDMScheduler() if str(rec["sched"]) == "pndm" else da.DDIMScheduler()
    got = da.sample_outfits(unet, encoder(rec), sched, olists=rec["olists"], all_latents=d("all_latents"),
                            init_latents=d("init_latents"), hist_latents=d("hist_sel"), null_latent=d("null_latent"),
                            category_prompts=d("category_prompts"), null_prompt=d("null_prompt"),
                            num_inference_steps=steps, cate_scale=sc, hist_scale=sh, mutual_scale=sm)
    lat_err = rel_err(got.cpu(), ref)
    print(case, "free-running final latents", f"{lat_err:.2e}")
    assert lat_err <= 8e-2


@pytest.mark.parametrize("case", TRAIN)
def test_training_loss_vs_reference_golden(case, unet):
    rec = load(f"train_{case}.npz")
    sched = da.DDIMScheduler(prediction_type=str(rec["pred_type"]))
    gamma = float(rec["snr_gamma"])
    taps = {}
    d = lambda k: rec[k].to(DEV)
    mask = rec["dropout_mask"].to(DEV) if "dropout_mask" in rec else None
    loss = da.train_forward(unet, encoder(rec), sched, latents=d("latents"), noise=d("noise"),
                            timesteps_outfit=rec["timesteps_outfit"], null_latent=d("null_latent"),
                            hist_latents=d("hist_sel"), ehs=d("ehs"), null_prompt=d("null_prompt"),
                            random_p=rec["random_p"], random_p_cate=rec["random_p_cate"],
                            snr_gamma=None if np.isnan(gamma) else gamma, use_history=bool(rec["use_history"]),
                            use_mutual_guidance=bool(rec["use_mutual"]), dropout_mask=mask, taps=taps)
    assert torch.equal(taps["timesteps"].cpu(), rec["timesteps"])
    torch.testing.assert_close(taps["x_in"].cpu(), rec["x_in"], rtol=0, atol=2e-3)
    assert torch.equal(taps["x_in"].cpu()[:, 4:], rec["x_in"][:, 4:])
    assert torch.equal(taps["ehs"].cpu()[:, 0, :4], rec["ehs_rows"])
    e = rel_err(taps["pred"].cpu(), rec["unet_out"])
    print(case, f"pred {e:.2e} loss {float(loss):.6f} ref {float(rec['loss']):.6f}")
    assert e <= 3e-2
    assert abs(float(loss) - float(rec["loss"])) <= 2e-2 * abs(float(rec["loss"]))
