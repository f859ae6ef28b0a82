"""-m gpu: the HIP CLIP text encoder (csrc/clip.hip behind ``difashion_amd.CLIPTextModel``) against fixtures produced by
the REAL ``transformers.CLIPTextModel`` (tests/golden/make_golden_clip.py) -- the class the reference instantiates at
DiFashion/models/difashion.py:70-72 and calls as ``text_encoder(input_ids)[0]`` at :224, :234, :340-342, :352.  The fixtures are data
(token ids, outputs); weights are regenerated from the case's seed, the fixture's checksum proves they are the same tensors.

Stated tolerance: fp32 against fp32 -- relative L2 <= 2e-5 on last_hidden_state, on pooler_output and on every recorded hidden state
(the HIP encoder is fp32 end to end on v_mfma_f32_16x16x4_f32; only the summation order differs).  Observed values are printed."""
import dataclasses

import numpy as np
import pytest
import torch

import difashion_amd as da
from difashion_amd.data import category_prompt
from difashion_amd.prompts import PromptTable
from oracle import clip_ref
from tests.gpu_util import DEV
from tests.helpers_clip import CASES, case_inputs, checksum, load_fixture, rel
from tests.helpers_data import StubTokenizer, synthetic_dataset

pytestmark = pytest.mark.gpu
TOL = 2e-5


def hip_clip(cfg, params):
    m = da.CLIPTextModel(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act, layer_norm_eps=cfg.layer_norm_eps,
                         eos_token_id=cfg.eos_token_id, bos_token_id=cfg.bos_token_id, pad_token_id=cfg.pad_token_id, init_seed=None)
    m.load_state_dict(params)
    return m.to(DEV).eval().requires_grad_(False)


@pytest.mark.parametrize("name", list(CASES))
def test_hip_clip_matches_the_real_transformers_class(name):
    cfg, params, ids = case_inputs(name)
    fx = load_fixture(name)
    np.testing.assert_array_equal(fx["input_ids"], ids.numpy())
    np.testing.assert_allclose(fx["checksum"], checksum(params, ids), rtol=1e-12)
    m = hip_clip(cfg, params)
    out = m(ids.to(DEV), output_hidden_states=True)
    last = out[0]
    assert last.shape == fx["last_hidden_state"].shape and last.dtype == torch.float32 and last.device.type == "cuda"
    e_last, e_pool = rel(last.cpu(), torch.from_numpy(fx["last_hidden_state"])), rel(out.pooler_output.cpu(), torch.from_numpy(fx["pooler_output"]))
    e_taps = {int(t): rel(out.hidden_states[int(t)].cpu(), torch.from_numpy(fx[f"hidden_{int(t)}"])) for t in fx["taps"]}
    print(name, f"last_hidden_state {e_last:.2e} pooler_output {e_pool:.2e} hidden states",
          " ".join(f"{t}:{e:.1e}" for t, e in e_taps.items()))
    assert e_last <= TOL and e_pool <= TOL and max(e_taps.values()) <= TOL
    # the reference's call forms: text_encoder(ids)[0], tuple output, rows independent of the batch they ride in (no cross-row op)
    again = m(ids.to(DEV))
    assert again.hidden_states is None and torch.equal(again[0], last)                  # no atomics: reruns are bit-identical
    assert torch.equal(m(ids.to(DEV), return_dict=False)[0], last)
    assert torch.equal(m(ids[1:2].to(DEV))[0], last[1:2])
    # causal mask: a later token cannot change earlier rows
    ids2 = ids.clone()
    ids2[:, ids.shape[1] // 2] = (ids2[:, ids.shape[1] // 2] + 5) % cfg.vocab_size
    assert torch.equal(m(ids2.to(DEV))[0][:, :ids.shape[1] // 2], last[:, :ids.shape[1] // 2])


def test_error_behaviour_follows_the_transformers_class():
    cfg, params, ids = case_inputs("tiny_short_seq")
    m = hip_clip(cfg, params)
    with pytest.raises(ValueError, match="specify input_ids"):
        m()
    with pytest.raises(ValueError, match="max_position_embeddings"):
        m(torch.zeros(1, cfg.max_position_embeddings + 1, dtype=torch.long, device=DEV))
    with pytest.raises(IndexError):
        m(torch.full((1, 5), cfg.vocab_size, dtype=torch.long, device=DEV))
    with pytest.raises(da.DfhError, match="no CPU fallback"):
        hip_clip(cfg, params).cpu()(ids)
    assert m.dtype == torch.float32 and not any(p.requires_grad for p in m.parameters())


def test_prompt_table_built_with_the_hip_encoder():
    """PromptTable.build (the once-per-run encoding of the closed prompt set, data_utils.py:96-111) through the HIP encoder equals
    encoding each batch's prompts on the fly the way difashion.py:218-224 / :340-353 do."""
    cfg = dataclasses.replace(clip_ref.TINY_CLIP, vocab_size=1001)           # StubTokenizer ids run to 1000
    m = hip_clip(cfg, clip_ref.init_params(cfg, 5))
    _, id_cate, _, _ = synthetic_dataset()
    tok = StubTokenizer()
    table = PromptTable.build(m, tok, id_cate, DEV, batch_size=4)
    assert table.table.shape == (len(id_cate) + 1, 16, cfg.hidden_size) and table.table.device.type == "cuda"
    cats = torch.tensor([[1, 2, 3, 5], [4, 6, 1, 3]], device=DEV)
    ids = StubTokenizer()([category_prompt(id_cate[int(c)]) for c in cats.reshape(-1).cpu()], max_length=16, padding="max_length",
                          truncation=True, return_tensors="pt").input_ids
    assert torch.equal(table.lookup(cats), m(ids.to(DEV))[0])
    null_ids = StubTokenizer()([""], max_length=16, padding="max_length", truncation=True, return_tensors="pt").input_ids
    assert torch.equal(table.null_prompt, m(null_ids.to(DEV))[0])


def test_clip_checkpoint_directory_round_trip(tmp_path):
    cfg, params, ids = case_inputs("tiny_gelu_eos")
    m = hip_clip(cfg, params)
    want = m(ids.to(DEV))[0]
    m.save_pretrained(str(tmp_path / "text_encoder"))
    m2 = da.CLIPTextModel.from_pretrained(str(tmp_path), subfolder="text_encoder").to(DEV)
    assert m2.config.hidden_act == "gelu" and m2.config.eos_token_id == cfg.eos_token_id
    assert torch.equal(m2(ids.to(DEV))[0], want)


def test_difashion_runs_end_to_end_on_the_hip_text_encoder():
    """The reference's own wiring (difashion.py:70-72, :340-353): ``DiFashion`` holding the HIP ``CLIPTextModel`` as ``text_encoder`` and
    calling ``text_encoder(input_ids)[0]`` per sampling call -- against the same run fed from a ``PromptTable`` built ONCE with that encoder
    (prompts.py): the encoder is deterministic and its rows independent, so the two runs must agree bit for bit; and against the fp32
    oracle text states within the sampler's tolerance."""
    import types
    from difashion_amd.difashion import DiFashion
    from oracle import clip_ref as cr
    from tests.helpers import GLUE_CFG, glue_unet_params, load
    from tests.test_gpu_difashion import CATE_NUM, H, IdentityVAE, TensorKeyDict, sample_inputs
    from tests.test_gpu_pipeline import encoder
    from tests.test_gpu_unet import hip_unet
    cfg = dataclasses.replace(cr.TINY_CLIP, hidden_size=GLUE_CFG.cross_attention_dim, vocab_size=64)
    params = cr.init_params(cfg, 9)
    text = hip_clip(cfg, params)
    unet = hip_unet(GLUE_CFG, glue_unet_params(), max_batch=32)
    rec = load("sample_fitb_full_ddim10.npz")
    olists = torch.tensor([[3, 0, 5, 6], [7, 8, 9, 0]])
    images, null_img, cats, ids, uids, oids, init, hist = sample_inputs(2, olists, seed=31)

    class Tok:                       # the category id is the prompt (one token + eos), "" is the bare eos row
        model_max_length = 77

        def __call__(self, texts, padding=None, max_length=77, truncation=True, return_tensors="pt"):
            out = torch.zeros(len(texts), max_length, dtype=torch.long)
            out[:, 0] = 1
            return types.SimpleNamespace(input_ids=out)

    ids = ids.clone()
    ids[:, :, 1] = 1
    args = types.SimpleNamespace(use_history=True, use_mutual_guidance=True, eta=0.1)
    d = lambda t: t.to(DEV)
    hist_dev = {u: TensorKeyDict({c: d(v) for c, v in h.items()}) for u, h in hist.items()}
    kw = dict(uids=uids, oids=oids, input_ids=ids, olists=olists, outfit_images=d(images.reshape(8, 4, H, H)), category=cats, history=hist_dev,
              num_inference_steps=4, category_guidance_scale=7.5, hist_guidance_scale=3.0, mutual_guidance_scale=2.0, null_img=d(null_img), eta=0.0,
              init_latents=d(init), output_type="latent", return_dict=True)
    live = DiFashion(args, vae=IdentityVAE(), unet=unet, fashion_encoder=encoder(rec), noise_scheduler=da.DDIMScheduler(), text_encoder=text,
                     tokenizer=Tok())
    got = live.fashion_generation(**kw)[0].images
    # the same prompts through a table built once: row c = text_encoder(ids of category c)[0], last row the empty prompt
    cat_ids = sorted(set(cats.reshape(-1).tolist()))
    rows = torch.zeros(len(cat_ids) + 1, 77, dtype=torch.long)
    rows[:-1, 0] = torch.tensor(cat_ids) + 5
    rows[:, 1] = 1
    rows[-1, 0], rows[-1, 1] = 1, 0
    table = PromptTable(text(rows.to(DEV))[0], cat_ids)
    cached = DiFashion(args, vae=IdentityVAE(), unet=unet, fashion_encoder=encoder(rec), noise_scheduler=da.DDIMScheduler(), prompt_table=table)
    assert torch.equal(cached.fashion_generation(**kw)[0].images, got)
    assert torch.isfinite(got).all() and got.shape == init.shape
    # fp32 oracle text states for the same ids: the HIP encoder's are 1e-6 away, far inside what the U-Net's bf16 path resolves
    want = cr.clip_text_forward(params, cfg, ids[0])[0]
    assert rel(text(ids[0].to(DEV))[0].cpu(), want) < TOL
