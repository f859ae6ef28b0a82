"""CPU tier: the CLIP text-encoder oracle (oracle/clip_ref.py) against the fixtures the REAL ``transformers.CLIPTextModel`` produced
(tests/golden/make_golden_clip.py; reference call sites DiFashion/models/difashion.py:224,234,340-353) -- this is what makes row 8f-2
the one arithmetic row with a PINNED oracle.  No transformers import here: the fixtures are data."""
import numpy as np
import pytest
import torch

from tests.helpers_clip import CASES, case_inputs, checksum, load_fixture, rel
from oracle import clip_ref

TOL = 2e-5          # relative L2, fp32 vs fp32: the restatement and the real class differ in summation order only


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_the_real_transformers_class(name):
    cfg, params, ids = case_inputs(name)
    fx = load_fixture(name)
    np.testing.assert_array_equal(fx["input_ids"], ids.numpy())
    np.testing.assert_allclose(fx["checksum"], checksum(params, ids), rtol=1e-12)       # the regenerated inputs ARE the fixture's inputs
    last, pooled, hidden = clip_ref.clip_text_forward(params, cfg, ids, output_hidden_states=True)
    assert rel(last, torch.from_numpy(fx["last_hidden_state"])) < TOL
    assert rel(pooled, torch.from_numpy(fx["pooler_output"])) < TOL
    for t in fx["taps"]:
        assert rel(hidden[int(t)], torch.from_numpy(fx[f"hidden_{int(t)}"])) < TOL, int(t)


def test_param_table_is_the_transformers_4_32_layout():
    names = [n for n, _ in clip_ref.param_shapes(clip_ref.SD15_CLIP)]
    assert names[0] == "text_model.embeddings.token_embedding.weight" and names[-1] == "text_model.final_layer_norm.bias"
    assert len(names) == 2 + 16 * 12 + 2
    n_params = sum(int(np.prod(s)) for _, s in clip_ref.param_shapes(clip_ref.SD15_CLIP))
    assert n_params == 123_060_480                      # CLIP ViT-L/14 text tower (the published text_encoder of SD-1.5)
    n_h = sum(int(np.prod(s)) for _, s in clip_ref.param_shapes(clip_ref.SD2_CLIP))
    assert n_h == 340_387_840                           # OpenCLIP ViT-H/14 text tower with 23 layers (SD-2)


def test_the_synthetic_weights_exercise_the_causal_softmax():
    """Known-answer properties that do not need the fixtures: position t sees only tokens <= t (changing a later token leaves earlier
    rows bit-identical), and the attention is not the uniform average a too-small init would give."""
    cfg, params, ids = case_inputs("tiny_quickgelu")
    last, _, _ = clip_ref.clip_text_forward(params, cfg, ids)
    ids2 = ids.clone()
    ids2[:, 40] = (ids2[:, 40] + 7) % cfg.vocab_size
    last2, _, _ = clip_ref.clip_text_forward(params, cfg, ids2)
    assert torch.equal(last[:, :40], last2[:, :40])
    assert not torch.allclose(last[:, 40:], last2[:, 40:], atol=1e-3)
    null = clip_ref.clip_text_forward(params, cfg, ids[:1])[0]
    assert torch.allclose(null, last[:1], atol=1e-5)                               # batch rows are independent
