"""CPU (-m "not gpu"): host-side logic of the product (index tables, guidance planning, scheduler
scalars, multi-GPU sharding) against the oracle.  No kernel is launched."""
import itertools
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import difashion_amd as da
from difashion_amd import dist as ddist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from difashion_amd import pipeline
from oracle import glue_ref, sched_ref


def test_guidance_plan_and_replica_flags_match_oracle():
    for sc, sh, sm, uh, um in itertools.product((1.0, 7.5), (1.0, 4.0), (1.0, 5.0), (True, False), (True, False)):
        name, rep = glue_ref.cfg_plan(sc, sh, sm, uh, um)
        assert pipeline.guidance_plan(sc, sh, sm, uh, um) == name
        mode, br = pipeline._BRANCHES[name]
        assert len(br) == rep and [tuple(b) for b in br] == [tuple(b) for b in glue_ref._BRANCHES[name]]


@pytest.mark.parametrize("olists", [[[0, 0, 0, 0]], [[3, 0, 5, 6], [7, 8, 9, 0]], [[0, 0, 5, 6], [7, 0, 0, 0], [1, 2, 3, 4]]])
def test_sampling_tables_reproduce_reference_sibling_sum(olists):
    """Emulate dfh_mutual_reduce's contract in torch and compare with difashion.py:475-489 (oracle)."""
    ol = torch.tensor(olists)
    g = torch.Generator().manual_seed(1)
    given = torch.randn(ol.numel(), 4, 8, 8, generator=g)
    gen = torch.randn(int((ol == 0).sum()), 4, 8, 8, generator=g)
    if gen.shape[0] == 0:
        pytest.skip("nothing to generate")
    tab, wt = pipeline.sampling_tables(ol)
    out = []
    for j in range(tab.shape[0]):
        acc = torch.zeros_like(gen[0])
        for k in range(tab.shape[1]):
            v = int(tab[j, k])
            acc = acc + wt[j, k] * (gen[v] if v >= 0 else given[-(v + 1)])
        out.append(acc)
    assert torch.equal(torch.stack(out), glue_ref.mutual_sum(ol, given, gen))


def test_training_tables_reproduce_reference_sibling_mean():
    g = torch.Generator().manual_seed(2)
    noisy = torch.randn(8, 4, 8, 8, generator=g)
    tab, wt = pipeline.training_tables(8, 4)
    out = []
    for j in range(8):
        acc = torch.zeros_like(noisy[0])
        for k in range(4):
            acc = acc + wt[j, k] * noisy[int(tab[j, k])]
        out.append(acc)
    assert torch.equal(torch.stack(out), glue_ref.mutual_mean(noisy, 4))


def test_ddim_scalars_reproduce_oracle_step():
    s, r = da.DDIMScheduler(), sched_ref.DDIMRef()
    s.set_timesteps(50)
    r.set_timesteps(50)
    assert s.timesteps.tolist() == r.timesteps.tolist() and s.order == 1 and s.init_noise_sigma == 1.0
    assert torch.equal(s.alphas_cumprod, r.alphas_cumprod)
    g = torch.Generator().manual_seed(3)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    for t in (981, 481, 21, 1):
        k = s.step_coef(t, 0.0)
        x0 = (x - torch.tensor(k.sqrt_b_t) * e) / torch.tensor(k.sqrt_a_t)
        prev = torch.tensor(k.sqrt_a_prev) * x0 + torch.tensor(k.dir_coef) * e
        torch.testing.assert_close(prev, r.step(e, t, x, return_dict=False)[0], rtol=1e-6, atol=1e-6)
    p, pr = da.PNDMScheduler(), sched_ref.PNDMRef()
    p.set_timesteps(50)
    pr.set_timesteps(50)
    assert p.timesteps.tolist() == pr.timesteps.tolist() and len(p.timesteps) == 51


def test_scheduler_config_roundtrip(tmp_path):
    s = da.DDIMScheduler(prediction_type="v_prediction")
    s.save_pretrained(str(tmp_path / "scheduler"))
    s2 = da.DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert s2.config.prediction_type == "v_prediction" and s2.config.num_train_timesteps == 1000


def test_shard_range_partitions_exactly():
    for n, w in itertools.product((0, 1, 7, 8, 9, 64), (1, 2, 3, 8)):
        got = [i for r in range(w) for i in ddist.shard_range(n, r, w)]
        assert got == list(range(n))
        sizes = [len(ddist.shard_range(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = ddist.init("gloo")
    # 5 "outfits" of 2 generated items each; a stand-in denoiser (the product's compute needs the GPU)
    n_outfits = 5
    mine = ddist.shard_range(n_outfits, r, w)
    lat = torch.stack([torch.full((4, 2, 2), float(o * 10 + j)) for o in mine for j in range(2)]) if len(mine) else torch.zeros(0, 4, 2, 2)
    ddist.barrier()
    elapsed = ddist.max_over_ranks(0.5 + r)               # slowest rank defines the step time
    total = ddist.sum_over_ranks(float(len(mine)))
    counts = [2 * len(ddist.shard_range(n_outfits, k, w)) for k in range(w)]
    allv = ddist.gather_outfit_latents(lat, counts)
    q.put((r, elapsed, total, allv[:, 0, 0, 0].tolist()))
    dist.destroy_process_group()


def test_world_size_2_gloo_sharding_and_reductions():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [float(o * 10 + j) for o in range(5) for j in range(2)]
    for r, elapsed, total, vals in res:
        assert elapsed == 1.5 and total == 5.0 and vals == expect


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = ddist.init("gloo")
    # data-parallel training step: every rank computes the gradient of ITS shard of the outfits on an identical replica,
    # the flat gradient buffer is averaged with one all-reduce, then every rank applies the same update
    torch.manual_seed(0)
    weight = torch.randn(6, 3)                                   # replica (same seed on every rank)
    data = torch.arange(8 * 3, dtype=torch.float32).view(8, 3) / 10.0
    mine = ddist.shard_range(8, r, w)
    x = data[mine.start:mine.stop]
    flat_grad = (2.0 * (x @ weight.T)).T @ x / len(mine)          # d/dW of mean_rows ||W x||^2 over the local shard
    flat_grad = flat_grad.reshape(-1).contiguous()
    ddist.all_reduce_gradients(flat_grad)
    weight = weight - 0.1 * flat_grad.view(6, 3)
    logged = ddist.gather_mean(torch.tensor(float(r + 1)))       # the logged loss: mean over the ranks (train.py:695)
    assert float(logged) == 1.5
    ddist.broadcast_parameters(weight)                           # no-op on identical replicas, exercises the collective
    q.put((r, flat_grad.tolist(), weight.reshape(-1).tolist()))
    dist.destroy_process_group()


def test_world_size_2_gloo_gradient_all_reduce_matches_single_process():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    weight = torch.randn(6, 3)
    data = torch.arange(8 * 3, dtype=torch.float32).view(8, 3) / 10.0
    full = ((2.0 * (data @ weight.T)).T @ data / 8).reshape(-1)   # equal shards: mean of shard means == global mean
    for r, g, wnew in res:
        torch.testing.assert_close(torch.tensor(g), full, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(torch.tensor(wnew), (weight - 0.1 * full.view(6, 3)).reshape(-1), rtol=1e-5, atol=1e-6)
    assert res[0][2] == res[1][2]                                 # replicas stay bit-identical


def test_all_reduce_gradients_is_identity_without_a_process_group():
    g = torch.arange(5, dtype=torch.float32)
    assert ddist.all_reduce_gradients(g) is g and g.tolist() == [0, 1, 2, 3, 4]


# ----------------------------------------------------------------------------- checkpoint directories (train.py:516-554)
def test_checkpoint_directories_round_trip_like_the_reference_hooks(tmp_path):
    """save hook: unet.save_pretrained / fashion_encoder.save_pretrained / ema.save_pretrained; load hook:
    from_pretrained(dir, subfolder=...), register_to_config(**loaded.config), load_state_dict, EMAModel.from_pretrained."""
    import difashion_amd as da
    torch.manual_seed(0)
    unet = da.UNet2DConditionModel(sample_size=16, in_channels=4, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
                                   attention_head_dim=(1, 2, 2, 2), init_seed=3)
    unet.conv_in = torch.nn.Conv2d(8, 32, 3, 1, 1)           # widened like difashion.py:84-93
    unet.register_to_config(in_channels=8)
    enc = da.MutualEncoder(cate_num=11, cate_emb_size=8, latent_channels=4, latent_size=16, hid_dim=32)
    ema = da.EMAModel(unet.parameters(), decay=0.99, model_cls=da.UNet2DConditionModel, model_config=unet.config)
    ema.optimization_step = 17
    with torch.no_grad():
        ema.shadow_params[0].add_(1.0)
    out = str(tmp_path)
    unet.save_pretrained(os.path.join(out, "unet"))
    enc.save_pretrained(os.path.join(out, "fashion_encoder"))
    ema.save_pretrained(os.path.join(out, "unet_ema"))
    for sub in ("unet", "fashion_encoder", "unet_ema"):
        assert sorted(os.listdir(os.path.join(out, sub))) == ["config.json", "diffusion_pytorch_model.safetensors"]

    u2 = da.UNet2DConditionModel.from_pretrained(out, subfolder="unet")
    assert u2.config.in_channels == 8 and tuple(u2.conv_in.weight.shape) == (32, 8, 3, 3)
    assert all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), u2.state_dict().values()))
    fresh = da.UNet2DConditionModel(sample_size=16, in_channels=8, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
                                    attention_head_dim=(1, 2, 2, 2), init_seed=None)
    fresh.register_to_config(**u2.config)
    fresh.load_state_dict(u2.state_dict())

    e2 = da.MutualEncoder.from_pretrained(out, subfolder="fashion_encoder")
    assert dict(e2.config) == dict(enc.config) and e2.config.hid_dim == 32
    assert all(torch.equal(a, b) for a, b in zip(enc.state_dict().values(), e2.state_dict().values()))

    ema2 = da.EMAModel.from_pretrained(os.path.join(out, "unet_ema"), da.UNet2DConditionModel)
    assert ema2.optimization_step == 17 and ema2.decay == 0.99
    assert all(torch.equal(a, b) for a, b in zip(ema.shadow_params, ema2.shadow_params))
    ema.load_state_dict(ema2.state_dict())                    # train.py:532


# ----------------------------------------------------------------------------- batch builder
def test_difashion_is_built_from_a_snapshot_directory_like_the_reference_constructor(tmp_path):
    """difashion.py:52-120: scheduler / text_encoder / vae / unet sub-folders of ``pretrained_model_name_or_path``, conv_in widened to 8 input
    channels (pretrained weights in the first four, zeros behind), fresh xavier-normal MutualEncoder, VAE and text encoder frozen.  Host
    logic only (no kernel runs): the snapshot is written by this package's own ``save_pretrained`` methods in the diffusers / transformers
    directory layouts."""
    import types
    import difashion_amd as da
    root = str(tmp_path)
    unet = da.UNet2DConditionModel(sample_size=16, in_channels=4, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
                                   attention_head_dim=(1, 2, 2, 2), init_seed=3)
    unet.save_pretrained(os.path.join(root, "unet"))
    da.AutoencoderKL(block_out_channels=(32, 64, 64, 64), sample_size=32, init_seed=4).save_pretrained(os.path.join(root, "vae"))
    da.CLIPTextModel(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                     init_seed=5).save_pretrained(os.path.join(root, "text_encoder"))
    da.PNDMScheduler(prediction_type="v_prediction").save_pretrained(os.path.join(root, "scheduler"))
    args = types.SimpleNamespace(pretrained_model_name_or_path=root, revision=None, non_ema_revision=None, category_emb_size=8, hid_dim=32,
                                 enable_xformers_memory_efficient_attention=True, use_history=True, use_mutual_guidance=True, eta=0.1)
    tok = types.SimpleNamespace(model_max_length=77)
    m = da.DiFashion.from_pretrained_pipeline(args, logger=None, cate_num=7, device=None, tokenizer=tok)
    assert isinstance(m.noise_scheduler, da.PNDMScheduler) and m.noise_scheduler.config.prediction_type == "v_prediction"
    assert isinstance(m.text_encoder, da.CLIPTextModel) and not any(p.requires_grad for p in m.text_encoder.parameters())
    assert isinstance(m.vae, da.AutoencoderKL) and not any(p.requires_grad for p in m.vae.parameters()) and m.vae_scale_factor == 8
    w = m.unet.conv_in.weight
    assert tuple(w.shape) == (32, 8, 3, 3) and m.unet.config.in_channels == 8
    assert torch.equal(w[:, :4], unet.conv_in.weight) and float(w[:, 4:].abs().max()) == 0.0
    # the reference copies the WEIGHT only (difashion.py:87-92): the widened conv's bias is nn.Conv2d's own fresh draw, not the pretrained one
    assert not torch.equal(m.unet.conv_in.bias, unet.conv_in.bias) and float(m.unet.conv_in.bias.abs().max()) <= 1.0 / (8 * 9) ** 0.5 + 1e-6
    enc = m.fashion_encoder
    assert enc.category_embedding.weight.shape == (7, 8) and enc.mlp[0].weight.shape == (32, 4 * 16 * 16)
    assert float(enc.mlp[0].bias.abs().max()) == 0.0 and float(enc.mlp[3].bias.abs().max()) == 0.0
    assert any(p.requires_grad for p in m.unet.parameters()) and m.tokenizer is tok
    assert torch.equal(m.text_encoder.state_dict()["text_model.final_layer_norm.weight"], torch.ones(64))


def test_preprocess_dataset_matches_the_reference_function(tmp_path):
    """difashion_amd.data.preprocess_dataset vs golden vectors captured from the REAL DiFashion/data_utils.py
    (tests/golden/make_golden_data.py): prompts, token ids, history latents (bit-exact), FITB masking."""
    import numpy as np
    from difashion_amd import data as dd
    from tests.helpers import GOLDEN
    from tests.helpers_data import StubTokenizer, synthetic_dataset
    rec = np.load(os.path.join(GOLDEN, "data_prep.npz"), allow_pickle=False)
    data, id_cate, history, latents = synthetic_dataset()
    np.save(os.path.join(str(tmp_path), "all_item_latents.npy"), latents.numpy())
    tok = StubTokenizer()
    out, hist = dd.preprocess_dataset(data, str(tmp_path), id_cate, history, None, tok, None, "cpu")
    assert tok.seen == list(rec["prompts"])
    assert "a pair of wide-leg pants" in tok.seen[5] and "A photo of a t-shirt," in tok.seen[0]
    for i in range(int(rec["n_outfits"])):
        assert np.array_equal(out["input_ids"][i].numpy(), rec[f"input_ids_{i}"])
        assert np.array_equal(out["category"][i].numpy(), rec[f"category_{i}"])
        assert out["outfits"][i].dtype == torch.int64 and np.array_equal(out["outfits"][i].numpy(), rec[f"outfits_{i}"])
    assert np.array_equal(hist["null"].numpy(), rec["hist_null"])
    for uid in history:
        for cate in history[uid]:
            assert np.array_equal(hist[uid][cate].numpy(), rec[f"hist_{uid}_{cate}"])       # bit-exact fp32 means
    fitb = dd.FashionFITBData(out, {"outfits": [o.tolist() for o in out["outfits"]]}, fill_num=2)[1]
    assert np.array_equal(fitb["outfits"].numpy(), rec["fitb_outfits_1"])
    item = dd.FashionDiffusionData(out)[2]
    assert set(item) == {"uids", "oids", "outfits", "input_ids", "category"} and item["oids"] == 102


def test_encode_item_latents_batches_through_the_vae():
    from difashion_amd import data as dd

    class FakeVAE:                                   # the cache-miss branch: 64-image batches, mode() * scaling_factor
        class config:
            scaling_factor = 0.5
        calls = []

        def to(self, dev):
            return self

        def encode(self, x):
            self.calls.append(x.shape[0])
            return type("O", (), {"latent_dist": type("D", (), {"mode": staticmethod(lambda: x[:, :1] * 2)})})

    imgs = [torch.full((3, 4, 4), float(i)) for i in range(130)]
    vae = FakeVAE()
    lat = dd.encode_item_latents(vae, imgs, "cpu")
    assert vae.calls == [64, 64, 2] and lat.shape == (130, 1, 4, 4) and float(lat[7, 0, 0, 0]) == 7.0


def test_prompt_table_equals_per_batch_encoding():
    """encoding the closed prompt set once + a gather == the reference's per-batch text_encoder calls
    (difashion.py:218-224, :340-353) for any deterministic encoder."""
    from difashion_amd.data import category_prompt
    from difashion_amd.prompts import PromptTable
    from tests.helpers_data import StubTokenizer, synthetic_dataset
    _, id_cate, _, _ = synthetic_dataset()
    emb = torch.nn.Embedding(1001, 6)
    encoder = lambda ids: (emb(ids),)                               # stand-in for CLIPTextModel: (last_hidden_state,)
    tok = StubTokenizer()
    table = PromptTable.build(encoder, tok, id_cate, "cpu", batch_size=3)
    assert table.table.shape == (len(id_cate) + 1, tok.model_max_length, 6)
    cats = torch.tensor([[1, 2, 3, 5], [4, 6, 1, 3]])
    got = table.lookup(cats)
    ids = StubTokenizer()([category_prompt(id_cate[int(c)]) for c in cats.reshape(-1)], max_length=16, padding="max_length",
                          truncation=True, return_tensors="pt").input_ids
    assert torch.equal(got, encoder(ids)[0])
    null_ids = StubTokenizer()([""], max_length=16, padding="max_length", truncation=True, return_tensors="pt").input_ids
    assert torch.equal(table.null_prompt, encoder(null_ids)[0])


def test_checkpoints_written_by_the_reference_stack_load(tmp_path):
    """diffusers 0.18.2 (the reference's pin) saves ``diffusion_pytorch_model.bin`` by default, so the directories the
    reference's save hook writes (train.py:518-524) hold .bin files; published SD VAE weights name the mid-block attention
    query / key / value / proj_attn (diffusers renames on load).  Both must load; unsupported U-Net config switches must be
    refused instead of silently ignored."""
    import json

    import difashion_amd as da
    from difashion_amd._lib import DfhError
    kw = dict(sample_size=16, in_channels=8, block_out_channels=(32, 64, 64, 64), cross_attention_dim=64, attention_head_dim=(1, 2, 2, 2))
    unet = da.UNet2DConditionModel(init_seed=5, **kw)
    d = os.path.join(str(tmp_path), "unet")
    unet.save_pretrained(d)
    os.remove(os.path.join(d, "diffusion_pytorch_model.safetensors"))
    with pytest.raises(FileNotFoundError, match="diffusion_pytorch_model.safetensors, diffusion_pytorch_model.bin"):
        da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    torch.save({k: v.clone() for k, v in unet.state_dict().items()}, os.path.join(d, "diffusion_pytorch_model.bin"))
    u2 = da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    assert all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), u2.state_dict().values()))
    # variant infix: diffusion_pytorch_model.fp16.bin wins over the plain file when asked for
    sd16 = {k: (v + 1).clone() for k, v in unet.state_dict().items()}
    torch.save(sd16, os.path.join(d, "diffusion_pytorch_model.fp16.bin"))
    u3 = da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet", variant="fp16")
    assert torch.equal(u3.state_dict()["conv_in.bias"], unet.state_dict()["conv_in.bias"] + 1)
    # a variant that does not exist is an error, never a silent fall back to the plain file (other weights than asked for)
    with pytest.raises(FileNotFoundError, match="no 'bf16' variant"):
        da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet", variant="bf16")
    # EMA directory as the reference's hook writes it (.bin) -> EMAModel.from_pretrained
    ema = da.EMAModel(unet.parameters(), decay=0.99, model_cls=da.UNet2DConditionModel, model_config=unet.config)
    ed = os.path.join(str(tmp_path), "unet_ema")
    ema.save_pretrained(ed)
    from safetensors.torch import load_file
    sd = load_file(os.path.join(ed, "diffusion_pytorch_model.safetensors"))
    os.remove(os.path.join(ed, "diffusion_pytorch_model.safetensors"))
    torch.save(sd, os.path.join(ed, "diffusion_pytorch_model.bin"))
    e2 = da.EMAModel.from_pretrained(ed, da.UNet2DConditionModel)
    assert all(torch.equal(a, b) for a, b in zip(ema.shadow_params, e2.shadow_params))

    # VAE: old attention key names, projections stored as 1x1 convs in some exports
    vae = da.AutoencoderKL(block_out_channels=(32, 64, 64, 64), sample_size=32, init_seed=6)
    vd = os.path.join(str(tmp_path), "vae")
    vae.save_pretrained(vd)
    os.remove(os.path.join(vd, "diffusion_pytorch_model.safetensors"))
    old = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in vae.state_dict().items():
        for new_name, old_name in ren.items():
            if f".attentions.0.{new_name}." in k:
                k = k.replace(f".attentions.0.{new_name}.", f".attentions.0.{old_name}.")
                if k.endswith("weight") and "encoder" in k:
                    v = v[:, :, None, None]
        old[k] = v.clone()
    assert any(".query." in k for k in old) and any(v.dim() == 4 and ".key." in k for k, v in old.items())
    torch.save(old, os.path.join(vd, "diffusion_pytorch_model.bin"))
    v2 = da.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    assert all(torch.equal(a, b) for a, b in zip(vae.state_dict().values(), v2.state_dict().values()))

    # config switches the native walk does not implement are refused
    for bad in (dict(act_fn="gelu"), dict(upcast_attention=True), dict(class_embed_type="timestep"), dict(resnet_time_scale_shift="scale_shift"),
                dict(only_cross_attention=[True, False, False, False]), dict(flip_sin_to_cos=False), dict(num_class_embeds=10)):
        with pytest.raises(DfhError, match="not implemented"):
            da.UNet2DConditionModel(init_seed=None, **kw, **bad)
    # ... and the SD defaults a real config.json carries are accepted
    cfg = json.load(open(os.path.join(d, "config.json")))
    cfg.update(act_fn="silu", upcast_attention=False, only_cross_attention=False, dual_cross_attention=False, class_embed_type=None,
               num_class_embeds=None, resnet_time_scale_shift="default", flip_sin_to_cos=True, freq_shift=0, center_input_sample=False,
               downsample_padding=1, mid_block_scale_factor=1, mid_block_type="UNetMidBlock2DCrossAttn")
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    da.UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")


def _bf16_wire_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddist.init("gloo")
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(1003, generator=g) * (10.0 ** torch.randint(-3, 3, (1003,), generator=g).float())     # ragged length, wide range
    mine = grad.clone()
    out = ddist.all_reduce_gradients(grad, wire="bf16")
    assert out is grad
    q.put((rank, mine.tolist(), grad.tolist()))
    dist.destroy_process_group()


def test_world_size_2_gloo_bf16_wire_gradient_exchange():
    """dist.exchange_bf16: bf16 on the wire, fp32 accumulation in rank order, identical results on every rank."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, g1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    want = ((g0.bfloat16().float() + g1.bfloat16().float()) / 2).bfloat16().float()        # the scheme's exact arithmetic
    assert res[0][2] == res[1][2]                                                          # bit-identical on both ranks
    assert torch.equal(torch.tensor(res[0][2]), want)
    assert bool(((want - (g0 + g1) / 2).abs() <= 2.0 ** -7 * (g0.abs() + g1.abs()) / 2 + 1e-30).all())   # bf16-close to the fp32 average


def _world8_worker(rank, world, port, q):
    """One rank of an 8-rank data-parallel job on CPU (gloo): outfit sharding + the in-backward range exchange of the flat gradient arena
    (unet.py _backward_overlapped hands out final ranges back to front, the last one ragged), fp32 all-reduce per range and the bf16 wire."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddist.init("gloo")
    mine = ddist.shard_range(37, rank, world)                      # 37 outfits over 8 ranks: ragged (5 x 5 + 3 x 4)
    n, bucket = 10007, 4096                                        # arena length not a multiple of the bucket nor of 8 ranks
    g = torch.Generator().manual_seed(500 + rank)
    arena = torch.randn(n, generator=g) * (10.0 ** torch.randint(-2, 3, (n,), generator=g).float())
    own = arena.clone()
    fp32, bf16 = arena.clone(), arena.clone()
    ranges = []
    hi = n
    while hi > 0:                                                  # back to front, like dfh_unet_backward_next
        lo = max(0, ((hi - 1) // bucket) * bucket)
        ranges.append((lo, hi))
        hi = lo
    for lo, hi in ranges:
        ddist.all_reduce_gradients(fp32[lo:hi])                    # the default wire: one fp32 all-reduce per range (reference DDP semantics)
        ddist.all_reduce_gradients(bf16[lo:hi], wire="bf16")       # opt-in wire: rank-order fp32 accumulation of bf16 contributions
    logged = float(ddist.gather_mean(torch.tensor(float(len(mine)))))
    q.put((rank, (mine.start, mine.stop), own.tolist(), fp32.tolist(), bf16.tolist(), ranges, logged))
    dist.destroy_process_group()


def test_world_size_8_gloo_sharding_and_in_backward_range_exchange():
    """the 8-rank shape of BASELINE configs[3] on CPU -- `shard_range` over a ragged outfit count and the range-by-range
    gradient exchange (ragged last range; fp32 wire and rank-order bf16 wire), every rank ending with identical values."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    spans = [r[1] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 37 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))       # exact partition, rank order
    assert sorted(b - a for a, b in spans) == [4, 4, 4, 5, 5, 5, 5, 5]
    assert res[0][5] == [(8192, 10007), (4096, 8192), (0, 4096)]                                                 # ragged range first
    own = torch.tensor([r[2] for r in res])
    want32 = own.double().sum(0) / world
    for r in res:
        assert r[3] == res[0][3] and r[4] == res[0][4]                                                           # replicas bit-identical
        assert abs(r[6] - 37 / 8) < 1e-6
    torch.testing.assert_close(torch.tensor(res[0][3]).double(), want32, rtol=1e-5, atol=1e-6)
    acc = own[0].bfloat16().float()
    for k in range(1, world):
        acc = acc + own[k].bfloat16().float()                                                                    # rank order, fp32
    assert torch.equal(torch.tensor(res[0][4]), (acc / world).bfloat16().float())


def test_bench_gpus_8_launcher_flow_without_gpus():
    """`python bench.py --gpus 8 --selftest-launcher`: the launcher path of the driver's N = 8 run with the GPU taken out -- launch_ranks
    starts 8 fresh ranks with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT, they rendezvous (gloo), one all-reduce
    counts them, rank 0's single JSON line is relayed and the launcher exits 0."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--selftest-launcher"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["world_size"] == 8 and rec["ranks_seen"] == 8 and rec["max_rank"] == 7 and rec["local_rank"] == 0


def test_profiles_readme_numbers_are_generated_from_the_artefacts():
    """profiles/README.md quotes each round's numbers between generated markers; the block must equal what scripts/profiles_readme.py
    derives from the committed artefacts of that round (bench JSON lines, rocprofv3 kernel_stats.csv, pmc_traffic.json), so prose and
    artefacts cannot drift apart (the r02 prose quoted 80.3 ms where the committed CSV summed to 85.05)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    readme = open(os.path.join(root, "profiles", "README.md")).read()
    tags = sorted(set(m for m in __import__("re").findall(r"<!-- (r\d\d):generated", readme)))
    assert tags, "no generated block in profiles/README.md"
    for tag in tags:
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "profiles_readme.py"), tag, "--check"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_bench_gpus_n_launches_its_own_ranks_and_fails_with_them():
    """`python bench.py --gpus 2` without a launcher environment becomes the launcher (bench.launch_ranks): it starts the ranks as fresh
    child processes and must exit non-zero -- promptly, without leaving ranks behind -- when they fail.  Here (no GPU) every rank
    refuses to start, which is exactly the failure path; the success path runs on the GPU box (tests/test_gpu_ddp.py)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the success path is covered by tests/test_gpu_ddp.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 1
    assert "ranks failed" in r.stderr and "needs an MI355X" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_wgrad_slice_plan_picks_within_a_few_percent_of_the_measured_best():
    """The decomposition of a weight-gradient launch comes from a cost model (wgrad.hip wgrad_plan) fitted to a sweep on the MI355X over
    the layer shapes of the training step (profiles/r05/wgrad_plan_sweep.txt: microseconds per shape for n pixel slices of every tile,
    `ms=n`, and for full rounds of whole tiles + the rest in n slices, `w+n`).  The plan is host code (dfh_gemm_wgrad_plan).  For every
    swept shape the candidate it picks must be one the sweep measured within 8 % of that shape's best time; slices keep >= 128 pixel
    rows; whole tiles come in full rounds of 512 blocks; and the slab request matches the plan (one 160 x 160 fp32 slot per sliced piece)."""
    import ctypes as C
    from difashion_amd import _lib
    lib = _lib.raw()
    lines = open(os.path.join(ROOT, "profiles", "r05", "wgrad_plan_sweep.txt")).read().splitlines()
    head = next(l for l in lines if l.startswith("shape"))
    cols = head.split("|")[1].split("(us)")[0].split()
    dummy = (C.c_char * 256)()
    ptr = C.addressof(dummy)
    checked = total_pick = total_best = 0
    for l in lines:
        if "|" not in l or l.startswith(("shape", "sum", "TFLOP")):
            continue
        name = l[:24].strip()
        M, N, K, tiles = (int(x) for x in l[24:].split("|")[0].split())
        vals = [float(x) for x in l.split("|")[1].split("best")[0].split()]
        times = {c: v for c, v in zip(cols, vals) if v == v and c != "plan"}
        d = _lib.GemmDesc()
        if name.startswith("conv"):
            hw = int(name.split("@")[1])
            d.conv_src, d.conv_c, d.conv = ptr, K // 9, 1
            d.batch, d.Hin, d.Win, d.stride, d.upsample = M // (hw * hw), hw, hw, 1, 0
        else:
            d.a0, d.a0_c = ptr, K
        d.M, d.N, d.zero_page = M, N, ptr
        t, w, ms = C.c_int(), C.c_int(), C.c_int()
        assert lib.dfh_gemm_wgrad_plan(C.byref(d), 0, C.byref(t), C.byref(w), C.byref(ms)) == 0
        assert t.value == tiles, (name, t.value, tiles)
        assert w.value % 512 == 0 and w.value < tiles and 1 <= ms.value <= 64 and (ms.value == 1 or M // ms.value >= 128), (name, w.value, ms.value)
        assert lib.dfh_gemm_wgrad_partial_floats(C.byref(d), 0) == (ms.value * (tiles - w.value) * 160 * 160 if ms.value > 1 else 0)
        key = f"w+{ms.value}" if w.value else f"ms={ms.value}"
        best = min(times.values())
        if key in times:                          # not every slice count was swept
            assert times[key] <= 1.08 * best, (name, key, times[key], best)
            checked += 1; total_pick += times[key]; total_best += best
    assert checked >= 16 and total_pick <= 1.03 * total_best, (checked, total_pick, total_best)
