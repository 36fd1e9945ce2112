"""Bulk tokenisation (reference scripts/pretokenization.py:150-259): record layout, the RAR jsonl and
VAR / LlamaGen npy formats as their readers consume them (data/webdataset_reader.py:253-267, dataset/imagenet.py:8-50),
and — on the GPU — tokens bit-identical to the reference's code indices on the committed model goldens."""
import json
import os

import numpy as np
import pytest
import torch

from imagefolder_amd import tokenize as tk


class _FakeQuant:
    vocab_size = 100


class _FakeModel(torch.nn.Module):
    """img_to_idx contract of VQModel: list over product branches of lists over scales"""
    def __init__(self, P=1, pns=(4,)):
        super().__init__()
        self.product_quant = P
        self.quantize = _FakeQuant()
        self.quantizes = [_FakeQuant() for _ in range(P)]
        self.pns = pns
        self.w = torch.nn.Parameter(torch.zeros(1))

    def img_to_idx(self, x):
        B = x.shape[0]
        code = (x.flatten(1).sum(1).abs() * 7).long() % 90           # depends on the image, invariant to a flip
        left = (x[:, 0, 0, 0] * 1000).long().abs() % 7               # NOT invariant to a flip
        return [[(code[:, None] + torch.arange(pn * pn)[None] + p + left[:, None]) % 100 for pn in self.pns]
                for p in range(self.product_quant)]


def test_records_flip_jsonl_and_npy_formats(tmp_path):
    torch.manual_seed(0)
    model = _FakeModel(P=2, pns=(1, 2, 3))
    batches = [(torch.randn(3, 3, 8, 8), torch.tensor([5, 6, 7])), (torch.randn(2, 3, 8, 8), torch.tensor([8, 9]))]
    bt = tk.BulkTokenizer(model, augment="flip").run(batches)
    cls, tok = bt.records
    L = 2 * (1 + 4 + 9)
    assert tok.shape == (10, L) and tok.dtype == np.int64
    assert cls.tolist() == [5, 6, 7, 5, 6, 7, 8, 9, 8, 9]                       # pretokenization.py:227-228 ordering
    assert tok[:, :14].max() < 100 and tok[:, 14:].min() >= 100                 # branch p offset by p * V
    direct = tk.tokens_from_images(model, torch.cat([batches[0][0], torch.flip(batches[0][0], dims=[-1])]))
    assert np.array_equal(tok[:6], direct.numpy())

    # RAR: per-rank json -> jsonl -> what PretoeknizedDataSetJSONL.__getitem__ returns
    bt.write_rank_json(str(tmp_path), rank=0)
    bt.write_rank_json(str(tmp_path), rank=1)                                    # a second rank's shard
    n = tk.convert_json_to_jsonl(os.path.join(tmp_path, "pretokenized_*.json"), os.path.join(tmp_path, "pretokenized.jsonl"))
    assert n == 20
    lines = open(os.path.join(tmp_path, "pretokenized.jsonl")).read().splitlines()
    assert len(lines) == 20 and set(json.loads(lines[0])) == {"class_id", "tokens"}
    c, t = tk.read_jsonl_record(os.path.join(tmp_path, "pretokenized.jsonl"), 4)
    assert c.item() == 6 and t.dtype == torch.int64 and np.array_equal(t.numpy(), tok[4])

    # VAR / LlamaGen through the public path: the two views of one image in one file, (1, n_aug, L) codes + (1,) label per source image
    c2, t2 = tk.regroup_flip(cls, tok, [3, 2])
    assert c2.tolist() == [5, 5, 6, 6, 7, 7, 8, 8, 9, 9]
    assert bt.batch_sizes == [3, 2] and bt.n_aug == 2
    n_img = bt.write_code_npy(os.path.join(tmp_path, "codes"), os.path.join(tmp_path, "labels"))
    assert n_img == 5
    for i, (label, orig, flipped) in enumerate([(5, 0, 3), (6, 1, 4), (7, 2, 5), (8, 6, 8), (9, 7, 9)]):
        feats = np.load(os.path.join(tmp_path, "codes", f"{i}.npy"))
        labels = np.load(os.path.join(tmp_path, "labels", f"{i}.npy"))
        assert feats.shape == (1, 2, L) and labels.shape == (1,) and labels[0] == label
        assert np.array_equal(feats[0, 0], tok[orig]) and np.array_equal(feats[0, 1], tok[flipped])   # CustomDataset: features[:, aug_idx]
    assert not np.array_equal(tok[0], tok[3])                      # the fake tokenizer is not flip-invariant: a mix-up would show
    with pytest.raises(ValueError, match="views per image"):
        bt.write_code_npy(os.path.join(tmp_path, "codes"), os.path.join(tmp_path, "labels"), n_aug=1)


def test_ten_crop_layout():
    model = _FakeModel()
    x = torch.randn(2, 10, 3, 8, 8)
    bt = tk.BulkTokenizer(model, augment="ten_crop").run([(x, torch.tensor([1, 2]))])
    cls, tok = bt.records
    assert cls.tolist() == [1] * 10 + [2] * 10 and tok.shape == (20, 16)
    assert bt.n_aug == 10


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["model_cfg1_cnn_vq4096", "model_cfg2_vitb_vq8192"])
def test_tokens_equal_reference_indices_on_model_goldens(oracle, name, tmp_path):
    from test_model_parity import build
    m, g = build(name)
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    out = tk.pretokenize(m, [(x, torch.tensor([3])), (x, torch.tensor([4]))], str(tmp_path), augment="none")
    c0, t0 = tk.read_jsonl_record(out, 0)
    c1, t1 = tk.read_jsonl_record(out, 1)
    assert (c0.item(), c1.item()) == (3, 4) and torch.equal(t0, t1)
    E = m.quantize.embedding.weight.detach().cpu().numpy()
    par = oracle.index_parity(g["f"], E, oracle.MODE_L2_NORMED, t0.numpy(), g["idx"], tol=2e-4)
    assert par["match_rate"] >= 0.98 and par["all_ties"], par                     # same contract as test_model_parity


TOKEN_CASES = {
    "tokens_cfg3_vp2_16384": dict(codebook_size=16384, codebook_embed_dim=32, v_patch_nums=[16], num_latent_tokens=256, product_quant=2, half_sem=True),
    "tokens_cfg4_msvr10p2_4096": dict(codebook_size=4096, codebook_embed_dim=32, v_patch_nums=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11], num_latent_tokens=121,
                                      product_quant=2, half_sem=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TOKEN_CASES))
def test_product_and_multiscale_tokens_equal_the_reference_indices(oracle, name):
    """f2 depth: the token rows of BASELINE config 3 (VP2-16384: P = 2 single-scale quantizers) and config 4 (MSVR10P2-4096: P = 2
    ten-scale ladders) against what the reference emits (xqgan_model.py:386-394 -> f_to_idxBl_or_fhat(to_fhat=False)), ViT-B
    tokenizer with deterministic weights, 256 x 256 images (oracle/make_golden.py gen_tokens).
      (i)  the HIP quantizers on the REFERENCE latents: every index of every branch and scale identical;
      (ii) images -> tokens end to end on the MI355X fp32 path, in the row layout of tokenize.tokens_from_images (branch p offset by
           p * V): rows agree except where the encoder's fp32 rounding flips a near-tie (a ladder flip re-routes the later scales of
           that sample's branch, so the bound is per row)."""
    from conftest import load_golden
    from oracle.det_init import det_state_dict
    from imagefolder_amd.xqgan_model import VQ_models
    g = load_golden(name)
    kw = dict(enc_type="dinov2", dec_type="dinov2", semantic_guide="none", detail_guide="none", abs_pos_embed=True,
              encoder_model="vit_base_patch14_dinov2.lvd142m", decoder_model="vit_base_patch14_dinov2.lvd142m", share_quant_resi=4,
              **TOKEN_CASES[name])
    seed, B = int(g["seed"]), int(g["B"])
    torch.manual_seed(seed)
    m = VQ_models["VQ-16"](**kw).eval()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    m = m.cuda()
    P, V, SN = kw["product_quant"], kw["codebook_size"], int(g["n_scales"])
    multi = SN > 1
    # (i) quantizers alone, on the reference's latents
    rows = []
    with torch.no_grad():
        for p in range(P):
            f = torch.from_numpy(g[f"f{p}"]).cuda()
            ids = m.quantizes[p].f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=kw["v_patch_nums"] if multi else None)
            assert len(ids) == SN
            for si, t in enumerate(ids):
                got = t.reshape(B, -1).cpu().numpy()
                want = g[f"idx{p}_{si}"]
                if not np.array_equal(got, want):      # only an fp64-verified tie may differ (single-scale: checkable directly)
                    assert not multi, f"branch {p} scale {si}: {(got != want).sum()} ladder indices differ from the reference"
                    E = m.quantizes[p].embedding.weight.detach().cpu().numpy()
                    par = oracle.index_parity(g[f"f{p}"], E, oracle.MODE_L2_NORMED, got.reshape(-1), want.reshape(-1))
                    assert par["all_ties"], par
                rows.append(t.reshape(B, -1).long().cpu() + p * V)
    assert np.array_equal(torch.cat(rows, dim=1).numpy(), g["tokens"]) or not multi
    # (ii) end to end
    x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(int(g["x_seed"]))) * 2 - 1).cuda()
    tok = tk.tokens_from_images(m, x).cpu().numpy()
    assert tok.shape == g["tokens"].shape and tok.dtype == np.int64
    assert tok[:, :tok.shape[1] // 2].max() < V and tok[:, tok.shape[1] // 2:].min() >= V
    agree = (tok == g["tokens"]).mean(axis=1)
    print(f"{name}: end-to-end token agreement per image {agree.tolist()}")
    assert agree.min() >= (0.9 if multi else 0.97)


def _pretok_worker(rank, world, port, cached):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    model = _FakeModel(P=1, pns=(2,))
    # every rank tokenises its own shard of the "dataset" (pretokenization.py:150-239: DistributedSampler shards, per-rank json)
    batches = [(torch.randn(3, 3, 8, 8), torch.tensor([10 * rank + 1, 10 * rank + 2, 10 * rank + 3]))]
    out = tk.pretokenize(model, batches, cached, augment="flip")
    assert os.path.basename(out) == "pretokenized.jsonl"
    dist.destroy_process_group()


def test_pretokenize_two_ranks_merge_into_one_jsonl(tmp_path):
    """scripts/pretokenization.py:236-259 under two ranks (gloo): each rank dumps pretokenized_{rank}.json, rank 0 merges them into
    pretokenized.jsonl after the barrier — 2 ranks x 3 images x 2 views = 12 records, both ranks' labels present."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_pretok_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["pretokenized.jsonl", "pretokenized_0.json", "pretokenized_1.json"]
    lines = [json.loads(l) for l in open(os.path.join(tmp_path, "pretokenized.jsonl"))]
    assert len(lines) == 12
    assert sorted(set(r["class_id"] for r in lines)) == [1, 2, 3, 11, 12, 13]
    assert all(len(r["tokens"]) == 4 for r in lines)
    c, t = tk.read_jsonl_record(os.path.join(tmp_path, "pretokenized.jsonl"), 7)
    assert t.dtype == torch.int64 and t.numel() == 4
