"""SentenceEncoder against a real (tiny, random) sentence-transformers checkpoint directory written by the test:
modules.json, 1_Pooling, 2_Dense, 3_Normalize, config_sentence_transformers.json, tokenizer files - the branch a
pretrained model takes.  The expected embeddings are the same modules applied by hand in torch fp64.
Mirrors: ec2/generate_embeddings/embeddings.py:10-40, embedders.py:1-4, compare_embeddings.py:11-12."""
import json
import os

import numpy as np
import pytest
import torch

from theoremsearch_amd.encoder import SentenceEncoder, pool_reference, read_pipeline

TEXTS = ["a tree on n vertices has n - 1 edges", "the graph", "alpha", "tree tree tree graph edge vertex the"]


def write_checkpoint(root, pooling="mean", dense=(), normalize=False, prompts=None, default_prompt=None, max_seq=16,
                     padding_side="right", extra_module=None):
    from safetensors.torch import save_file
    from transformers import BertConfig, BertModel, BertTokenizer
    os.makedirs(root, exist_ok=True)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [chr(c) for c in range(97, 123)] + \
            ["the", "tree", "graph", "##s", "edge", "edges", "vertex", "vertices", "alpha", "has", "on", "-", "1", "query", ":"]
    with open(os.path.join(root, "vocab.txt"), "w") as f:
        f.write("\n".join(vocab))
    tok = BertTokenizer(os.path.join(root, "vocab.txt"))
    tok.padding_side = padding_side
    tok.save_pretrained(root)
    torch.manual_seed(7)
    cfg = BertConfig(vocab_size=len(vocab), hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=64, max_position_embeddings=64)
    BertModel(cfg, add_pooling_layer=False).save_pretrained(root)
    json.dump({"max_seq_length": max_seq, "do_lower_case": False}, open(os.path.join(root, "sentence_bert_config.json"), "w"))
    modules = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    os.makedirs(os.path.join(root, "1_Pooling"), exist_ok=True)
    keys = {"cls": "pooling_mode_cls_token", "mean": "pooling_mode_mean_tokens", "max": "pooling_mode_max_tokens",
            "mean_sqrt_len": "pooling_mode_mean_sqrt_len_tokens", "lasttoken": "pooling_mode_lasttoken",
            "weightedmean": "pooling_mode_weightedmean_tokens"}
    pcfg = {"word_embedding_dimension": 32, "include_prompt": True}
    pcfg.update({v: (k == pooling) for k, v in keys.items()})
    json.dump(pcfg, open(os.path.join(root, "1_Pooling", "config.json"), "w"))
    weights = []
    in_f = 32
    for j, (out_f, bias, act) in enumerate(dense):
        name = f"{2 + j}_Dense"
        os.makedirs(os.path.join(root, name), exist_ok=True)
        json.dump({"in_features": in_f, "out_features": out_f, "bias": bias, "activation_function": act},
                  open(os.path.join(root, name, "config.json"), "w"))
        lin = torch.nn.Linear(in_f, out_f, bias=bias)
        save_file({f"linear.{k}": v.detach().clone() for k, v in lin.state_dict().items()}, os.path.join(root, name, "model.safetensors"))
        weights.append((lin, act))
        modules.append({"idx": len(modules), "name": str(len(modules)), "path": name, "type": "sentence_transformers.models.Dense"})
        in_f = out_f
    if normalize:
        modules.append({"idx": len(modules), "name": str(len(modules)), "path": f"{len(modules)}_Normalize",
                        "type": "sentence_transformers.models.Normalize"})
    if extra_module:
        modules.append({"idx": len(modules), "name": str(len(modules)), "path": "x", "type": extra_module})
    json.dump(modules, open(os.path.join(root, "modules.json"), "w"))
    json.dump({"prompts": prompts or {}, "default_prompt_name": default_prompt, "similarity_fn_name": "cosine"},
              open(os.path.join(root, "config_sentence_transformers.json"), "w"))
    return weights


def by_hand(root, texts, pooling, weights, normalize, max_seq=16):
    """The checkpoint's modules applied one by one in fp64."""
    from transformers import AutoModel, AutoTokenizer
    tok = AutoTokenizer.from_pretrained(root)
    model = AutoModel.from_pretrained(root).double().eval()
    out = []
    with torch.inference_mode():
        for t in texts:                                   # one text at a time: no padding at all
            enc = tok([t], padding=True, truncation=True, max_length=max_seq, return_tensors="pt")
            h = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]).last_hidden_state
            e = pool_reference(h, enc["attention_mask"], pooling)
            for lin, act in weights:
                e = lin.double()(e)
                if act.endswith("Tanh"):
                    e = torch.tanh(e)
            if normalize:
                e = torch.nn.functional.normalize(e, p=2, dim=1)
            out.append(e[0])
    return torch.stack(out).numpy()


@pytest.mark.parametrize("pooling,padding_side", [("mean", "right"), ("cls", "right"), ("lasttoken", "left"),
                                                  ("lasttoken", "right"), ("max", "right"), ("mean_sqrt_len", "right")])
def test_checkpoint_pipeline_is_read_from_the_checkpoint(tmp_path, pooling, padding_side):
    root = str(tmp_path / "ckpt")
    weights = write_checkpoint(root, pooling=pooling, padding_side=padding_side)
    enc = SentenceEncoder(root, device="cpu")
    assert enc.pretrained and enc.pooling == pooling and enc.max_seq_length == 16
    assert enc.model.dtype == torch.float32                      # pretrained weights run in fp32 unless asked otherwise
    got = enc.encode(TEXTS, batch_size=3)                        # batched: padding on the checkpoint's side
    want = by_hand(root, TEXTS, pooling, weights, False)
    assert got.shape == (4, 32) and np.allclose(got, want, atol=2e-5), np.abs(got - want).max()
    unit = enc.encode(TEXTS, normalize_embeddings=True)
    assert np.allclose(unit, want / np.linalg.norm(want, axis=1, keepdims=True), atol=2e-5)


def test_dense_normalize_and_prompts_like_embeddinggemma(tmp_path):
    """google/embeddinggemma-300m (ec2/generate_embeddings/embedders.py:3) = Transformer + mean Pooling + two Dense modules
    (no bias, identity) + Normalize, with named prompts: everything must come from the files."""
    root = str(tmp_path / "gemma_like")
    ident = "torch.nn.modules.linear.Identity"
    weights = write_checkpoint(root, pooling="mean", dense=[(48, False, ident), (24, True, "torch.nn.modules.activation.Tanh")],
                               normalize=True, prompts={"query": "query : ", "document": "the "}, default_prompt="document")
    pipe = read_pipeline(root)
    assert pipe.normalize and len(pipe.dense) == 2 and pipe.default_prompt_name == "document"
    enc = SentenceEncoder(root, device="cpu")
    assert enc.get_sentence_embedding_dimension() == 24
    got = enc.encode(TEXTS)                                       # default prompt "document", Normalize module: unit rows
    want = by_hand(root, ["the " + t for t in TEXTS], "mean", weights, True)
    assert np.allclose(got, want, atol=2e-5) and np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    got_q = enc.encode(TEXTS, prompt_name="query")
    want_q = by_hand(root, ["query : " + t for t in TEXTS], "mean", weights, True)
    assert np.allclose(got_q, want_q, atol=2e-5) and not np.allclose(got_q, got, atol=1e-3)
    with pytest.raises(ValueError):
        enc.encode(TEXTS, prompt_name="nope")


def test_truncation_uses_the_checkpoints_max_seq_length(tmp_path):
    root = str(tmp_path / "short")
    weights = write_checkpoint(root, pooling="mean", max_seq=6)
    enc = SentenceEncoder(root, device="cpu")
    long = ["tree " * 40]
    assert np.allclose(enc.encode(long), by_hand(root, long, "mean", weights, False, max_seq=6), atol=2e-5)


def test_unsupported_modules_raise_instead_of_guessing(tmp_path):
    root = str(tmp_path / "odd")
    write_checkpoint(root, extra_module="sentence_transformers.models.LayerNorm")
    with pytest.raises(NotImplementedError):
        SentenceEncoder(root, device="cpu")
    root2 = str(tmp_path / "odd2")
    write_checkpoint(root2, pooling="weightedmean")
    with pytest.raises(NotImplementedError):
        SentenceEncoder(root2, device="cpu")


def test_missing_checkpoint_raises_unless_random_init_is_asked_for(monkeypatch):
    monkeypatch.delenv("TS_ALLOW_RANDOM_ENCODER", raising=False)
    with pytest.raises(FileNotFoundError) as e:
        SentenceEncoder("Qwen/Qwen3-Embedding-0.6B", device="cpu")
    assert "allow_random_init" in str(e.value)
    from theoremsearch_amd import generate_embeddings
    with pytest.raises(FileNotFoundError):
        generate_embeddings.get_embedder("gemma")
    monkeypatch.setenv("TS_ALLOW_RANDOM_ENCODER", "1")
    enc = SentenceEncoder("Qwen/Qwen3-Embedding-0.6B", device="cpu", num_layers=1)
    assert not enc.pretrained and enc.pooling == "lasttoken" and enc.embedding_dim == 1024


def test_encode_multi_process_fans_out_over_replicas(tmp_path):
    """ec2/generate_embeddings/embeddings.py:32-38: a page is split over worker processes, each with its own replica of
    the model; the concatenation is in input order and equals the single-process result (replicas only)."""
    root = str(tmp_path / "ckpt")
    write_checkpoint(root, pooling="mean", normalize=True)
    enc = SentenceEncoder(root, device="cpu")
    texts = [TEXTS[i % 4] + " " + "tree " * (i % 5) for i in range(23)]
    want = enc.encode(texts, batch_size=4)
    pool = enc.start_multi_process_pool(["cpu", "cpu"])
    try:
        got = enc.encode_multi_process(texts, pool=pool, batch_size=4)
        assert got.shape == want.shape and np.allclose(got, want, atol=1e-6)
        got2 = enc.encode_multi_process(texts, pool=pool, batch_size=4, chunk_size=5, normalize_embeddings=True)
        assert np.allclose(got2, want, atol=1e-6)
    finally:
        enc.stop_multi_process_pool(pool)
    # pool=None with fewer than two visible GPUs: the same call runs in this process
    assert np.allclose(enc.encode_multi_process(texts, pool=None, batch_size=4), want, atol=1e-6)


def test_encode_multi_process_notices_a_dead_replica(tmp_path):
    """a worker that dies without answering (killed for memory, a GPU fault) must not leave the caller waiting
    forever: the call raises and the pool is stopped."""
    root = str(tmp_path / "ckpt")
    write_checkpoint(root, pooling="mean", normalize=True)
    enc = SentenceEncoder(root, device="cpu")
    texts = [TEXTS[i % 4] for i in range(12)]
    pool = enc.start_multi_process_pool(["cpu", "cpu"])
    try:
        assert enc.encode_multi_process(texts, pool=pool, batch_size=4).shape[0] == 12     # both replicas are up
        pool["workers"][1][0].kill()                                                      # ... and one is gone
        pool["workers"][1][0].join(timeout=10)
        with pytest.raises(RuntimeError, match="died"):
            enc.encode_multi_process(texts, pool=pool, batch_size=4)
        assert not any(p.is_alive() for p, _, _ in pool["workers"])
    finally:
        enc.stop_multi_process_pool(pool)


def test_the_qwen3_stand_in_is_a_qwen3_shaped_model():
    """allow_random_init for the production embedder's name (streamlit_app.py:55) builds a Qwen3Model of the published shape
    (1024 wide, 16 query / 8 key-value heads of 128, gated MLP 3072, RMSNorm, rotary positions) with last-token pooling -
    not a BERT of that width; its forward on CPU is the model's own (the fused one needs the HIP kernels)."""
    enc = SentenceEncoder("Qwen/Qwen3-Embedding-0.6B", allow_random_init=True, num_layers=2, device="cpu")
    cfg = enc.model.config
    assert type(enc.model).__name__ == "Qwen3Model" and enc._fused is None
    assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size) == (1024, 16, 8, 128, 3072)
    assert enc.pooling == "lasttoken" and enc.embedding_dim == 1024
    texts = ["a tree on n vertices has n - 1 edges", "x", "every bounded monotone sequence converges"]
    a = enc.encode(texts, normalize_embeddings=True)
    assert a.shape == (3, 1024) and np.allclose((a * a).sum(1), 1.0, atol=1e-5)
    # causal + right padding: a sentence's embedding does not depend on what it is batched with
    b = enc.encode(texts[1:2], normalize_embeddings=True)
    assert np.allclose(a[1], b[0], atol=1e-5)


def test_random_stand_ins_have_the_published_shapes_of_the_three_embedders():
    """allow_random_init builds the architecture of the model that was asked for (no weights offline): BERT-base for the legacy
    embedder, Qwen3 (last-token pooling, 1024-d) for the production one, a Gemma3 text model with bidirectional attention, mean
    pooling, two Dense modules and Normalize for google/embeddinggemma-300m (ec2/generate_embeddings/embedders.py:1-4)."""
    import numpy as np
    from theoremsearch_amd.encoder import SentenceEncoder
    enc = SentenceEncoder("google/embeddinggemma-300m", allow_random_init=True, num_layers=1, device="cpu")
    cfg = enc.model.config
    assert type(enc.model).__name__ == "Gemma3TextModel" and cfg.use_bidirectional_attention
    assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size) == (768, 3, 1, 256, 1152)
    assert enc.pooling == "mean" and len(enc.pipeline.dense) == 2 and enc.pipeline.normalize and enc.embedding_dim == 768
    out = enc.encode(["Let $G$ be a finite group.", "Every compact metric space is separable, and more words follow here."])
    assert out.shape == (2, 768) and np.allclose(np.sum(out * out, axis=1), 1.0, atol=1e-4)
    q = SentenceEncoder("Qwen/Qwen3-Embedding-0.6B", allow_random_init=True, num_layers=1, device="cpu")
    assert type(q.model).__name__ == "Qwen3Model" and q.pooling == "lasttoken" and q.embedding_dim == 1024
