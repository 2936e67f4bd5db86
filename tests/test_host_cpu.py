"""CPU tests of the host-side mirror of the reference API (no GPU: construction, state-dict keys, checkpoint migration,
factory flag semantics, tokenizer).  Compute calls are covered by the -m gpu suites."""
import os
import types

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _args(**mc):
    a = types.SimpleNamespace()
    a.model_config = types.SimpleNamespace(**mc)
    return a


def test_state_dict_keys_equal_the_reference_modules():
    """keys captured from the reference's own SimpleCLIP(CLIBDImageEncoder, CLIBDDNAEncoder, CLIBDLanguageEncoder)"""
    from clibd_amd.model import (BertConfigLite, BertForMaskedLM, BertModel, CLIBDDNAEncoder, CLIBDImageEncoder, CLIBDLanguageEncoder,
                                 SimpleCLIP, VisionTransformer)

    gs = torch.load(os.path.join(G, "step_tiny_golden.pt"), map_location="cpu", weights_only=False)
    tiny = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    m = SimpleCLIP(CLIBDImageEncoder(VisionTransformer(embed_dim=128, depth=2, num_heads=2, num_classes=10), 4, 128),
                   CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **tiny)), 4, 128),
                   CLIBDLanguageEncoder(BertModel(BertConfigLite(vocab_size=1000, **tiny)), 4, 128))
    assert list(m.state_dict().keys()) == gs["state_dict_keys"]
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert all(("linear_a_" in n or "linear_b_" in n or ".w_a." in n or ".w_b." in n or "head." in n or "decoder" in n or "proj." in n
                or n == "logit_scale") for n in trainable)


def test_trainable_parameter_counts_match_survey():
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    m = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), 4, 768), CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), 4, 768), None)
    n = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n == 1_476_097  # 738,048 (image) + 738,048 (DNA) + logit_scale


def test_lora_layer_quirks_and_rank():
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, CLIBDImageEncoder, VisionTransformer
    from clibd_amd.model.dna_encoder import _LoRALayer
    from clibd_amd.model.image_encoder import _LoRA_qkv_timm

    ie = CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=2, num_heads=1, num_classes=0), 4, 32, lora_layer=[])
    assert all(isinstance(b.attn.qkv, _LoRA_qkv_timm) for b in ie.base_image_encoder.blocks)  # `if lora_layer:` quirk
    de = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=64, num_hidden_layers=2, num_attention_heads=1,
                                                        intermediate_size=128)), 4, 32, lora_layer=[])
    assert not any(isinstance(l.attention.self.query, _LoRALayer) for l in de.base_dna_encoder.bert.encoder.layer)
    assert float(ie.base_image_encoder.blocks[0].attn.qkv.linear_b_q.weight.abs().sum()) == 0.0
    with pytest.raises(AssertionError):
        CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=1, num_heads=1), 0)
    assert tuple(CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=1, num_heads=1), 9).w_As[0].weight.shape) == (9, 64)   # any r > 0, as the reference (round 4)
    assert tuple(CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=1, num_heads=1), 8).w_Bs[0].weight.shape) == (64, 8)
    assert tuple(CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=1, num_heads=1), 2).w_As[0].weight.shape) == (2, 64)


def test_load_clip_model_flag_semantics():
    from clibd_amd.model import load_clip_model

    a = _args(output_dim=768, image=types.SimpleNamespace(input_type="image", pre_train_model="vit_small_patch16_224"),
              dna=types.SimpleNamespace(input_type="sequence"), disable_lora=False)
    a.bioscan_bert_checkpoint = "/nonexistent/ckpt.pth"
    m = load_clip_model(a)
    assert m.language_encoder is None and m.image_encoder is not None and m.dna_encoder is not None
    assert not m.image_encoder.base_image_encoder.blocks[0].mlp.fc1.weight.requires_grad
    assert m.image_encoder.base_image_encoder.head.weight.requires_grad and m.logit_scale.requires_grad
    # using_open_clip overwrites disable_lora (reference quirk, simple_clip.py:114-116)
    b = _args(output_dim=768, dna=types.SimpleNamespace(input_type="sequence", freeze=True), disable_lora=True, using_open_clip=False)
    b.bioscan_bert_checkpoint = None
    m2 = load_clip_model(b)
    assert not any(p.requires_grad for p in m2.dna_encoder.parameters())  # freeze
    c = _args(output_dim=768, for_bio_clip=True)
    with pytest.raises(NotImplementedError):
        load_clip_model(c)


def test_checkpoint_name_migration_and_roundtrip(tmp_path):
    from clibd_amd.checkpoint import load_reference_checkpoint, update_checkpoint_param_names
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, SimpleCLIP

    tiny = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128)
    m = SimpleCLIP(None, CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **tiny)), 4, 32), None)
    sd = m.state_dict()
    # a checkpoint as an old reference version + DDP would have written it
    legacy = {"module." + k.replace("base_dna_encoder", "lora_barcode_bert"): v.clone() + 1.0 for k, v in sd.items()}
    assert any("lora_barcode_bert" in k for k in legacy)
    renamed = update_checkpoint_param_names({k[len("module."):]: v for k, v in legacy.items()})
    assert sorted(renamed) == sorted(sd)
    path = tmp_path / "best.pth"
    torch.save(legacy, path)
    res = load_reference_checkpoint(m, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.allclose(m.state_dict()["logit_scale"], sd["logit_scale"] + 0)  # loaded in place
    k0 = "dna_encoder.base_dna_encoder.bert.embeddings.word_embeddings.weight"
    assert torch.equal(m.state_dict()[k0], legacy["module." + k0.replace("base_dna_encoder", "lora_barcode_bert")])


def test_sequence_pipeline_matches_oracle():
    from clibd_amd.model import get_sequence_pipeline
    from oracle import clibd_oracle as O

    pipe = get_sequence_pipeline()
    for s in ("ACGTA" * 132, "ACGTAC", "A" * 1000, "ACGTN" * 100, ""):
        assert pipe(s) == O.kmer_tokenize(s)


def test_scale_learning_rate():
    from clibd_amd.train import scale_learning_rate

    assert scale_learning_rate(0.001, 500, world_size=4) == pytest.approx(0.004)


def test_reference_checkpoint_with_hf_position_id_buffers_loads_strict(tmp_path):
    """transformers==4.29.2 (the reference's pin) stores `embeddings.position_ids` as a persistent buffer in best.pth / last.pth
    of both BERT towers; the strict load must drop them."""
    from clibd_amd.checkpoint import load_reference_checkpoint, load_training_state
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, BertModel, CLIBDDNAEncoder, CLIBDLanguageEncoder, SimpleCLIP

    tiny = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128)
    m = SimpleCLIP(None, CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **tiny)), 4, 32),
                   CLIBDLanguageEncoder(BertModel(BertConfigLite(vocab_size=100, **tiny)), 4, 32))
    ck = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    ck["module.dna_encoder.base_dna_encoder.bert.embeddings.position_ids"] = torch.arange(512)[None]
    ck["module.language_encoder.base_language_encoder.embeddings.position_ids"] = torch.arange(512)[None]
    ck["module.language_encoder.base_language_encoder.embeddings.token_type_ids"] = torch.zeros(1, 512, dtype=torch.long)
    path = tmp_path / "last.pth"
    torch.save(ck, path)
    res = load_reference_checkpoint(m, str(path))          # strict=True
    assert not res.missing_keys and not res.unexpected_keys
    torch.save({"model": ck, "epoch": 3}, tmp_path / "state.pth")
    assert load_training_state(str(tmp_path / "state.pth"), m) == 3


def test_load_clip_model_accepts_the_reference_simclr_checkpoint_key(tmp_path):
    """model_config.image.image_encoder_trained_with_simclr_style_ckpt_path (reference simple_clip.py:154-165): {"state_dict": ...}
    with a DDP prefix, loaded strictly into the ViT body."""
    from clibd_amd.model import create_vit, load_clip_model

    vit = create_vit("vit_small_patch16_224")
    with torch.no_grad():
        vit.cls_token.fill_(0.125)
    path = tmp_path / "simclr.pth"
    torch.save({"state_dict": {"module." + k: v for k, v in vit.state_dict().items()}}, path)
    a = _args(output_dim=128, image=types.SimpleNamespace(input_type="image", pre_train_model="vit_small_patch16_224",
                                                          image_encoder_trained_with_simclr_style_ckpt_path=str(path)))
    m = load_clip_model(a)
    assert float(m.image_encoder.base_image_encoder.cls_token.mean()) == 0.125
    torch.save({"state_dict": {"module.bogus": torch.zeros(1)}}, path)
    with pytest.raises(RuntimeError):
        load_clip_model(a)


def test_gradient_groups_cover_the_trainable_parameters_in_backward_order():
    """towers._Tower.grad_groups (the unit of the bucketed data-parallel all-reduce, train.Trainer): every parameter a tower
    can produce a gradient for sits in exactly one group; head first, layers top -> bottom, embeddings last; and the trainer
    lays the flat bucket out in that order, logit_scale at the end."""
    from clibd_amd.model import (BertConfigLite, BertForMaskedLM, BertModel, CLIBDDNAEncoder, CLIBDImageEncoder, CLIBDLanguageEncoder,
                                 SimpleCLIP, VisionTransformer)
    from clibd_amd.train import _backward_order, _gradient_reachable

    ie = CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=3, num_heads=1, num_classes=0), 4, 32)
    cfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128)
    de = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **cfg)), 4, 32)
    te = CLIBDLanguageEncoder(BertModel(BertConfigLite(vocab_size=100, **cfg)), 4, 32)
    model = SimpleCLIP(ie, de, te)
    for p in model.parameters():
        p.requires_grad_(True)                     # full fine-tune: every group is populated
    for enc, depth in ((ie, 3), (de, 2), (te, 2)):
        tw = enc.tower()
        groups = tw.grad_groups()
        assert len(groups) == depth + 2
        flat = [id(p) for g in groups for p in g]
        assert len(flat) == len(set(flat)) and set(flat) == {id(p) for p in tw.trainable_params()}
        names = {id(p): n for n, p in enc.named_parameters()}
        for k, g in enumerate(groups[1:-1]):       # layer groups: top layer first
            layer = depth - 1 - k
            assert all(f".{layer}." in names[id(p)] for p in g), (k, [names[id(p)] for p in g])
    params = _gradient_reachable(model, None)
    ordered, counts = _backward_order(model, params)
    assert {id(p) for p in ordered} == {id(p) for p in params} and len(ordered) == len(params)
    assert ordered[-1] is model.logit_scale and [len(c) for c in counts] == [5, 4, 4]
    assert sum(sum(c) for c in counts) == len(ordered) - 1


def test_exported_checkpoint_carries_the_reference_position_id_buffers(tmp_path):
    """a best.pth written here must load STRICTLY in the reference, whose pinned transformers 4.29.2 keeps
    `embeddings.position_ids` as a persistent buffer of both BERT towers; and it must still load here."""
    from clibd_amd.checkpoint import export_reference_state_dict, load_reference_checkpoint, save_reference_checkpoint
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, BertModel, CLIBDDNAEncoder, CLIBDLanguageEncoder, SimpleCLIP

    tiny = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128)
    m = SimpleCLIP(None, CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **tiny)), 4, 32),
                   CLIBDLanguageEncoder(BertModel(BertConfigLite(vocab_size=100, **tiny)), 4, 32))
    sd = export_reference_state_dict(m)
    extra = sorted(set(sd) - set(m.state_dict()))
    assert extra == ["dna_encoder.base_dna_encoder.bert.embeddings.position_ids",
                     "language_encoder.base_language_encoder.embeddings.position_ids"]
    for k in extra:
        n = sd[k.replace("position_ids", "position_embeddings.weight")].shape[0]
        assert sd[k].dtype == torch.int64 and torch.equal(sd[k], torch.arange(n)[None])
    path = tmp_path / "best.pth"
    save_reference_checkpoint(m, str(path))
    res = load_reference_checkpoint(m, str(path))
    assert not res.missing_keys and not res.unexpected_keys


class _FakeFlatOptimizer:
    """The attributes of optim.FusedAdamW that checkpoint.py touches (the real one needs a GPU)."""

    def __init__(self, params):
        self.param_groups = [dict(params=list(params), lr=1e-3)]
        self._offsets, off = [], 0
        for p in params:
            self._offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.exp_avg, self.exp_avg_sq, self.step_count = torch.zeros(off), torch.zeros(off), 0


def test_training_state_records_and_remaps_the_optimizer_layout(tmp_path):
    """the flat moment buffers are saved with (name, offset, numel) per parameter; a state saved under another
    parameter order is re-mapped by name, one with other parameters (or no layout and another size) is refused loudly."""
    from clibd_amd.checkpoint import load_training_state, save_training_state
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, SimpleCLIP

    tiny = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128)
    m = SimpleCLIP(None, CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **tiny)), 4, 32), None)
    train = [p for p in m.parameters() if p.requires_grad]
    assert len(train) >= 4
    a = _FakeFlatOptimizer(train)
    for i, (p, off) in enumerate(zip(train, a._offsets)):
        a.exp_avg[off:off + p.numel()] = i + 1.0
        a.exp_avg_sq[off:off + p.numel()] = 10.0 * (i + 1)
    a.step_count = 7
    path = str(tmp_path / "state.pth")
    save_training_state(path, m, a, epoch=2)
    b = _FakeFlatOptimizer(train)                     # same layout: taken as is
    assert load_training_state(path, m, b) == 2 and b.step_count == 7
    assert torch.equal(b.exp_avg, a.exp_avg) and torch.equal(b.exp_avg_sq, a.exp_avg_sq)
    c = _FakeFlatOptimizer(train[::-1])               # another order (e.g. other towers / fix_temperature): moved by name
    load_training_state(path, m, c)
    for i, p in enumerate(train):
        j = len(train) - 1 - i
        off = c._offsets[j]
        assert float(c.exp_avg[off]) == i + 1.0 and float(c.exp_avg_sq[off + p.numel() - 1]) == 10.0 * (i + 1)
    extra = torch.nn.Parameter(torch.zeros(5))
    d = _FakeFlatOptimizer(train + [extra])           # a parameter the state has no moments for
    with pytest.raises(ValueError, match="lacks optimizer moments"):
        load_training_state(path, m, d)
    st = torch.load(path, weights_only=False)
    del st["optimizer"]["layout"]                      # a state from before the layout was recorded
    torch.save(st, path)
    load_training_state(path, m, _FakeFlatOptimizer(train))
    with pytest.raises(ValueError, match="no optimizer layout"):
        load_training_state(path, m, d)


def _cpu_adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """torch.optim.AdamW arithmetic on the flat bucket (the contract of clibd_adamw_step, include/clibd_hip.h)."""
    with torch.no_grad():
        gg = g * grad_scale
        p.mul_(1 - lr * weight_decay)
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        p.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt().add_(eps), value=-lr)


@pytest.mark.parametrize("sched", ["one_cycle", "cosine", "exponential", "step"])
def test_fused_adamw_follows_the_references_lr_schedulers(monkeypatch, sched):
    """`FusedAdamW` is a real torch.optim.Optimizer, so the reference's schedulers (scripts/train_cl.py:222-246: OneCycleLR with
    pct_start 0.3 / cos / cycle_momentum=False, CosineAnnealingLR, ExponentialLR, StepLR) drive `param_groups[0]['lr']`, which
    the fused step reads every call: ten steps under each scheduler equal torch.optim.AdamW under the same scheduler.  The
    device kernel is replaced by its CPU statement here (the kernel itself: tests/test_ops_gpu.py::test_adamw_matches_torch and
    ::test_fused_adamw_under_one_cycle_lr_on_the_gpu)."""
    from torch.optim import lr_scheduler

    from clibd_amd import optim

    monkeypatch.setattr(optim.ops, "adamw_step", _cpu_adamw_step)
    monkeypatch.setattr(optim.FusedAdamW, "_check_params", staticmethod(lambda ps, dev: None))
    g = torch.Generator().manual_seed(31)
    shapes = [(4, 96), (96, 4), (33,), ()]
    init = [torch.randn(s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(t.clone()) for t in init]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    lr = 1e-3 * 2048 / 500          # util.scale_learning_rate at the metric's global batch
    fo = optim.FusedAdamW(mine, lr=lr)
    ro = torch.optim.AdamW(ref, lr=lr)
    total = 10

    def make(o):
        if sched == "one_cycle":
            return lr_scheduler.OneCycleLR(o, max_lr=4 * lr, total_steps=total, pct_start=0.3, anneal_strategy="cos", cycle_momentum=False)
        if sched == "cosine":
            return lr_scheduler.CosineAnnealingLR(o, T_max=total, eta_min=1e-9)
        if sched == "exponential":
            return lr_scheduler.ExponentialLR(o, gamma=0.95)
        return lr_scheduler.StepLR(o, step_size=3, gamma=0.5)

    fs, rs = make(fo), make(ro)
    lrs = []
    for step in range(total):
        fo.zero_grad()
        ro.zero_grad()
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g)
            a.grad.copy_(gr)            # gradients live as views of the flat bucket
            b.grad = gr.clone()
        fo.step()
        ro.step()
        fs.step()
        rs.step()
        assert fo.param_groups[0]["lr"] == ro.param_groups[0]["lr"]
        lrs.append(fo.param_groups[0]["lr"])
    assert len(set(lrs)) > 1                      # the schedule moved, and the fused step saw it
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (sched, (a - b).abs().max())


def test_numerics_switches_are_explicit_settings(monkeypatch, tmp_path):
    """the backward's arithmetic switches are per-stack settings.  The CLIBD_* environment variables only give the default
    a tower is CONSTRUCTED with; `SimpleCLIP.set_numerics` / `Trainer(numerics=...)` change them per model; unknown names or values
    raise; `save_training_state` records what produced a checkpoint; the fp8 tower selection validates its names."""
    from clibd_amd import engine
    from clibd_amd.checkpoint import save_training_state
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, VisionTransformer

    def build():
        ie = CLIBDImageEncoder(VisionTransformer(embed_dim=64, depth=1, num_heads=1, num_classes=0), 4, 32)
        de = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                                                            intermediate_size=128)), 4, 32)
        return SimpleCLIP(ie, de, None)

    monkeypatch.delenv("CLIBD_GELU_GRAD", raising=False)
    monkeypatch.delenv("CLIBD_RESIDUAL_GRAD", raising=False)
    monkeypatch.delenv("CLIBD_LN_FOLD", raising=False)
    monkeypatch.delenv("CLIBD_DGRAD", raising=False)
    a = build()
    assert a.numerics()["image_encoder"] == dict(residual_grad="bf16", gelu_grad="bf16", attn_bwd="2phase", ln_fold="off", dgrad="bf16", forward="bf16")
    monkeypatch.setenv("CLIBD_GELU_GRAD", "u8")
    b = build()                                    # the variable is read at construction ...
    assert b.numerics()["dna_encoder"]["gelu_grad"] == "u8" and a.numerics()["dna_encoder"]["gelu_grad"] == "bf16"   # ... not by `a`
    a.set_numerics(residual_grad="fp32")           # two models of one process differ
    assert a.numerics()["image_encoder"]["residual_grad"] == "fp32" and b.numerics()["image_encoder"]["residual_grad"] == "bf16"
    with pytest.raises(ValueError):
        a.set_numerics(residual_grad="fp16")
    with pytest.raises(ValueError):
        a.set_numerics(no_such_switch="x")
    assert engine.check_numerics(dict(attn_bwd="sp")) == dict(attn_bwd="sp")
    path = tmp_path / "state.pth"
    save_training_state(str(path), a, epoch=3)
    st = torch.load(str(path), map_location="cpu", weights_only=False)
    assert st["numerics"]["image_encoder"]["residual_grad"] == "fp32" and st["epoch"] == 3
    with pytest.raises(ValueError):
        a.enable_fp8_forward(towers=("vision_tower",))
    assert SimpleCLIP.FP8_TOWER_SETS["pooled"] == ("dna_encoder", "language_encoder")
    # round 5: the 8-bit dgrad is a per-tower numerics switch; the model-level selection sets it tower by tower
    a.enable_fp8_dgrad(towers="pooled")
    assert a.numerics()["dna_encoder"]["dgrad"] == "fp8" and a.numerics()["image_encoder"]["dgrad"] == "bf16"
    a.enable_fp8_dgrad(towers="all")
    assert a.numerics()["image_encoder"]["dgrad"] == "fp8"
    a.enable_fp8_dgrad(enabled=False)
    assert all(v["dgrad"] == "bf16" for v in a.numerics().values())
    with pytest.raises(ValueError):
        a.enable_fp8_dgrad(towers=("vision_tower",))
    assert SimpleCLIP.FP8_TOWER_SETS["pooled_ffn"] == ("dna_encoder", "language_encoder") and SimpleCLIP.FP8_TOWER_SITES["pooled_ffn"]["dna_encoder"] == ("fc1_in", "fc2_in")
