"""`SimpleCLIP` and `load_clip_model`: drop-in for `bioscanclip.model.simple_clip` (reference
model/simple_clip.py:21-61, 100-246) on the MI355X HIP engine.

forward(image_input, dna_input, language_input) ->
    (image_output, dna_output, language_output, logit_scale.exp(), logit_bias)
with every present tower output L2-normalised (K8 kernel) and `None` for absent towers — the reference contract.
The open_clip / BioCLIP branches of the reference (simple_clip.py:137-146) are outside the hot path
and raise.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .dna_encoder import CLIBDDNAEncoder, load_pre_trained_bioscan_bert
from .image_encoder import CLIBDImageEncoder, create_vit
from .language_encoder import CLIBDLanguageEncoder, load_pre_trained_bert

F32 = torch.float32


class _L2NormFn(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) (simple_clip.py:45,58,60)."""

    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_fwd(x.detach().to(F32).contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy.to(F32).contiguous(), y, inv)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    return _L2NormFn.apply(x)


class SimpleCLIP(nn.Module):
    def __init__(self, image_encoder, dna_encoder, language_encoder, open_clip_model=None, init_logit_scale: float = np.log(1 / 0.07),
                 init_logit_bias: Optional[float] = None, for_bio_clip=False):
        super().__init__()
        if open_clip_model is not None or for_bio_clip:
            raise NotImplementedError("open_clip / BioCLIP towers are not part of the MI355X hot path (SURVEY §2, row 3)")
        self.image_encoder = image_encoder
        self.dna_encoder = dna_encoder
        self.language_encoder = language_encoder
        self.open_clip_model = None
        self.tokenizer_for_open_clip = None
        self.logit_scale = nn.Parameter(torch.ones([]) * init_logit_scale)
        if init_logit_bias is not None:
            self.logit_bias = nn.Parameter(torch.ones([]) * init_logit_bias)
        else:
            self.logit_bias = None

    # The towers are independent until the loss, so each one runs on its own HIP stream: the persistent GEMM grids are one
    # workgroup per CU, and the CUs a kernel's last partial round leaves idle (up to ~25 % of a 591-tile projection GEMM)
    # pick up the other tower's workgroups instead of waiting.  autograd replays each tower's backward on its forward
    # stream, so the backward overlaps the same way.  CLIBD_TOWER_STREAMS=0 keeps everything on the current stream.
    overlap_towers = os.environ.get("CLIBD_TOWER_STREAMS", "1") != "0"

    def _side_streams(self, device):
        cache = self.__dict__.setdefault("_streams", {})
        if device not in cache:
            # HIGH priority, so that the side streams never share a hardware queue with the current stream.  ROCm multiplexes
            # the streams of one priority onto GPU_MAX_HW_QUEUES (4) hardware queues, and two streams on one queue run their kernels
            # back to back: once torch.distributed has created its process group (RCCL brings streams of its own) the towers'
            # streams landed on the current stream's queue and the overlap was gone — 284 -> 296 ms at b=2048, 37.7 -> 42.0 ms
            # at b=256, i.e. in every multi-GPU run (found with the one-rank RCCL group of CLIBD_FORCE_COLLECTIVES).
            # High-priority streams draw from their own queue pool.  CLIBD_TOWER_STREAM_PRIORITY=0 restores normal priority.
            prio = int(os.environ.get("CLIBD_TOWER_STREAM_PRIORITY", "-1"))
            cache[device] = (torch.cuda.Stream(device=device, priority=prio), torch.cuda.Stream(device=device, priority=prio))
        return cache[device]

    def forward(self, image_input, dna_input, language_input):
        towers = [(self.dna_encoder, dna_input), (self.image_encoder, image_input), (self.language_encoder, language_input)]
        present = [i for i, (enc, x) in enumerate(towers) if enc is not None]
        outs = [None, None, None]
        dev = self.logit_scale.device
        if not (self.overlap_towers and dev.type == "cuda" and len(present) > 1):
            for i in present:
                outs[i] = l2_normalize(towers[i][0](towers[i][1]))
        else:
            main = torch.cuda.current_stream(dev)
            sides = self._side_streams(dev)
            # the image tower (largest) stays on the current stream; DNA and text go to side streams
            side_of = {0: sides[0], 2: sides[1]} if 1 in present else {present[1]: sides[0]}
            for i in present:
                st = side_of.get(i)
                if st is None:
                    continue
                st.wait_stream(main)  # inputs, parameters (optimizer step) and weight caches were produced on `main`
                with torch.cuda.stream(st):
                    outs[i] = l2_normalize(towers[i][0](towers[i][1]))
            for i in present:
                if i not in side_of:
                    outs[i] = l2_normalize(towers[i][0](towers[i][1]))
            for i, st in side_of.items():
                if i in present:
                    main.wait_stream(st)
                    outs[i].record_stream(main)
        dna_output, image_output, language_output = outs
        return image_output, dna_output, language_output, self.logit_scale.exp(), self.logit_bias

    # fp8-forward mode: which towers take it.  "pooled" (default) = the towers whose head AVERAGES its tokens — BarcodeBERT's
    # softmax-mean over 133 tokens (dna_encoder.py:131-137), BERT-small's mean over 20 positions (language_encoder.py:89): the e4m3
    # operand noise of a token row (~4 % per GEMM output) averages out in the embedding (measured 6-8e-3 on unit-norm rows) and the
    # contrastive gradient stays that of the bf16 step (cosine 0.990-0.9999 on trained weights).  The ViT reads ONE row, the class
    # token, whose embedding moves by 4-5e-2 under e4m3 — x14.3 in the logits — whatever the scale granularity (per tensor, per row,
    # MXFP8 per 32) and whichever rows / blocks are kept in bf16 (tools/fp8_policy_study.py, profiles/r04_exp_fp8_policy_study.log):
    # "all" adds it for an embedding-grade mode that is 6 % faster still and not gradient-faithful.
    # Round 5: "pooled_mlp" = "pooled" plus the MLP pair (fc1, fc2) of every ViT block — the oracle study of SITE selections
    # (profiles/r05_exp_fp8_vit_sites.log) found the class row's sensitivity to sit in the attention half (fp8 on QKV + projection alone:
    # cosine 0.954; on fc1 + fc2 alone 0.985; on everything 0.937, training batch).
    # Round 5, later: "pooled_ffn" = the MLP pair (fc1, fc2) of the mean-pooled towers ONLY, their attention half on bf16 — the oracle study of
    # site selections inside the DNA tower (profiles/r05_exp_fp8_dna_sites.log) puts the pooled towers' loss of gradient fidelity in the
    # attention half too (fresh batch: all four sites 0.9901, QKV + projection alone 0.9902, fc1 + fc2 alone 0.9987, fc2 alone 0.9991).
    FP8_TOWER_SETS = {"pooled": ("dna_encoder", "language_encoder"), "all": ("image_encoder", "dna_encoder", "language_encoder"),
                      "pooled_mlp": ("image_encoder", "dna_encoder", "language_encoder"), "pooled_ffn": ("dna_encoder", "language_encoder")}
    FP8_TOWER_SITES = {"pooled_mlp": {"image_encoder": ("fc1_in", "fc2_in")},    # towers of a set that take a site selection
                       "pooled_ffn": {"dna_encoder": ("fc1_in", "fc2_in"), "language_encoder": ("fc1_in", "fc2_in")}}

    def enable_fp8_forward(self, scales: Optional[dict] = None, enabled: bool = True, calibration_inputs=None, margin: float = 2.0,
                           towers=None):
        """fp8-forward mode (BASELINE.json configs[4]; not in the reference, which trains under bf16 autocast): the selected towers'
        forward GEMMs run on the fp8 MFMA, see TransformerStack.enable_fp8.  Needs frozen base weights (LoRA mode).
        towers: "pooled" (default: the mean-pooled towers, training-grade), "all" (adds the ViT: embedding-grade), or an iterable of
        encoder attribute names; None keeps the previous selection (what Trainer's periodic re-calibration passes).
        calibration_inputs = (image_input, dna_input, language_input): one bf16 no-grad forward over that batch measures
        max |activation| per layer and site and sets the per-layer power-of-two scales with `margin` headroom; without it
        the static FP8_SCALES are used."""
        if towers is None:
            towers = self.__dict__.get("_fp8_towers", "pooled")
        names = self.FP8_TOWER_SETS[towers] if isinstance(towers, str) else tuple(towers)
        for n in names:
            if n not in self.FP8_TOWER_SETS["all"]:
                raise ValueError(f"enable_fp8_forward: unknown tower {n!r}")
        self.__dict__["_fp8_towers"] = towers if isinstance(towers, str) else names
        every = [getattr(self, n).tower().stack for n in self.FP8_TOWER_SETS["all"]
                 if getattr(self, n) is not None and hasattr(getattr(self, n), "tower")]
        live = [n for n in names if getattr(self, n) is not None and hasattr(getattr(self, n), "tower")]
        stacks = [getattr(self, n).tower().stack for n in live]
        site_sel = self.FP8_TOWER_SITES.get(towers, {}) if isinstance(towers, str) else {}
        sites = [site_sel.get(n) for n in live]
        for st in every:
            st.disable_fp8()
        if not enabled:
            return self
        amax = [None] * len(stacks)
        if calibration_inputs is not None:
            for st in stacks:
                st.calibrate(True)
            with torch.no_grad():
                self(*calibration_inputs)
            self.join_streams()
            amax = [st.calibrate(False) for st in stacks]
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                # data-parallel replicas must quantise alike: take the maximum over the ranks' calibration batches
                flat = torch.tensor([v for am in amax for d in am for k, v in sorted(d.items())], dtype=torch.float32, device=self.logit_scale.device)
                torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.MAX)
                it = iter(flat.tolist())
                amax = [[{k: next(it) for k, _ in sorted(d.items())} for d in am] for am in amax]
        for st, am, ss in zip(stacks, amax, sites):
            st.enable_fp8(scales, amax=am if am else None, margin=margin, sites=ss)
        return self

    def enable_fp8_dgrad(self, towers="pooled", enabled: bool = True):
        """The 8-bit dgrad (engine numerics dgrad = "fp8") on a SELECTION of towers, the bf16 dgrad on the others:
        "pooled" = the mean-pooled towers (BarcodeBERT, BERT-small), where the oracle study and the MI355X measurement put its cost at
        <= 2e-4 of gradient cosine; "all" adds the ViT (about 1e-4 per block in the oracle study, more on MI355X's trained weights): the
        gradient stays within cosine 0.987-0.9998 of the bf16 dgrad's; together with the "pooled_ffn" fp8 FORWARD 0.978-0.9997 (fresh
        batches 0.978-0.988), with the all-site "pooled" forward 0.976-0.9996.  set_numerics(dgrad="fp8") is this with towers="all".
        An iterable of encoder attribute names selects towers explicitly."""
        names = self.FP8_TOWER_SETS[towers] if isinstance(towers, str) else tuple(towers)
        for n in names:
            if n not in self.FP8_TOWER_SETS["all"]:
                raise ValueError(f"enable_fp8_dgrad: unknown tower {n!r}")
        for n in self.FP8_TOWER_SETS["all"]:
            enc = getattr(self, n)
            if enc is not None and hasattr(enc, "tower"):
                enc.tower().stack.set_numerics(dgrad="fp8" if (enabled and n in names) else "bf16")
        return self

    def _stacks(self):
        return [enc.tower().stack for enc in (self.image_encoder, self.dna_encoder, self.language_encoder)
                if enc is not None and hasattr(enc, "tower")]

    def set_numerics(self, **settings):
        """Backward arithmetic switches of every tower (clibd_amd.engine.NUMERICS_CHOICES: residual_grad = "bf16" | "fp32",
        gelu_grad = "e4m7" | "bf16" | "u8", attn_bwd = "2phase" | "sp", ln_fold, dgrad).  The CLIBD_* environment variables only give the defaults a
        tower is constructed with; this is the explicit form (per model, recorded by bench.py and save_training_state)."""
        for st in self._stacks():
            st.set_numerics(**settings)
        return self

    def numerics(self) -> dict:
        """{tower: settings}: what produced this model's gradients (written into the bench line and the training state)."""
        out = {}
        for name in ("image_encoder", "dna_encoder", "language_encoder"):
            enc = getattr(self, name)
            if enc is not None and hasattr(enc, "tower"):
                st = enc.tower().stack
                out[name] = dict(st.numerics, forward="fp8 (e4m3) GEMM operands" if st.fp8 is not None else "bf16")
        return out

    def join_streams(self):
        """Make the current stream wait for the towers' side streams (call after backward(): gradients written into the
        fused optimizer's flat buffers by a tower's backward are produced on that tower's stream)."""
        for dev, sides in self.__dict__.get("_streams", {}).items():
            main = torch.cuda.current_stream(dev)
            for st in sides:
                main.wait_stream(st)


def _get(cfg, name, default=None):
    return getattr(cfg, name) if hasattr(cfg, name) else default


def load_clip_model(args, device=None):
    """Factory with the reference's flag semantics (`args.model_config.*`, simple_clip.py:100-246), including its quirks:
    `using_open_clip` overwrites `disable_lora` (simple_clip.py:114-116); the image tower treats `lora_layer=[]` as "all
    layers" while the BERT towers treat it as "none" (SURVEY §3.4).  Pretrained weights come from LOCAL checkpoints only
    (`args.bioscan_bert_checkpoint`, `args.model_config.image.image_encoder_trained_with_simclr_style_ckpt_path`,
    `args.model_config.image.vit_checkpoint`); absent ones mean random initialisation."""
    mc = args.model_config
    disable_lora = bool(_get(mc, "disable_lora", False))
    if hasattr(mc, "using_open_clip"):
        disable_lora = bool(mc.using_open_clip)
    image_cfg, lang_cfg, dna_cfg = _get(mc, "image"), _get(mc, "language"), _get(mc, "dna")
    image_model = _get(image_cfg, "model") if image_cfg is not None else None
    language_model = _get(lang_cfg, "model") if lang_cfg is not None else None
    if _get(mc, "for_bio_clip", False) or (image_model == "lora_clip_image" and language_model == "lora_clip_text"):
        raise NotImplementedError("open_clip / BioCLIP towers are not part of the MI355X hot path (SURVEY §2, row 3)")
    out_dim = mc.output_dim

    image_encoder = dna_encoder = language_encoder = None
    if image_cfg is not None:
        if _get(image_cfg, "input_type", "image") != "image":
            raise NotImplementedError("pre-extracted feature (MLP) towers are out of scope (SURVEY §2 row 8)")
        vit = create_vit(_get(image_cfg, "pre_train_model", "vit_base_patch16_224"))
        # reference key (simple_clip.py:154-165): a SimCLR-style ViT checkpoint {"state_dict": ...}, DDP prefix stripped, loaded
        # STRICTLY into the timm-shaped body; `vit_checkpoint` is this package's alias for a plain local timm state dict
        # (the reference downloads `pretrained=True` weights instead, which needs the network)
        ck = _get(image_cfg, "image_encoder_trained_with_simclr_style_ckpt_path")
        if ck:
            sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
            vit.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()})
        elif _get(image_cfg, "vit_checkpoint"):
            sd = torch.load(_get(image_cfg, "vit_checkpoint"), map_location="cpu", weights_only=False)
            sd = sd.get("state_dict", sd)
            vit.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}, strict=False)
        image_encoder = CLIBDImageEncoder(vit_model=vit, r=4, num_classes=out_dim, lora_layer=[] if disable_lora else None)
    if lang_cfg is not None:
        if _get(lang_cfg, "input_type", "sequence") != "sequence":
            raise TypeError(f"Using {lang_cfg.input_type} as language input is not support yet.")
        _, bert = load_pre_trained_bert(_get(lang_cfg, "pre_train_model", "prajjwal1/bert-small"))
        language_encoder = CLIBDLanguageEncoder(model=bert, r=4, num_classes=out_dim, lora_layer=[] if disable_lora else None)
    if dna_cfg is not None:
        if _get(dna_cfg, "input_type", "sequence") != "sequence":
            raise NotImplementedError("pre-extracted feature (MLP) towers are out of scope (SURVEY §2 row 8)")
        ckpt = _get(args, "bioscan_bert_checkpoint")
        pre = _get(mc, "pre_train_for_barcode_bert")
        if pre == "BIOSCAN-5M":
            ckpt = _get(args, "bioscan_bert_checkpoint_trained_with_bioscan_5_m", ckpt)
        elif pre == "CANADA-1-5M":
            ckpt = _get(args, "bioscan_bert_checkpoint_trained_with_canada_1_5_m", ckpt)
        import os

        bert = load_pre_trained_bioscan_bert(ckpt if (ckpt and os.path.exists(str(ckpt))) else None)
        dna_encoder = CLIBDDNAEncoder(model=bert, r=4, num_classes=out_dim, lora_layer=[] if disable_lora else None)

    model = SimpleCLIP(image_encoder=image_encoder, dna_encoder=dna_encoder, language_encoder=language_encoder)
    if device is not None:
        model.to(device)
    if disable_lora:
        for p in model.parameters():
            p.requires_grad = True  # full fine-tune (simple_clip.py:235-237): the towers switch to their weight-gradient paths
    for cfg, enc in ((image_cfg, model.image_encoder), (dna_cfg, model.dna_encoder), (lang_cfg, model.language_encoder)):
        if cfg is not None and _get(cfg, "freeze", False) and enc is not None:
            for p in enc.parameters():
                p.requires_grad = False
    return model


def initialize_model_and_load_from_checkpoint(args, device=None):
    """reference simple_clip.py:248-285 (local checkpoints only); see clibd_amd.checkpoint."""
    from ..checkpoint import initialize_model_and_load_from_checkpoint as _impl

    return _impl(args, device)
