"""Round-2 golden vectors, again by IMPORTING THE REFERENCE in the authoring container (data only is written):

    python tests/golden/make_golden_r2.py         # needs /root/reference and the round-1 goldens; writes two files

  autocast_golden.pt  the reference's tiny towers and full step (same weights and inputs as dna/text/image/step_tiny_golden.pt)
                      run under torch.autocast('cpu', dtype=torch.bfloat16) — the reference's optional bf16 mode
                      (epoch/train_epoch.py:42-46, `enable_autocast`); the loss stays outside autocast as in :52-57.
                      Anchors the HIP path's bf16 numerics to the reference itself: the HIP outputs must be at least as close
                      to the reference's fp32 outputs as the reference's own bf16 mode is.
  loss_w2_golden.pt   the reference's ClipLoss (model/loss_func.py:110-201) run on TWO gloo ranks (rows split across ranks):
                      per-rank loss, local feature gradients, logit-scale gradient (SURVEY §8c G1, W = 2).
"""
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (stubs + reference import helpers)


def _load(name):
    return torch.load(os.path.join(HERE, name), map_location="cpu", weights_only=False)


def _w2_worker(rank, world, port, cases, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref = mg.import_reference()
        lf = ref["loss_func"]
        res = []
        for c in cases:
            N = c["labels"].numel()
            b = N // world
            sl = slice(rank * b, (rank + 1) * b)
            feats = [None if f is None else f[sl].clone().requires_grad_(True) for f in c["features"]]
            ls = c["log_scale"].clone().requires_grad_(True)
            crit = lf.ClipLoss(local_loss=False, gather_with_grad=c["gather_with_grad"], rank=rank, world_size=world, criterion=nn.CrossEntropyLoss(),
                               bind_to=c["bind_to"], no_image_text_loss=c["no_image_text_loss"])
            loss = crit(feats[0], feats[1], feats[2], c["labels"][sl], ls.exp())
            present = [f for f in feats if f is not None]
            gs = torch.autograd.grad(loss, present + [ls], allow_unused=True)
            res.append({"loss": loss.detach().clone(), "grads": [torch.zeros_like(p) if g is None else g.clone() for p, g in zip(present + [ls], gs)]})
        out[rank] = res
    finally:
        dist.destroy_process_group()


def world2_loss_goldens():
    import torch.multiprocessing as mp

    g = torch.Generator().manual_seed(20)
    cases = []
    for N, D, nmod, dup, bind_to, no_it, gwg in [(16, 64, 2, False, None, False, True), (32, 128, 2, True, None, False, True),
                                                  (24, 64, 3, True, None, False, True), (16, 64, 3, False, "dna", False, True),
                                                  (16, 64, 3, True, None, True, True), (16, 64, 2, True, None, False, False)]:
        feats = [torch.randn(N, D, generator=g) for _ in range(nmod)] + [None] * (3 - nmod)
        labels = torch.arange(N) // 2 if dup else torch.arange(N)
        if dup:
            labels = labels[torch.randperm(N, generator=g)]   # duplicates across the two ranks' blocks
        cases.append({"features": feats, "labels": labels, "log_scale": torch.tensor(2.6592600), "bind_to": bind_to, "no_image_text_loss": no_it,
                      "gather_with_grad": gwg})
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_w2_worker, args=(2, 29641, cases, out), nprocs=2, join=True)
    for i, c in enumerate(cases):
        c["per_rank"] = [out[r][i] for r in range(2)]
    torch.save({"world_size": 2, "cases": cases}, os.path.join(HERE, "loss_w2_golden.pt"))
    print("[make_golden_r2] W=2 losses:", [(float(c["per_rank"][0]["loss"]), float(c["per_rank"][1]["loss"])) for c in cases])


def autocast_goldens():
    from transformers import BertConfig, BertForMaskedLM, BertModel

    from oracle import clibd_oracle as O

    ref = mg.import_reference()
    lf = ref["loss_func"]
    gd, gt, gi, gs = _load("dna_tiny_golden.pt"), _load("text_tiny_golden.pt"), _load("image_tiny_golden.pt"), _load("step_tiny_golden.pt")
    enc = ref["dna_encoder"].CLIBDDNAEncoder(model=BertForMaskedLM(BertConfig(vocab_size=1027, output_hidden_states=True, **gd["config"])), r=4, num_classes=128)
    enc.load_state_dict(gd["state_dict"], strict=True)
    tenc = ref["language_encoder"].CLIBDLanguageEncoder(model=BertModel(BertConfig(vocab_size=gt["vocab"], **gt["config"])), r=4, num_classes=128)
    tenc.load_state_dict(gt["state_dict"], strict=True)
    c = gi["config"]
    ienc = ref["image_encoder"].CLIBDImageEncoder(vit_model=O.VisionTransformer(img_size=224, patch=16, dim=c["dim"], depth=c["depth"], heads=c["heads"], num_classes=10),
                                                  r=4, num_classes=128)
    ienc.load_state_dict(gi["state_dict"], strict=True)
    for m in (enc, tenc, ienc):
        m.eval()
    out = {"torch": torch.__version__, "mode": "torch.autocast('cpu', dtype=torch.bfloat16) around the model forward; loss in fp32 outside"}
    for name, m, x, g in (("dna", enc, gd["ids"], gd), ("text", tenc, gt["inputs"], gt), ("image", ienc, gi["image_u8"].float() / 255.0, gi)):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = m(x)
        y32 = y.float()
        grads = mg.grads_of(m, (y32 * g["cot"]).sum())
        out[name] = {"out": y32.detach().clone(), "out_dtype": str(y.dtype), "grads": grads,
                     "err_vs_fp32": float((y32.detach() - g["out"]).abs().max())}
        print(f"[make_golden_r2] {name}: autocast out dtype {y.dtype}, max |bf16 - fp32| = {out[name]['err_vs_fp32']:.3e}")
    model = ref["simple_clip"].SimpleCLIP(image_encoder=ienc, dna_encoder=enc, language_encoder=tenc)
    with torch.no_grad():
        model.logit_scale.copy_(gs["logit_scale"])
    model.eval()
    img = gs["image_u8"].float() / 255.0
    for tag, use_text in (("id", False), ("idt", True)):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            io, do_, to, scale, _ = model(img, gs["dna"], gs["text"])
        crit = lf.ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=nn.CrossEntropyLoss())
        loss = crit(io.float(), do_.float(), to.float() if use_text else None, gs["labels"], scale.float())
        out[f"loss_{tag}"] = loss.detach().clone()
        out[f"features_{tag}"] = [io.float().detach().clone(), do_.float().detach().clone(), to.float().detach().clone()]
        print(f"[make_golden_r2] step {tag}: autocast loss {float(loss):.6f} (fp32 reference {float(gs[f'loss_{tag}']):.6f})")
    torch.save(out, os.path.join(HERE, "autocast_golden.pt"))


if __name__ == "__main__":
    world2_loss_goldens()
    import torch.distributed as dist

    if not dist.is_initialized():   # the reference's ClipLoss all-gathers the labels even at world_size 1
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29642"
        dist.init_process_group("gloo", rank=0, world_size=1)
    autocast_goldens()
    for f in ("autocast_golden.pt", "loss_w2_golden.pt"):
        print(f"  {f}: {os.path.getsize(os.path.join(HERE, f)) / 1e6:.2f} MB")
