"""Round 5: the full-size model (ViT-B/16 + BERT-base) at batch 64 — the smallest batch at which numerics ln_fold="on" engages — with
the fold off and on, against the oracle's bf16 mode (standard and with the same fold) and its fp32 mode: north_star's 1e-3 numbers.
    python tools/ln_fold_full_size.py"""
import sys, time, torch
sys.path.insert(0, ".")
from oracle import clibd_oracle as O
from clibd_amd.data import synthetic_batch
from clibd_amd.model import ClipLoss, CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert
dev = torch.device("cuda:0")
torch.manual_seed(13)
B = 64
om = O.build_image_dna_model()
with torch.no_grad():
    for n, p in om.named_parameters():
        if "linear_b_" in n or ".w_b." in n:
            p.normal_(0, 0.02)
model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                   CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
model.load_state_dict(om.state_dict(), strict=True)
model = model.to(dev).eval()
batch = synthetic_batch(B, torch.device("cpu"), seed=5, rank=0, with_text=False)
labels = torch.arange(B) % 23
crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
t0 = time.time()
with torch.no_grad():
    ref = {}
    for name, ctx in (("bf16", (O.precision("bf16"),)), ("bf16+fold", (O.precision("bf16"), O.ln_fold(True))), ("fp32", ())):
        for c in ctx: c.__enter__()
        oi, od, _, osc, _ = om(batch["image"], batch["dna"], None)
        ref[name] = (oi, od, float(O.contrastive_loss([oi, od, None], labels, osc)))
        for c in reversed(ctx): c.__exit__(None, None, None)
    print(f"oracle: {time.time() - t0:.0f} s; image embedding bf16 vs fp32 {(ref['bf16'][0] - ref['fp32'][0]).abs().max():.2e}, bf16+fold vs fp32 {(ref['bf16+fold'][0] - ref['fp32'][0]).abs().max():.2e}, "
          f"bf16+fold vs bf16 {(ref['bf16+fold'][0] - ref['bf16'][0]).abs().max():.2e}; loss bf16 {ref['bf16'][2]:.6f} fold {ref['bf16+fold'][2]:.6f} fp32 {ref['fp32'][2]:.6f}", flush=True)
    for mode in ("off", "on"):
        model.set_numerics(ln_fold=mode)
        hi, hd, _, scale, _ = model(batch["image"].to(dev), batch["dna"].to(dev), None)
        loss = float(crit(hi, hd, None, labels.to(dev), scale))
        hi = hi.cpu()
        print(f"HIP ln_fold={mode}: image embedding (unit-norm rows, max abs) vs oracle bf16 {(hi - ref['bf16'][0]).abs().max():.2e}  vs oracle bf16+fold {(hi - ref['bf16+fold'][0]).abs().max():.2e}  "
              f"vs oracle fp32 {(hi - ref['fp32'][0]).abs().max():.2e};  |loss - bf16| {abs(loss - ref['bf16'][2]):.2e}  |loss - bf16+fold| {abs(loss - ref['bf16+fold'][2]):.2e}  |loss - fp32| {abs(loss - ref['fp32'][2]):.2e}", flush=True)
