"""Diagnostic: batch-split invariance with / without tower stream overlap, no per-op host syncs."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

dev = torch.device("cuda:0")
NB, CH = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256
torch.manual_seed(2048)
model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                   CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
with torch.no_grad():
    for enc in (model.image_encoder, model.dna_encoder):
        for wb in enc.w_Bs:
            wb.weight.normal_(0, 0.02)
model = model.to(dev).eval()
batch = synthetic_batch(NB, dev, seed=42, rank=0, with_text=False)


def fwd(sl, sync=False):
    with torch.no_grad():
        i, d, _, _, _ = model(batch["image"][sl], batch["dna"][sl], None)
    if sync:
        torch.cuda.synchronize()
    return i, d


ref = {}
for overlap in (False, True):
    for sync in (True, False):
        model.overlap_towers = overlap
        i_full, d_full = fwd(slice(0, NB), sync)
        ch = [fwd(slice(s, s + CH), sync) for s in range(0, NB, CH)]
        torch.cuda.synchronize()
        i_ch, d_ch = torch.cat([c[0] for c in ch]), torch.cat([c[1] for c in ch])
        per_i = [(i_full[s:s + CH] - i_ch[s:s + CH]).abs().max().item() for s in range(0, NB, CH)]
        per_d = [(d_full[s:s + CH] - d_ch[s:s + CH]).abs().max().item() for s in range(0, NB, CH)]
        print(f"overlap={overlap} sync_between_calls={sync}: image per-chunk max diff {['%.1e' % v for v in per_i]}")
        print(f"                                         dna   per-chunk max diff {['%.1e' % v for v in per_d]}")
        key = (overlap, sync)
        ref[key] = (i_full.clone(), d_full.clone(), i_ch.clone(), d_ch.clone())
base = ref[(False, True)]
for key, v in ref.items():
    print(key, "vs serial+sync: full image", (v[0] - base[0]).abs().max().item(), "full dna", (v[1] - base[1]).abs().max().item(),
          "chunk image", (v[2] - base[2]).abs().max().item(), "chunk dna", (v[3] - base[3]).abs().max().item())
