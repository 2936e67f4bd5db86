"""Run-to-run determinism of the full-size two-tower forward + backward (towers on two streams), optionally alternating with an
fp8-forward pass (what tests/test_fp8_gpu.py::test_full_size_fp8_forward_close_to_bf16_path does between its two bf16 runs).
python tools/stress_model_determinism.py [iters] [alternate_fp8] [with_backward]"""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_fp8_gpu import _full_size_pair
from clibd_amd.data import synthetic_batch
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
alt = len(sys.argv) > 2 and sys.argv[2] == "1"
bwd = len(sys.argv) > 3 and sys.argv[3] == "1"
model = _full_size_pair(dev)
batch = synthetic_batch(16, dev, seed=5, rank=0, with_text=False)
from clibd_amd.model import ClipLoss
labels = (torch.arange(16) % 11).to(dev)
crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())


def fwd():
    hi, hd, _, scale, _ = model(batch["image"], batch["dna"], None)   # grad mode on, as in the test (activations are kept)
    if bwd:
        loss = crit(hi, hd, None, labels, scale)
        ps = [p for p in model.parameters() if p.requires_grad]
        torch.autograd.grad(loss, ps, allow_unused=True)
    model.join_streams(); torch.cuda.synchronize()
    return hi.detach().float().cpu(), hd.detach().float().cpu()


ri, rd = fwd()
bad_i = bad_d = 0
for k in range(iters):
    if alt:
        model.enable_fp8_forward(towers="all"); fwd(); model.enable_fp8_forward(enabled=False)
    i, d = fwd()
    if not torch.equal(i, ri):
        bad_i += 1
        rows = (i != ri).any(dim=1).nonzero().flatten().tolist()
        print(f"  iter {k}: image rows {rows} differ, max abs {float((i - ri).abs().max()):.3e}", flush=True)
    if not torch.equal(d, rd):
        bad_d += 1
        rows = (d != rd).any(dim=1).nonzero().flatten().tolist()
        print(f"  iter {k}: dna rows {rows} differ, max abs {float((d - rd).abs().max()):.3e}", flush=True)
print(f"alternate_fp8={int(alt)} backward={int(bwd)}: image mismatches {bad_i}/{iters}, dna mismatches {bad_d}/{iters}")
