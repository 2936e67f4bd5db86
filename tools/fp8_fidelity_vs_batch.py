"""Gradient cosine of every fp8 mode against the bf16 step as a function of the batch the gradient is taken on (round 6).
Protocol of tests/test_fp8_gpu.py::test_fp8_gradient_fidelity_against_the_batch_size: random-init full-size towers, adapters / heads trained
40 bf16 AdamW steps on 32 synthetic pairs, then unseen batches of B pairs; same dropout masks in every mode.   python tools/fp8_fidelity_vs_batch.py [B ...]"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_fp8_gpu import _cosv, _full_size_pair, _named_grads  # noqa: E402
from clibd_amd.data import synthetic_batch  # noqa: E402
from clibd_amd.model import ClipLoss  # noqa: E402
from clibd_amd.train import Trainer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sizes = [int(a) for a in sys.argv[1:]] or [32, 128, 512, 1024]
    model = _full_size_pair(dev)
    batch = synthetic_batch(32, dev, seed=3, rank=0, with_text=False)
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    losses = [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(40)]
    print(f"loss {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
    for tw in (model.image_encoder.tower(), model.dna_encoder.tower()):
        tw.grad_sink = None

    def grads(bt):
        torch.manual_seed(77)
        hi, hd, _, scale, _ = model(bt["image"], bt["dna"], None)
        g = _named_grads(model, crit(hi, hd, None, bt["labels"], scale))
        model.join_streams()
        torch.cuda.synchronize()
        return torch.cat([g[n].flatten().double() for n in sorted(g)])

    MODES = ((None, "pooled"), (None, "all"), ("pooled_ffn", None), ("pooled_ffn", "pooled"), ("pooled_ffn", "all"), ("pooled", None), ("pooled", "all"),
             ("pooled_mlp", None), ("pooled_mlp", "all"), ("all", None), ("all", "all"))
    for B in sizes:
        bt = synthetic_batch(B, dev, seed=11, rank=0, with_text=False)
        model.enable_fp8_forward(enabled=False)
        model.enable_fp8_dgrad(enabled=False)
        g16 = grads(bt)
        for fwd, dg in MODES:
            if fwd:
                model.enable_fp8_forward(calibration_inputs=(bt["image"], bt["dna"], None), towers=fwd)
            else:
                model.enable_fp8_forward(enabled=False)
            if dg:
                model.enable_fp8_dgrad(towers=dg)
            c = _cosv(grads(bt), g16)
            model.enable_fp8_dgrad(enabled=False)
            print(f"unseen batch of {B:5d} pairs  forward {str(fwd):10s} dgrad8 {str(dg):6s}  cosine(fp8, bf16) {c:.4f}", flush=True)
    model.enable_fp8_forward(enabled=False)


if __name__ == "__main__":
    main()
