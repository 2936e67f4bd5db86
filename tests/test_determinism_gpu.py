"""Run-to-run bit reproducibility of the LoRA training step (round 6).

Rounds 1-5 accumulated the loss value, d(logit_scale), the trainable heads' bias gradients and — at batches below 8192 token rows — the
adapters' gradients with float atomics: two runs of the same step agreed to ~1e-7 only, AdamW turned that into different trajectories, and
every "gradient fidelity after 40 training steps" figure of the fp8 modes (tests/test_fp8_gpu.py) scattered by +-0.005 from run to run.
Every sum on the frozen-base path now has a fixed order (csrc/loss.hip reduce_rows_add_kernel, csrc/gemm.hip colsum_partials_kernel,
csrc/lora.hip lora_reduce_kernel at every whole-slab M), so: same parameters + same batch + same dropout seed -> the same bits.

(Full fine-tune mode keeps float atomics in the LayerNorm / embedding parameter gradients: csrc/layernorm.hip PG, csrc/paramgrad.hip.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(dev, seed=11):
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    torch.manual_seed(seed)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    return model.to(dev)


@pytest.mark.parametrize("train_mode", [False, True], ids=["eval", "train"])
def test_training_trajectory_repeats_bit_for_bit(dev, train_mode):
    """Two fresh models from the same seed, six AdamW steps each on the same 32 pairs (ViT-B/16: 6 304 token rows, BarcodeBERT: 4 256 —
    the sizes of the fp8 fidelity study), train mode drawing the same dropout seeds: every loss value and every trainable parameter
    (adapters, heads, logit_scale) must be IDENTICAL."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.train import Trainer

    B = 32
    batch = synthetic_batch(B, dev, seed=3, rank=0, with_text=False)

    def run():
        model = _pair(dev)
        model.train(train_mode)
        tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
        torch.manual_seed(123)   # the towers draw their dropout base seeds from the CPU generator
        losses = [tr.step(batch["image"], batch["dna"], None, batch["labels"]).clone() for _ in range(6)]
        torch.cuda.synchronize()
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        return torch.stack(losses).cpu(), {n: p.detach().clone().cpu() for n, p in model.named_parameters() if n in names}

    l1, p1 = run()
    l2, p2 = run()
    assert torch.isfinite(l1).all() and l1[-1] < l1[0], l1
    bad = [n for n in p1 if not torch.equal(p1[n], p2[n])]
    assert torch.equal(l1, l2), (l1, l2)
    assert not bad, f"{len(bad)} of {len(p1)} trainable tensors differ between two identical runs: {bad[:8]}"


def test_single_step_gradients_repeat_bit_for_bit(dev):
    """One forward + backward twice on the same model (no optimizer in between): loss and every gradient identical, names of the
    offenders reported otherwise."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss

    model = _pair(dev).eval()
    B = 32
    batch = synthetic_batch(B, dev, seed=4, rank=0, with_text=False)
    labels = (torch.arange(B) % 13).to(dev)   # duplicate labels: the multi-hot rows of the soft-target loss
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    ps = {n: p for n, p in model.named_parameters() if p.requires_grad}

    def run():
        hi, hd, _, scale, _ = model(batch["image"], batch["dna"], None)
        loss = crit(hi, hd, None, labels, scale)
        gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
        model.join_streams()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: (torch.zeros_like(p) if g is None else g.detach().clone()) for (n, p), g in zip(ps.items(), gs)}

    l1, g1 = run()
    l2, g2 = run()
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert torch.equal(l1, l2), (float(l1), float(l2))
    assert not bad, f"{len(bad)} of {len(g1)} gradients differ between two identical backward passes: {bad[:8]}"
    assert float(g1["logit_scale"].abs()) > 0
