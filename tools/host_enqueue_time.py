"""Host time to enqueue one training step (no sync) against the device time of the step, per-GPU batch 256 (the N = 8 point of the
scaling run): how far the CPU runs ahead of the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert
from clibd_amd.train import Trainer
dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                   CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None).to(dev)
tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
batch = synthetic_batch(b, dev, seed=42, rank=0, with_text=False)
for _ in range(5):
    tr.step(batch["image"], batch["dna"], None, batch["labels"])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
host = []
for _ in range(n):
    h0 = time.perf_counter()
    tr.step(batch["image"], batch["dna"], None, batch["labels"])
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"b={b}: host enqueue {t_enq / n * 1e3:.2f} ms/step (median {sorted(host)[n // 2] * 1e3:.2f}), wall {t_all / n * 1e3:.2f} ms/step")
