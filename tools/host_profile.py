"""Where the host's time goes while it enqueues one training step (cProfile, per-GPU batch 256 = the 8-GPU per-rank shape).
usage: python tools/host_profile.py [batch] > gpurun_out/<tag>/host_profile.txt"""
import cProfile, io, os, pstats, sys, time
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
# un-profiled: wall and CPU time per step, synchronising after every step so that back-pressure never enters
w, c = [], []
for _ in range(10):
    t0, c0 = time.perf_counter(), time.thread_time()
    tr.step(batch["image"], batch["dna"], None, batch["labels"])
    w.append(time.perf_counter() - t0); c.append(time.thread_time() - c0)
    torch.cuda.synchronize()
print(f"b={b}: host enqueue per step (queue empty at the start of every step): wall median {sorted(w)[5] * 1e3:.2f} ms, cpu median {sorted(c)[5] * 1e3:.2f} ms")
pr = cProfile.Profile()
n = 5
for _ in range(n):
    pr.enable()
    tr.step(batch["image"], batch["dna"], None, batch["labels"])
    pr.disable()
    torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats(key).print_stats(28)
    print(f"==== {n} steps, sorted by {key}")
    print(sio.getvalue())
