#!/usr/bin/env python3
"""Headline benchmark: CLIBD Image+DNA contrastive TRAINING step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: the script starts torch.distributed.run itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward of the ViT-B/16 + BarcodeBERT(BERT-base) towers with rank-4 LoRA adapters, L2-normalise,
packed RCCL all-gather of the embeddings, soft-target InfoNCE over the global batch, backward (dgrad through the frozen
bases, adapter + head gradients), flat-bucket gradient all-reduce and fused AdamW — on synthetic 224x224 images and
660-nt barcodes (133 tokens), random-init weights of the reference's shapes.  The GLOBAL batch is held at 2048
(BASELINE.json `metric`: "global batch 2048, 1/2/4/8 GPU"): per-GPU batch = 2048 / N, so N=1 runs b=2048 on one GPU
and N=8 is BASELINE configs[2] (b=256 per GPU) — strong scaling.  `--per-gpu-batch 256` gives BASELINE configs[1].

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 MFMA GEMM): algorithmic FLOPs of every
GEMM launch in the timed region / their summed HIP-event durations on the launch stream, against the 2.5 PFLOP/s
dense bf16 peak.  `cpu_baseline` is the CPU oracle's training step (oracle/clibd_oracle.py, the reference's CPU path
restated and pinned to it) timed on this host's cores with a bounded sample, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time

# More hardware queues than ROCm's default of 4 per priority level, before the HIP runtime starts: with a process group the process
# holds RCCL's streams beside the towers', and streams that share a queue serialize.  The towers' side streams are
# also high-priority (their own queue pool), so this is the second line of defence.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver (also under torchrun)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
MAX_SCLK_MHZ = 2400.0         # the shader clock that figure is quoted at
GF_PER_PAIR_TRAIN = 117.6     # LoRA training step, I+D pair (fwd 58.78 GF + dgrad-only bwd) — the reference model's FLOPs
REFERENCE_NUMERICS = dict(residual_grad="fp32", gelu_grad="bf16", attn_bwd="2phase", ln_fold="off", dgrad="bf16")   # the reference's backward semantics (engine.NUMERICS_CHOICES)
GF_PER_PAIR_FULLFT = 176.3    # full fine-tune (dgrad + wgrad), the authors' final configuration (`disable_lora: true`)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=2048, help="held constant over N (BASELINE.json metric)")
    ap.add_argument("--per-gpu-batch", type=int, default=None, help="override: fixed per-GPU batch (256 = BASELINE configs[1]); weak scaling")
    ap.add_argument("--tri-modal", action="store_true", help="add the BERT-small text tower (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="BASELINE configs[0]: batch 32")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU steps (median reported) after one warm-up")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-batch (PCIe-inclusive) side measurement")
    ap.add_argument("--no-gemm-timing", action="store_true")
    ap.add_argument("--no-ref-numerics", action="store_true", help="skip the side measurement at the reference's backward numerics")
    ap.add_argument("--gemm-breakdown", action="store_true", help="per-shape GEMM time table on stderr")
    ap.add_argument("--fp8-forward", nargs="?", const="pooled", default=None, choices=["pooled", "pooled_ffn", "pooled_mlp", "all"],
                    help="BASELINE configs[4] (not the headline config): forward GEMMs on the fp8 MFMA.  'pooled' (default) = the towers whose head "
                         "averages its tokens (BarcodeBERT, BERT-small): gradient-faithful; 'all' adds the ViT: embedding-grade")
    ap.add_argument("--dgrad", choices=["bf16", "fp8", "fp8-pooled"], default=None,
                    help="numerics switch dgrad (BASELINE configs[4]): fp8 = the MLP / projection activation-gradient GEMMs of every tower on e4m3 operands with per-row scales; "
                         "fp8-pooled = of the mean-pooled towers only (BarcodeBERT, BERT-small).  Not the headline config")
    ap.add_argument("--no-configs4", action="store_true", help="skip the BASELINE configs[4] side record (per-GPU batch 1024: bf16 vs three fp8 modes — pooled_ffn forward + fp8-pooled dgrad, 8-bit dgrad on both towers, both — with in-run gradient cosines)")
    ap.add_argument("--configs4-batch", type=int, default=1024, help="per-GPU batch of the configs[4] side record (BIOSCAN-5M-shaped run: 1024)")
    ap.add_argument("--full-finetune", action="store_true", help="model_config.disable_lora: every encoder weight trainable (not the headline config)")
    ap.add_argument("--eval", action="store_true", help="eval path (SURVEY §8f-2, not the headline metric): no-grad embedding forward + K10 top-k search")
    ap.add_argument("--eval-queries", type=int, default=1024, help="--eval: queries per top-k launch")
    ap.add_argument("--eval-keys", type=int, default=409600, help="--eval: size of the key bank (BIOSCAN-5M-sized by default; 21000 = BIOSCAN-1M)")
    return ap.parse_args()


class GemmTimer:
    """HIP-event brackets around every GEMM launch (same stream as the launch) + algorithmic FLOPs."""

    def __init__(self):
        self.events, self.flops, self.enabled, self.shapes, self.bytes = [], 0.0, False, [], 0.0
        self.flops_fp8 = 0.0

    def install(self):
        from clibd_amd import ops

        inner = ops.gemm_nt
        timer = self

        def timed(a, w, **kw):
            if not timer.enabled:
                return inner(a, w, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner(a, w, **kw)
            e1.record()
            timer.events.append((e0, e1))
            timer.flops += 2.0 * a.shape[0] * w.shape[0] * a.shape[1]
            per_out = sum(b_ for k_, b_ in (("out_bf16", 2), ("out_pre", 2), ("out_f32", 4), ("residual", 4), ("aux", 2)) if kw.get(k_) is not None)
            timer.bytes += 2.0 * a.shape[1] * (a.shape[0] + w.shape[0]) + float(per_out) * a.shape[0] * w.shape[0]
            timer.shapes.append((a.shape[0], w.shape[0], a.shape[1], "+".join(sorted(k for k, v in kw.items() if v is not None))))

        inner8 = ops.gemm_fp8_nt

        def timed8(a, w, cs, **kw):
            if not timer.enabled:
                return inner8(a, w, cs, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner8(a, w, cs, **kw)
            e1.record()
            timer.events.append((e0, e1))
            fl = 2.0 * a.shape[0] * w.shape[0] * a.shape[1]
            timer.flops += fl
            timer.flops_fp8 += fl
            per_out = sum(b_ for k_, b_ in (("out_bf16", 2), ("out_pre", 2), ("out_f32", 4), ("residual", 4), ("gelu_out_fp8", 1)) if kw.get(k_) is not None)
            timer.bytes += 1.0 * a.shape[1] * (a.shape[0] + w.shape[0]) + float(per_out) * a.shape[0] * w.shape[0]
            timer.shapes.append((a.shape[0], w.shape[0], a.shape[1], "fp8+" + "+".join(sorted(k for k, v in kw.items() if v is not None))))

        inner8d = ops.gemm_fp8_dgrad_nt

        def timed8d(a, w, cs, **kw):   # the 8-bit dgrad forms (numerics dgrad = fp8): priced at the fp8 peak like the fp8 forward
            if not timer.enabled:
                return inner8d(a, w, cs, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner8d(a, w, cs, **kw)
            e1.record()
            timer.events.append((e0, e1))
            fl = 2.0 * a.shape[0] * w.shape[0] * a.shape[1]
            timer.flops += fl
            timer.flops_fp8 += fl
            per_out = sum(b_ for k_, b_ in (("out_bf16", 2), ("out_fp8", 1), ("aux", 2)) if kw.get(k_) is not None)
            timer.bytes += 1.0 * a.shape[1] * (a.shape[0] + w.shape[0]) + float(per_out) * a.shape[0] * w.shape[0] + (4.0 * a.shape[0] if kw.get("a_row_dequant") is not None else 0.0)
            timer.shapes.append((a.shape[0], w.shape[0], a.shape[1], "fp8dgrad+" + "+".join(sorted(k for k, v in kw.items() if v is not None and k not in ("act", "out_fp8_scale")))))

        ops.gemm_nt = timed
        ops.gemm_fp8_nt = timed8
        ops.gemm_fp8_dgrad_nt = timed8d
        import clibd_amd.engine as eng
        import clibd_amd.towers as tw

        assert eng.ops is ops and tw.ops is ops

    def result(self):
        if not self.events:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self.events)
        # time the same launches would take at the dense MFMA peaks (fp8 launches priced at the fp8 peak = 2 x bf16)
        ideal_ms = ((self.flops - self.flops_fp8) / (PEAK_BF16_TFLOPS * 1e12) + self.flops_fp8 / (2 * PEAK_BF16_TFLOPS * 1e12)) * 1e3
        return {"launches": len(self.events), "total_ms": ms, "tflops": self.flops / (ms * 1e-3) / 1e12,
                "bytes_per_launch": self.bytes / len(self.events), "frac": ideal_ms / ms, "fp8_flop_share": self.flops_fp8 / max(self.flops, 1.0)}

    def breakdown(self, steps):
        agg = {}
        for (e0, e1), key in zip(self.events, self.shapes):
            n, t = agg.get(key, (0, 0.0))
            agg[key] = (n + 1, t + e0.elapsed_time(e1))
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        for (M, N, K, kinds), (n, t) in rows:
            print(f"[gemm] M={M:6d} N={N:5d} K={K:5d} x{n // steps:3d}/step {t / steps:7.3f} ms/step {t / n * 1e3:7.1f} us each "
                  f"{2.0 * M * N * K * n / (t * 1e-3) / 1e12:7.1f} TF  {kinds}", file=sys.stderr, flush=True)


class CollectiveTimer:
    """HIP-event brackets around the step's collectives (torch.distributed = RCCL), recorded on the stream each one is issued from:
    the packed all-gather and the reduce-scatter inside ClipLoss, the flat gradient all-reduce of the trainer.  A synchronous collective
    is bracketed call-to-return (the issuing stream waits for RCCL's stream, so the bracket holds the transfer AND the wait for the
    slowest rank); an async one (bucketed full-fine-tune all-reduce) from issue to the end of its wait()."""

    NAMES = {"all_gather_into_tensor": "all_gather", "reduce_scatter_tensor": "reduce_scatter", "all_reduce": "all_reduce"}

    def __init__(self):
        self.enabled, self.events = False, {v: [] for v in self.NAMES.values()}
        self.bytes = {v: 0 for v in self.NAMES.values()}

    def install(self, dist):
        timer = self

        class _Work:
            def __init__(self, w, e0, key):
                self._w, self._e0, self._key = w, e0, key

            def wait(self, *a, **k):
                r = self._w.wait(*a, **k)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                timer.events[self._key].append((self._e0, e1))
                return r

            def __getattr__(self, n):
                return getattr(self._w, n)

        def wrap(fname, key):
            inner = getattr(dist, fname)

            def timed(*a, **kw):
                if not timer.enabled:
                    return inner(*a, **kw)
                t = a[0]
                timer.bytes[key] += t.numel() * t.element_size()
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = inner(*a, **kw)
                if kw.get("async_op"):
                    return _Work(r, e0, key)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                timer.events[key].append((e0, e1))
                return r

            setattr(dist, fname, timed)

        for fname, key in self.NAMES.items():
            wrap(fname, key)

    def result(self, steps):
        out = {}
        for key, evs in self.events.items():
            out[key] = {"ms_per_step": sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(steps, 1), "calls_per_step": len(evs) / max(steps, 1),
                        "bytes_per_step": self.bytes[key] / max(steps, 1)}
        return out


def _pci_slot(local_rank: int):
    """'dddd:bb:dd' of the bench's own device (torch device properties; no GPU call beyond what the bench already made)."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}"
    except Exception:
        return None


class BoardSampler(threading.Thread):
    """Board power and shader clock (sysfs hwmon, read-only, no GPU call) at 10 Hz while the timed region runs — of the bench's OWN
    device only: the card whose PCI slot (`/sys/class/drm/card*/device/uevent` PCI_SLOT_NAME) equals the slot torch reports for
    `local_rank`.  On a shared or multi-GPU host another tenant's card may draw more, so nothing is guessed: when the slot cannot
    be matched there is no `board` entry and no `frac_at_delivered_clock`."""

    def __init__(self, local_rank=0, period=0.1):
        super().__init__(daemon=True)
        self.period, self.samples, self._stop_ev = period, [], threading.Event()
        self.card = None
        slot = _pci_slot(local_rank)
        for dev_dir in sorted(glob.glob("/sys/class/drm/card*/device")):
            name = None
            try:
                with open(f"{dev_dir}/uevent") as f:
                    for line in f:
                        if line.startswith("PCI_SLOT_NAME="):
                            name = line.split("=", 1)[1].strip().lower()
            except OSError:
                continue
            if slot is None or name is None or not name.startswith(slot):   # PCI_SLOT_NAME = dddd:bb:dd.f
                continue
            for d in sorted(glob.glob(f"{dev_dir}/hwmon/hwmon*")):
                for pn in ("power1_input", "power1_average"):
                    if self._rd(f"{d}/{pn}") is not None:
                        self.card = (d, pn, name)
                        break
                if self.card:
                    break
            break

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return float(f.read()) / 1e6
        except (OSError, ValueError):
            return None

    def run(self):
        if self.card is None:
            return
        d, pn, _ = self.card
        while not self._stop_ev.is_set():
            self.samples.append((time.perf_counter(), self._rd(f"{d}/{pn}"), self._rd(f"{d}/freq1_input")))
            self._stop_ev.wait(self.period)

    def stop(self):
        self._stop_ev.set()
        self.join(timeout=2)

    def window(self, t0, t1):
        if self.card is None:
            return None
        rows = [(p, c) for t, p, c in self.samples if t0 <= t <= t1]
        pw = sorted(p for p, _ in rows if p is not None)
        ck = [c for _, c in rows if c is not None]
        if not pw:
            return None
        d, pn, name = self.card
        cap = self._rd(f"{d}/power1_cap")
        sclk = sum(ck) / len(ck) if ck else None
        return {"board_power_w": sum(pw) / len(pw), "board_power_w_p90": pw[min(len(pw) - 1, int(0.9 * len(pw)))], "power_cap_w": cap,
                "sclk_mhz": sclk, "samples": len(pw), "source": f"{d} ({pn}, freq1_input; PCI {name} = this rank's device), 10 Hz over the timed region"}


def pmc_traffic(per_gpu_batch: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic*.json,
    written by tools/pmc_traffic.py; PMC counters cannot be collected from inside this process).  Only a file measured on
    THIS kernel source (csrc hash) and THIS per-GPU batch is quoted; anything else is reported as stale with traffic = null."""
    import glob

    from clibd_amd.build import csrc_hash

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic*.json")))
    if not files:
        return None, None, "no PMC file"
    cur = csrc_hash()
    pick, why = None, None
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("csrc_sha16") == cur and d.get("per_gpu_batch") == per_gpu_batch:
            pick = (f, d)
            break
    if pick is None:
        return None, os.path.basename(files[-1]), f"stale: no PMC file for csrc {cur} at per-GPU batch {per_gpu_batch}"
    f, d = pick
    n = b = 0.0
    for name, k in d["kernels"].items():
        if "gemm256" in name:
            launches = max(k["launches_fetch_pass"], 1)
            n += launches
            b += launches * k["hbm_bytes_per_launch"]
    return (b / n if n else None), os.path.basename(f), None


def cpu_baseline(batch: int, steps: int):
    """Oracle (CPU restatement of the reference's fp32 path, pinned to it by tests/golden) training step on BASELINE configs[0]:
    fwd + loss + bwd + AdamW, full-size towers, batch 32, fp32.  One warm-up step (oneDNN primitive creation, allocator),
    a one-step probe per candidate thread count, then the median of `steps` timed steps at the fastest count."""
    from oracle import clibd_oracle as O

    ncpu = os.cpu_count() or 1
    torch.manual_seed(42)
    model = O.build_image_dna_model()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    g = torch.Generator().manual_seed(42)
    image = torch.rand(batch, 3, 224, 224, generator=g)
    dna = torch.cat([torch.zeros(batch, 1, dtype=torch.long), torch.randint(3, 1027, (batch, 132), generator=g)], dim=1)
    labels = torch.arange(batch)

    def one():
        t0 = time.perf_counter()
        loss = O.train_step(model, opt, image, dna, labels)
        return time.perf_counter() - t0, float(loss)

    # thread count: all hardware threads of a 2-socket SMT host is NOT the fastest setting for this step (M = 6304-row GEMMs
    # split into tiny per-thread panels + cross-socket traffic), so a few counts are probed, one step each, after a warm-up
    cands = sorted({min(c, ncpu) for c in (16, 32, 64)})   # measured on the 256-thread GPU host: 32: 7.4 s, 64: 11.7 s, 128: 32.7 s, 256: 319 s per step
    torch.set_num_threads(cands[0])
    one()  # warm-up (not timed)
    probe = {}
    for c in cands:
        torch.set_num_threads(c)
        probe[c] = one()[0]
        if probe[c] > 1.5 * min(probe.values()):
            break   # past the knee: more threads only get slower
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    times, loss = [], 0.0
    for _ in range(max(steps, 1)):
        dt, loss = one()
        times.append(dt)
    med = sorted(times)[len(times) // 2]
    return {"value": batch / med, "unit": "paired samples/s", "cores": threads, "kind": "port", "host_cpu_count": ncpu,
            "sample": f"BASELINE configs[0]: Image+DNA training step (fwd+loss+bwd+AdamW), batch {batch}, fp32, ViT-B/16 + BERT-base(133 tok) "
                      f"LoRA r=4, torch {torch.__version__} CPU; 1 warm-up step, thread probe {{{', '.join(f'{c}: {t:.1f}s' for c, t in probe.items())}}}, "
                      f"then median of {len(times)} steps at {threads} threads ({', '.join(f'{t:.1f}' for t in times)} s), loss {loss:.4f}"}


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X fp32-input MFMA = the fp32 vector rate (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0


def eval_bench(args, model, batch, b, world, rank, dev, dist, timer):
    """`--eval`: the inference side of the path (reference epoch/inference_epoch.py:43-111 + util/util.py:521-528).
    value = paired samples/s of the no-grad embedding forward (both towers, L2-normalise), whole job; `roofline` = the forward
    GEMMs against the bf16 MFMA peak; `k10` = the streaming top-k search of `--eval-queries` image embeddings against a key
    bank of `--eval-keys` unit vectors (exact fp32 scores on the fp32 MFMA: bound by that pipe, 2 Q Nk D FLOP at 157 TFLOP/s;
    its bytes — the bank is streamed once per 64-query block out of L2 / Infinity Cache — are quoted against HBM peak too)."""
    from clibd_amd import ops

    model.eval()

    def fwd():
        with torch.no_grad():
            out = model(batch["image"], batch["dna"], batch["text"])
        model.join_streams()
        return out

    for _ in range(args.warmup):
        out = fwd()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fwd()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    gemm = None
    if not args.no_gemm_timing:
        model.overlap_towers = False            # per-launch durations of the kernel itself (towers serialized), as in the training line
        fwd()
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(min(args.steps, 5)):
            fwd()
        torch.cuda.synchronize()
        timer.enabled = False
        model.overlap_towers = True
        gemm = timer.result()
        gemm["steps"] = min(args.steps, 5)
    k10 = None
    if rank == 0:
        Q, Nk, D, k = min(args.eval_queries, b), args.eval_keys, out[0].shape[1], 5
        g = torch.Generator(device=dev).manual_seed(7)
        keys = torch.nn.functional.normalize(torch.randn((Nk, D), generator=g, device=dev), dim=1)
        q = out[0][:Q].detach().float().contiguous()
        ops.topk_ip(q, keys, k)                  # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2
        e0.record()
        for _ in range(reps):
            sim, idx = ops.topk_ip(q, keys, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * Q * Nk * D
        algo_bytes = 4.0 * D * (Q + Nk) + 12.0 * Q * k
        k10 = {"kernel": "topk_ip_stream_kernel (fused exact-fp32 scores + running top-8, no Q x Nk matrix)", "queries": Q, "keys": Nk, "dim": D,
               "k": k, "ms_per_launch": ms, "bound": "mfma-fp32", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
               "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
               "queries_per_s": Q / (ms * 1e-3), "algorithmic_GBps": algo_bytes / (ms * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBPS,
               "hbm_frac": algo_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
               "note": "exact fp32 inner products (faiss.IndexFlatIP semantics, indices bit-exact) run on the fp32-input MFMA at 1/16 of the "
                       "bf16 rate: the search is bound by that pipe, not by HBM (the key bank is 4 D Nk bytes, read once per 64-query block "
                       "from L2 / Infinity Cache)"}
        if D % 64 == 0 and Nk >= 4096:
            # the pre-filtered search: same indices and similarities (checked here against the exact launch above), bf16 MFMA rate
            eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eb0.record()
            bank = ops.KeyBank(keys)
            eb1.record()
            ops.topk_ip_fast(q, bank, k)              # warm-up
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for _ in range(reps):
                fsim, fidx, fovf = ops.topk_ip_fast(q, bank, k)
            f1.record()
            torch.cuda.synchronize()
            fms = f0.elapsed_time(f1) / reps
            nov = int(fovf.sum())
            okr = fovf == 0
            bf16_bytes = 2.0 * D * Nk * ((Q + 127) // 128)     # the bf16 bank is streamed once per 128-query block
            k10["prefiltered"] = {
                "kernel": "topk_bf16_stream_kernel + topk_rescore_kernel (bf16 approximate scores -> exact fp32 re-score inside the rigorous error band)",
                "ms_per_launch": fms, "speedup_vs_exact": ms / fms, "queries_per_s": Q / (fms * 1e-3), "prepare_key_bank_ms_once": eb0.elapsed_time(eb1),
                "identical_to_exact": bool(torch.equal(fidx[okr], idx[okr]) and torch.equal(fsim[okr], sim[okr])), "overflowed_queries": nov,
                "bound": "hbm", "achieved": bf16_bytes / (fms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": bf16_bytes / (fms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "mfma_bf16_TFLOPs": flops / (fms * 1e-3) / 1e12,
                "note": "algorithmic bytes = the bf16 key bank (2 D Nk) once per 128-query block; it exceeds the Infinity Cache at this size, so "
                        "the stream comes from HBM"}
    if rank == 0:
        GF_PER_PAIR_FWD = 58.78   # ViT-B/16 35.18 + BarcodeBERT 23.60 GFLOP per pair, forward
        pairs_per_s = b * world * args.steps / elapsed
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": "gemm256_bf16_nt_kernel (forward kinds)", "step_frac": pairs_per_s * GF_PER_PAIR_FWD * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12),
                "traffic_note": "no PMC pass for the eval path"}
        if gemm:
            roof.update(achieved=gemm["tflops"], frac=gemm["frac"], launches=gemm["launches"], gemm_ms_per_step=gemm["total_ms"] / gemm["steps"],
                        avg_launch_us=gemm["total_ms"] / gemm["launches"] * 1e3, algorithmic_bytes_per_launch=gemm["bytes_per_launch"])
        out_line = {"metric": f"eval: paired samples/sec, no-grad embedding forward (I+D), batch {b * world}", "value": pairs_per_s,
                    "unit": "paired samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                    "higher_is_better": True, "scaling": "strong" if args.per_gpu_batch is None else "weak", "vs_baseline": None, "dtype": "bf16",
                    "data": "synthetic (rand 224x224 images, random 660-nt barcodes = 133 5-mer tokens, random-init weights, random unit key bank)",
                    "config": {"workload": f"eval path (secondary, SURVEY 8f-2): get_feature_and_label-style no-grad forward of ViT-B/16 + BarcodeBERT at "
                                           f"{world} GPU x {b}, then top-5 inner-product search", "per_gpu_batch": b, "global_batch": b * world,
                               "parallelism": f"dp{world}"},
                    "roofline": roof, "k10": k10}
        emit(out_line)


def configs4_record(args, model, trainer, dev, world, rank, dist, timer):
    """BASELINE configs[4] at its per-GPU batch (1024), bf16 against the recommended fp8 mode, same model, same process: see the call site."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.train import Trainer

    b4 = args.configs4_batch
    with_full = args.full_finetune
    batch4 = synthetic_batch(b4, dev, seed=44, rank=rank, with_text=False)
    fresh4 = synthetic_batch(b4, dev, seed=45, rank=rank, with_text=False)
    nsteps = min(args.steps, 5)

    def timed_steps():
        trainer.step(batch4["image"], batch4["dna"], None, batch4["labels"])     # warm-up at this shape / mode (workspaces, weight images)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            trainer.step(batch4["image"], batch4["dna"], None, batch4["labels"])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()) / nsteps * 1e3

    towers = [enc.tower() for enc in (model.image_encoder, model.dna_encoder) if enc is not None]
    params = list(trainer.optimizer.param_groups[0]["params"])
    crit_box = [trainer.criterion]

    def grad_vector(bt):
        """the step's gradient (forward + loss + backward, no optimizer) as one fp32 vector; same dropout masks on every call"""
        sinks = [tw.grad_sink for tw in towers]
        for tw in towers:
            tw.grad_sink = None          # plain autograd outputs, the trainer's flat bucket untouched
        try:
            torch.manual_seed(4242)      # the towers draw their dropout base seeds from the CPU generator
            hi, hd, _, scale, _ = model(bt["image"], bt["dna"], None)
            loss = crit_box[0](hi, hd, None, bt["labels"], scale)
            gs = torch.autograd.grad(loss, params, allow_unused=True)
            model.join_streams()
            torch.cuda.synchronize()
            return torch.cat([(torch.zeros_like(p) if g is None else g).detach().flatten().double() for p, g in zip(params, gs)])
        finally:
            for tw, sk in zip(towers, sinks):
                tw.grad_sink = sk

    def set_mode(on):
        """False: bf16.  True: the recommended mode.  "dgrad_all": bf16 forward, 8-bit dgrad on BOTH towers (the fastest mode that holds the 0.98 gate)."""
        if on == "dgrad_all":
            model.enable_fp8_forward(enabled=False)
            model.enable_fp8_dgrad(towers="all", enabled=True)
            return
        if on == "ffn_dgrad_all":        # the recommended forward selection with the 8-bit dgrad on both towers: the fastest combination measured
            model.enable_fp8_dgrad(towers="all", enabled=True)
            model.enable_fp8_forward(calibration_inputs=(batch4["image"], batch4["dna"], None), towers="pooled_ffn")
            return
        model.enable_fp8_dgrad(towers="pooled", enabled=bool(on))
        if with_full:
            return      # trainable base weights: the 8-bit dgrad only (the fp8 FORWARD needs frozen weights: its weight gradients would want bf16 GEMM inputs)
        if on:
            model.enable_fp8_forward(calibration_inputs=(batch4["image"], batch4["dna"], None), towers="pooled_ffn")
        else:
            model.enable_fp8_forward(enabled=False)

    second = True               # the 8-bit dgrad on BOTH towers: runs with frozen and with trainable base weights (tests/test_dgrad8_gpu.py, both towers)
    third = not with_full       # + the fp8 forward: frozen base weights only

    try:
        ms16 = timed_steps()
        set_mode(True)
        ms8 = timed_steps()
        ms8b = ms8c = None
        if second:
            set_mode("dgrad_all")
            ms8b = timed_steps()
            if third:
                set_mode("ffn_dgrad_all")
                ms8c = timed_steps()
            set_mode(True)
        share = None
        if timer is not None and not args.no_gemm_timing:
            kept = (timer.events, timer.flops, timer.flops_fp8, timer.shapes, timer.bytes)     # the headline's per-launch records (--gemm-breakdown prints them later)
            timer.events, timer.flops, timer.flops_fp8, timer.shapes, timer.bytes = [], 0.0, 0.0, [], 0.0
            timer.enabled = True
            trainer.step(batch4["image"], batch4["dna"], None, batch4["labels"])
            torch.cuda.synchronize()
            timer.enabled = False
            r = timer.result()
            share = r["fp8_flop_share"] if r else None
            timer.events, timer.flops, timer.flops_fp8, timer.shapes, timer.bytes = kept
        def cosines():
            out_ = {}
            for name, bt in (("train_batch", batch4), ("fresh_batch", fresh4)):
                set_mode(True)
                g8 = grad_vector(bt)
                if second:
                    set_mode("dgrad_all")
                    g8b = grad_vector(bt)
                    if third:
                        set_mode("ffn_dgrad_all")
                        g8c = grad_vector(bt)
                set_mode(False)
                g16 = grad_vector(bt)
                out_[name] = float((g8 @ g16) / (g8.norm() * g16.norm()).clamp_min(1e-300))
                if second:
                    out_["dgrad_all_" + name] = float((g8b @ g16) / (g8b.norm() * g16.norm()).clamp_min(1e-300))
                if third:
                    out_["ffn_dgrad_all_" + name] = float((g8c @ g16) / (g8c.norm() * g16.norm()).clamp_min(1e-300))
            return out_

        def spread_of(bt):
            """mean mutual cosine of the image embeddings of 256 pairs: ~1 at random init (every row nearly parallel), falls as the model learns"""
            with torch.no_grad():
                hi = model(bt["image"][:256], bt["dna"][:256], None)[0]
            model.join_streams()
            n_ = hi.shape[0]
            return float(((hi @ hi.T).sum() - n_) / (n_ * (n_ - 1)))

        # gradient fidelity (a) of the model exactly as the timed steps left it — at random init every embedding is nearly parallel to every
        # other, the gradient is one common direction and ANY mode reads ~1.0 here — and (b) after a short spreading phase (bf16, `spread_steps`
        # optimizer steps on the first 32 pairs at lr 1e-3, the protocol of tests/test_fp8_gpu.py::test_fp8_gradients_on_spread_embeddings), where
        # the gradient separates samples and the fp8 operands' noise shows
        set_mode(False)
        cos_raw, spread_raw = cosines(), spread_of(fresh4)
        # The timed steps (global batch 2048 at the scaled learning rate, random-init towers) leave the embeddings MORE parallel than random
        # init does (mutual cosine 0.9998), and 40 small-batch steps from there do not spread them.  The spreading phase therefore starts from
        # freshly initialised adapters / heads / temperature with a fresh AdamW (the frozen towers are the same), exactly the protocol of the
        # fidelity test; the timed trainer is not used again after this point.
        import math as _m
        with torch.no_grad():
            for enc in (model.image_encoder, model.dna_encoder):
                for wa in enc.w_As:
                    torch.nn.init.kaiming_uniform_(wa.weight, a=_m.sqrt(5))
                for wb in enc.w_Bs:
                    wb.weight.normal_(0, 0.02)
            model.image_encoder.base_image_encoder.head.reset_parameters()
            model.dna_encoder.base_dna_encoder.cls.predictions.decoder.reset_parameters()
            model.logit_scale.fill_(_m.log(1 / 0.07))
        for tw in towers:
            tw.invalidate_weight_images()
        spread_steps = 40
        spread_tr = Trainer(model, lr=1e-3, world_size=world, rank=rank, all_gather=True)
        params[:] = list(spread_tr.optimizer.param_groups[0]["params"])
        crit_box[0] = spread_tr.criterion
        nb = min(32, b4)
        for _ in range(spread_steps):
            spread_tr.step(batch4["image"][:nb], batch4["dna"][:nb], None, batch4["labels"][:nb])
        cos, spread_after = cosines(), spread_of(fresh4)
        set_mode(False)
    except Exception as e:   # the side record must never take the headline line down with it
        try:
            set_mode(False)
        except Exception:
            pass
        return {"error": repr(e)}
    if world > 1:   # one number per job: the worst rank's cosine
        keys = ["train_batch", "fresh_batch"] + (["dgrad_all_train_batch", "dgrad_all_fresh_batch"] if second else []) + (["ffn_dgrad_all_train_batch", "ffn_dgrad_all_fresh_batch"] if third else [])
        c = torch.tensor([cos[k] for k in keys] + [cos_raw[k] for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.MIN)
        cos = {k: float(c[i]) for i, k in enumerate(keys)}
        cos_raw = {k: float(c[len(keys) + i]) for i, k in enumerate(keys)}
    # the fastest of the measured modes whose in-run cosine (the lower of training / unseen batch) holds the 0.98 gate
    cands = [("fp8", ms8, min(cos["train_batch"], cos["fresh_batch"]))]
    if second:
        cands += [("fp8_dgrad_all", ms8b, min(cos["dgrad_all_train_batch"], cos["dgrad_all_fresh_batch"]))]
    if third:
        cands += [("fp8_ffn_dgrad_all", ms8c, min(cos["ffn_dgrad_all_train_batch"], cos["ffn_dgrad_all_fresh_batch"]))]
    ok = [c_ for c_ in cands if c_[2] >= 0.98]
    best = min(ok, key=lambda c_: c_[1]) if ok else None
    rec = {"workload": f"BASELINE configs[4] per-rank shape: {world} GPU x {b4} pairs (global {world * b4}), Image+DNA, " +
                        ("FULL fine-tune" if with_full else "LoRA r=4") + ", train mode",
            "per_gpu_batch": b4, "global_batch": world * b4, "steps": nsteps,
            "bf16": {"ms_per_step": ms16, "value": world * b4 / (ms16 * 1e-3), "unit": "paired samples/s"},
            "fp8": {"ms_per_step": ms8, "value": world * b4 / (ms8 * 1e-3), "unit": "paired samples/s",
                    "mode": ("--full-finetune --dgrad fp8-pooled: e4m3 operands on the MLP / projection activation-gradient GEMMs of the mean-pooled tower(s) "
                             "(BarcodeBERT), weights re-quantised every step; forward, attention, QKV and every weight gradient stay bf16" if with_full else
                             "--fp8-forward pooled_ffn --dgrad fp8-pooled: e4m3 operands on fc1 / fc2 of the mean-pooled tower(s) (BarcodeBERT) forward, and on "
                             "their MLP / projection activation-gradient GEMMs backward (per-row power-of-two scales); the ViT, attention, QKV and every "
                             "weight gradient stay bf16"),
                    "fp8_flop_share": share},
            "speedup": ms16 / ms8,
            **({"fp8_dgrad_all": {"ms_per_step": ms8b, "value": world * b4 / (ms8b * 1e-3), "unit": "paired samples/s", "speedup": ms16 / ms8b,
                                  "mode": ("--full-finetune " if with_full else "") + "--dgrad fp8 (bf16 forward): e4m3 operands on the MLP / projection activation-gradient GEMMs of BOTH towers (per-row power-of-two "
                                          "scales); every forward GEMM, attention, QKV and every weight gradient stay bf16 — the fastest mode whose gradient holds the 0.98 gate "
                                          "(tests/test_fp8_gpu.py: dgrad8(all))",
                                  "gradient_cosine_vs_bf16": {"train_batch": cos["dgrad_all_train_batch"], "fresh_batch": cos["dgrad_all_fresh_batch"],
                                                              "as_timed": {"train_batch": cos_raw["dgrad_all_train_batch"], "fresh_batch": cos_raw["dgrad_all_fresh_batch"]}}}}
               if second else {}),
            **({"fp8_ffn_dgrad_all": {"ms_per_step": ms8c, "value": world * b4 / (ms8c * 1e-3), "unit": "paired samples/s", "speedup": ms16 / ms8c,
                                      "mode": "--fp8-forward pooled_ffn --dgrad fp8: the recommended forward selection with the 8-bit dgrad on both towers",
                                      "gradient_cosine_vs_bf16": {"train_batch": cos["ffn_dgrad_all_train_batch"], "fresh_batch": cos["ffn_dgrad_all_fresh_batch"],
                                                                  "as_timed": {"train_batch": cos_raw["ffn_dgrad_all_train_batch"], "fresh_batch": cos_raw["ffn_dgrad_all_fresh_batch"]}}}}
               if third else {}),
            "gradient_cosine_vs_bf16": dict({k: v for k, v in cos.items() if not k.startswith(("dgrad_all_", "ffn_dgrad_all_"))}, spread_steps=spread_steps, image_embedding_mutual_cosine=spread_after,
                                            as_timed={"train_batch": cos_raw["train_batch"], "fresh_batch": cos_raw["fresh_batch"], "image_embedding_mutual_cosine": spread_raw}),
            "note": "same model, same process, measured after the headline passes; gradient_cosine = cos(fp8-mode gradient, bf16 gradient) over ALL trainable "
                    "tensors, same dropout masks, on the configs4 training batch (its first 32 pairs are the spreading phase's) and on a batch never seen, at per-GPU "
                    "batch; as_timed = the model exactly as the timed steps left it, where the embeddings are nearly parallel (image_embedding_mutual_cosine ~ 1) "
                    "and every mode reads ~1.0; train_batch / fresh_batch = the same towers with adapters / heads / temperature re-initialised and trained for "
                    "`spread_steps` bf16 AdamW steps (lr 1e-3) on 32 pairs, the protocol of tests/test_fp8_gpu.py (random-init towers, synthetic pairs: no pretrained "
                    "weights exist on this box); at N > 1 the minimum over ranks"}
    rec["fastest_at_cosine_0.98"] = ({"record": best[0], "speedup": ms16 / best[1], "gradient_cosine_vs_bf16_min": best[2]} if best else None)
    return rec


_JSON_FD = None   # the process's original stdout, saved by main() before fd 1 is pointed at stderr


def emit(obj) -> None:
    """The ONE line of stdout: written to the saved descriptor (libraries that print to fd 1 - librccl's banner - land on stderr)."""
    sys.stdout.flush()
    if _JSON_FD is None:
        print(json.dumps(obj), flush=True)
    else:
        os.write(_JSON_FD, (json.dumps(obj) + "\n").encode())


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher (the reference starts its own ranks too: scripts/train_cl.py:365 mp.spawn):
    start `torch.distributed.run` with N fresh rank processes as a CHILD of this process — which has not touched the GPU
    (no HIP call is made before this point, and nothing is exec'ed) — pass its stdout through (rank 0's JSON line is the only
    thing any rank prints there) and return its exit code: a failed rank fails the launcher, which fails this process."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # stdout carries the JSON line only: anything else a rank or a backend writes there (gloo's connection banner in the
    # shared-GPU debug mode) is passed on to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    # stdout carries rank 0's JSON line and nothing else: librccl prints a version banner to fd 1 when a process group is
    # created (seen with the one-rank group of CLIBD_FORCE_COLLECTIVES), so fd 1 is pointed at stderr for the whole run and the
    # line is written to the saved descriptor at the end
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # CLIBD_BENCH_SHARED_GPU=1 (debugging only): run the N-rank code path on a box with fewer GPUs than ranks — ranks share
    # devices and gloo moves the device tensors (RCCL refuses two ranks per device).  The line it prints is marked invalid.
    shared_gpu = os.environ.get("CLIBD_BENCH_SHARED_GPU") == "1" and torch.cuda.device_count() < world
    if shared_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    # CLIBD_FORCE_COLLECTIVES=1 at N = 1 (code-path check on a 1-GPU box): a ONE-rank RCCL group, and the step takes its data-parallel
    # path over it (packed all-gather, reduce-scatter, gradient all-reduce: clibd_amd.model.loss_func.collectives_forced)
    forced = world == 1 and os.environ.get("CLIBD_FORCE_COLLECTIVES") == "1"
    if forced:
        import socket

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if world > 1 or forced:
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctimer = None
    if world > 1 or forced:
        ctimer = CollectiveTimer()
        ctimer.install(dist)   # before clibd_amd.model / .train bind `dist` (they call through the module attribute)

    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import (CLIBDDNAEncoder, CLIBDImageEncoder, CLIBDLanguageEncoder, SimpleCLIP, create_vit,
                                 load_pre_trained_bert, load_pre_trained_bioscan_bert)
    from clibd_amd.train import Trainer, scale_learning_rate

    torch.manual_seed(42)
    if args.per_gpu_batch is not None:
        b, scaling = args.per_gpu_batch, "weak"
    else:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} GPUs")
        b, scaling = args.global_batch // world, "strong"
    image_enc = CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768)
    dna_enc = CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768)
    text_enc = CLIBDLanguageEncoder(load_pre_trained_bert()[1], r=4, num_classes=768) if args.tri_modal else None
    model = SimpleCLIP(image_enc, dna_enc, text_enc).to(dev)
    with torch.no_grad():  # exercise the adapters: B != 0 (B = 0 at init would make half the LoRA backward trivially zero)
        for enc in (image_enc, dna_enc, text_enc):
            if enc is not None:
                for wb in enc.w_Bs:
                    wb.weight.normal_(0, 0.02)
    if args.full_finetune:
        for p_ in model.parameters():
            p_.requires_grad_(True)
    trainer = Trainer(model, lr=scale_learning_rate(1e-3, b, world_size=world), world_size=world, rank=rank, all_gather=True)
    batch = synthetic_batch(b, dev, seed=42, rank=rank, with_text=args.tri_modal)
    if args.dgrad:
        model.enable_fp8_dgrad(towers="pooled" if args.dgrad == "fp8-pooled" else "all", enabled=args.dgrad != "bf16")
    if args.fp8_forward:   # per-layer activation scales from one bf16 forward over the batch (outside the timed region)
        model.enable_fp8_forward(calibration_inputs=(batch["image"], batch["dna"], batch["text"]), towers=args.fp8_forward)

    timer = GemmTimer()
    if not args.no_gemm_timing:
        timer.install()
    if args.eval:
        eval_bench(args, model, batch, b, world, rank, dev, dist, timer)
        if world > 1:
            dist.destroy_process_group()
        return

    def one_step():
        return trainer.step(batch["image"], batch["dna"], batch["text"], batch["labels"])

    for _ in range(args.warmup):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sampler = BoardSampler(local_rank) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()   # the timed region carries NO per-launch event records (they cost ~1.3 ms per step of host+GPU time)
    host_wall, host_cpu = [], []   # per step: wall and CPU time this thread spent ENQUEUEING it (two clock reads per step, no sync)
    for _ in range(args.steps):
        hw, hc = time.perf_counter(), time.thread_time()
        loss = one_step()
        host_wall.append(time.perf_counter() - hw)
        host_cpu.append(time.thread_time() - hc)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0      # this rank's own clock, BEFORE the closing barrier makes every rank wait for the slowest
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    board = None
    if sampler is not None:
        sampler.stop()
        board = sampler.window(t0 + min(0.5, 0.25 * elapsed), t0 + elapsed)   # (the first half second: the clocks are still settling)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    loss_val = float(loss.item())

    # Kernel passes for `roofline` (same step, right after the timed region, not part of `value`): HIP events around every
    # GEMM launch, recorded on the launch stream.  Pass 1 runs exactly like the timed region: the two towers overlap on
    # separate streams, so a launch's event-to-event time includes the time its CUs were shared with the other tower's
    # kernels.  Pass 2 serializes the towers on one stream: its per-launch durations are what the kernel itself achieves.
    overlap_on = bool(getattr(model, "overlap_towers", False)) and (model.image_encoder is not None) and (model.dna_encoder is not None)
    overlapped = None
    if not args.no_gemm_timing:
        timer.enabled = True
        for _ in range(min(args.steps, 5)):
            one_step()
        torch.cuda.synchronize()
        timer.enabled = False
        overlapped = timer.result()
        overlapped["steps"] = min(args.steps, 5)
    serial = None
    if overlapped and overlap_on:
        timer.events, timer.flops, timer.flops_fp8, timer.shapes, timer.bytes = [], 0.0, 0.0, [], 0.0
        model.overlap_towers = False
        serial_steps = min(args.steps, 5)
        one_step()
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(serial_steps):
            one_step()
        torch.cuda.synchronize()
        timer.enabled = False
        model.overlap_towers = True
        serial = timer.result()
        serial["steps"] = serial_steps

    # Side measurement (never `value`): the same step at the REFERENCE's backward numerics.  The headline runs with the gradient of the
    # residual stream kept in bf16 between block halves (config.numerics.residual_grad, DESIGN §4's budget); the reference's autograd keeps
    # that stream in fp32, also under autocast.  Both figures belong in the line.
    ref_num = None
    if not args.no_ref_numerics and not args.fp8_forward and args.dgrad in (None, "bf16"):
        cur = {k: v for k, v in next(iter(model.numerics().values())).items() if k in REFERENCE_NUMERICS}
        if cur != REFERENCE_NUMERICS:
            model.set_numerics(**REFERENCE_NUMERICS)
            rsteps = min(args.steps, 5)
            one_step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tr = time.perf_counter()
            for _ in range(rsteps):
                one_step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            er = torch.tensor([time.perf_counter() - tr], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(er, op=dist.ReduceOp.MAX)
            ref_num = {"value": b * world * rsteps / float(er.item()), "unit": "paired samples/s", "ms_per_step": float(er.item()) / rsteps * 1e3,
                       "steps": rsteps, "numerics": dict(REFERENCE_NUMERICS),
                       "note": "the same step with every backward-numerics switch at the reference's semantics (fp32 residual-gradient stream, bf16 "
                               "GELU', two-phase attention backward, LayerNorm as its own pass): what the headline would read without the disclosed relaxations of config.numerics"}
            model.set_numerics(**cur)
            one_step()   # (the weight-image / workspace state of the default numerics is what the following passes measure)
            torch.cuda.synchronize()
        else:
            ref_num = {"value": None, "note": "the headline already runs at the reference's numerics"}

    # Side measurement (never `value`): the same step with the batch handed over as HOST tensors, as the reference's loop does
    # (epoch/train_epoch.py:26-32 `.to(device)` per step): pinned host buffers, async copies on the compute stream.
    h2d = None
    if not args.no_h2d:
        from clibd_amd.data import DevicePrefetcher

        host = {k: batch[k].cpu().pin_memory() for k in ("image", "dna", "labels")}
        host["text"] = None if batch["text"] is None else {k: v.cpu().pin_memory() for k, v in batch["text"].items()}
        nbytes = sum(t.numel() * t.element_size() for t in (host["image"], host["dna"], host["labels"])) + \
            (0 if host["text"] is None else sum(t.numel() * t.element_size() for t in host["text"].values()))
        hsteps = min(args.steps, 8)

        def timed(run):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            th = time.perf_counter()
            run()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            eh = torch.tensor([time.perf_counter() - th], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(eh, op=dist.ReduceOp.MAX)
            return float(eh.item())

        def sync_loop():      # the reference's loop: copy at the top of the step, on the compute stream
            for _ in range(hsteps):
                text_d = None if host["text"] is None else {k: v.to(dev, non_blocking=True) for k, v in host["text"].items()}
                trainer.step(host["image"].to(dev, non_blocking=True), host["dna"].to(dev, non_blocking=True), text_d,
                             host["labels"].to(dev, non_blocking=True))

        # clibd_amd.data.DevicePrefetcher: batch i+1 crosses PCIe on a copy stream under step i.  STEADY STATE: the prefetcher is built
        # (and its first batch requested) before the clock starts — in a training epoch that first copy is paid once per epoch, not once
        # per `hsteps`; round 4 timed it inside a 5-step loop, which charged 24 ms / 5 = 4.8 ms to every step.
        def prefetch_timed(hb):
            pf = DevicePrefetcher((hb for _ in range(hsteps + 1)), dev)
            first = next(pf)
            trainer.step(first["image"], first["dna"], first["text"], first["labels"])   # untimed: its copy was not under a step

            def run():
                for bt in pf:
                    trainer.step(bt["image"], bt["dna"], bt["text"], bt["labels"])
            return timed(run)

        sync_loop()           # warm-up (pinned staging, allocator)
        t_sync = timed(sync_loop)
        t_pref = prefetch_timed(host)
        # WHAT-IF leg: the dataset holds image BYTES (uint8 HDF5, util/dataset.py:185-195), but the reference augments on the host AFTER ToTensor
        # (Resize, RandomResizedCrop, flips, rotation on float tensors), so what it sends is fp32.  IF the pipeline emitted uint8 crops (or ran on
        # the device), handing those over and dividing on the device (clibd_patchify_u8) would move a quarter of the bytes.  The synthetic fp32
        # images are not multiples of 1/255, so this leg runs on its own uint8 batch of the same shape: same kernels, same FLOPs, different pixels.
        host8 = dict(host)
        host8["image"] = torch.randint(0, 256, tuple(host["image"].shape), dtype=torch.uint8).pin_memory()
        nbytes8 = nbytes - host["image"].numel() * 4 + host8["image"].numel()
        prefetch_timed(host8)   # warm-up of the uint8 patch gather
        t_pref8 = prefetch_timed(host8)
        h2d = {"value": b * world * hsteps / t_pref, "unit": "paired samples/s", "ms_per_step": t_pref / hsteps * 1e3,
               "steps": hsteps, "host_bytes_per_step_per_gpu": nbytes,
               "copy_at_top_of_step": {"value": b * world * hsteps / t_sync, "ms_per_step": t_sync / hsteps * 1e3},
               "uint8_images": {"value": b * world * hsteps / t_pref8, "ms_per_step": t_pref8 / hsteps * 1e3, "host_bytes_per_step_per_gpu": nbytes8,
                                "note": "WHAT-IF leg: uint8 224x224 crops handed over as bytes and divided by 255 on the device (clibd_patchify_u8: bit-identical "
                                        "to patchify(u8.float() / 255)): a quarter of the PCIe bytes.  Not equivalent to the reference's data path, whose ToTensor is "
                                        "followed on the host by Resize / RandomResizedCrop / flips / rotation on float tensors (util/dataset.py): it applies only if "
                                        "the augmentation pipeline emits uint8 crops or runs on the device"},
               "note": "same step with the batch handed over as pinned HOST tensors every step (PCIe-inclusive); `value` here = next batch "
                       "prefetched on a copy stream under the current step (clibd_amd.data.DevicePrefetcher, steady state: the first batch's "
                       "copy is requested before the clock starts), copy_at_top_of_step = the reference's loop shape (synchronous "
                       ".to(device) in front of the step).  Reported beside the headline `value`, never as it"}
        del host8
        del host


    # Multi-GPU diagnostics (never `value`): what a first SCALE run needs to be read from its one line.  HIP-event
    # brackets around the step's three collectives on the streams they are issued from (a few extra steps right after the timed region),
    # every rank's OWN step time (clock stopped before the closing barrier), and what the process group says it is.
    multi = None
    if ctimer is not None:
        csteps = min(args.steps, 5)
        one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctimer.enabled = True
        for _ in range(csteps):
            one_step()
        torch.cuda.synchronize()
        ctimer.enabled = False
        own = torch.tensor([own_elapsed / args.steps * 1e3], dtype=torch.float64, device=dev)
        allr = torch.empty((world,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, own)
        per_rank = [float(v) for v in allr.tolist()]
        try:
            nccl_v = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            nccl_v = None
        multi = {"collectives_ms": {k: v["ms_per_step"] for k, v in ctimer.result(csteps).items()},
                 "collectives_detail": ctimer.result(csteps),
                 "per_rank_ms": per_rank, "rank_skew_ms": max(per_rank) - min(per_rank),
                 "rccl_ranks": {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": nccl_v,
                                "devices": torch.cuda.device_count()},
                 "note": "collectives_ms: rank 0, HIP events on the issuing stream around each collective (all_gather = the packed embeddings + labels "
                         "inside ClipLoss, reduce_scatter = its backward, all_reduce = the flat gradient bucket + the loss slot), mean per step over "
                         f"{csteps} steps after the timed region — a bracket holds the transfer AND the wait for the slowest rank; per_rank_ms: each "
                         "rank's own mean step time over the timed region, its clock stopped before the closing barrier; rank_skew_ms = max - min of those"}

    # Side record (never `value`): BASELINE configs[4] — the BIOSCAN-5M-shaped run, per-GPU batch 1024, "fp8 MFMA attention/GEMM path" — as this
    # build recommends it: fp8 forward on the MLP pair of the mean-pooled towers ("pooled_ffn") + their 8-bit dgrad ("fp8-pooled"), against
    # the bf16 step at the same batch, with the gradient cosine of THIS model (as the steps above left it) on a fresh batch measured in-run.
    cfg4 = None
    if not args.no_configs4 and not args.fp8_forward and args.dgrad in (None, "bf16") and not args.tri_modal:
        cfg4 = configs4_record(args, model, trainer, dev, world, rank, dist, timer)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        pairs_per_s = b * world * args.steps / elapsed
        gemm = serial if serial is not None else overlapped
        if gemm and args.gemm_breakdown:
            timer.breakdown(gemm["steps"])
        gf_pair = GF_PER_PAIR_FULLFT if args.full_finetune else GF_PER_PAIR_TRAIN   # reference model FLOPs of the configuration being run
        step_frac = pairs_per_s * gf_pair * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12)
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": "gemm256_bf16_nt_kernel (256x256x64 persistent tiles; gemm_bf16_nt_kernel 128x128 for small shapes)", "step_frac": step_frac,
                "step_frac_gflop_per_pair": gf_pair,
                "note": "achieved = sum of 2MNK over every GEMM launch / summed HIP-event durations (rank 0, events on the launch "
                        "stream), measured right after the timed region on the same step with the towers serialized on one stream; "
                        "timed_region = the same quantity with the towers overlapping on two streams exactly as in the timed region, where a "
                        "launch's duration includes CU sharing; step_frac = pairs/s x the REFERENCE model's FLOPs per pair (117.6 GF LoRA step, "
                        "176.3 GF full fine-tune: BASELINE.md section 3; the class-row-only last ViT block executes ~4 % fewer) / (n_gpus x peak): "
                        "the whole step against the MFMA roof"}
        traffic, traffic_src, stale = pmc_traffic(b)
        if traffic is not None:
            roof["traffic"] = traffic
            roof["traffic_source"] = f"profiles/{traffic_src}: mean HBM bytes per gemm256 launch (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate --pmc passes)"
        else:
            roof["traffic_note"] = stale
        if gemm:
            gsteps = gemm["steps"]
            roof.update(achieved=gemm["tflops"], frac=gemm["frac"], launches=gemm["launches"],
                        gemm_ms_per_step=gemm["total_ms"] / gsteps, avg_launch_us=gemm["total_ms"] / gemm["launches"] * 1e3,
                        algorithmic_bytes_per_launch=gemm["bytes_per_launch"])
            if gemm.get("fp8_flop_share"):   # configs[4] lines: frac prices every launch at the dense peak of ITS operand type (fp8 = 2 x bf16)
                roof["fp8_flop_share"] = gemm["fp8_flop_share"]
                roof["frac_note"] = "fp8 launches (forward and 8-bit dgrad) priced at 5.0 PFLOP/s, bf16 launches at 2.5"
            if serial is not None:
                roof["timed_region"] = {"achieved": overlapped["tflops"], "gemm_ms_per_step": overlapped["total_ms"] / overlapped["steps"],
                                        "avg_launch_us": overlapped["total_ms"] / overlapped["launches"] * 1e3, "streams": 2}
        if board is not None and board.get("sclk_mhz"):
            # The board holds its power cap by lowering the shader clock: the MFMA peak the step was actually offered
            board["mfma_peak_at_sclk_tflops"] = PEAK_BF16_TFLOPS * board["sclk_mhz"] / MAX_SCLK_MHZ
            if roof.get("achieved"):
                roof["frac_at_delivered_clock"] = roof["achieved"] / board["mfma_peak_at_sclk_tflops"]
            board["note"] = ("the step runs at the board's power cap: the shader clock is what gives (DESIGN.md §3.1, last bullet). Measured on this "
                             "board (profiles/r03_exp_power_roof.log): bare v_mfma_f32_16x16x32_bf16 with nothing else running 2034 TFLOP/s at "
                             "2.07 GHz / 1313 W; the same beside a 3.07 TB/s copy 1281 TFLOP/s at 1.61 GHz / 1400 W; every GEMM shape of the step "
                             "draws 1400 W at 1.63-1.83 GHz")
            roof["board"] = board
        out = {
            "metric": (f"paired samples/sec/step (I+D contrastive), global batch {b * world}" if not args.tri_modal
                       else f"triples/sec/step (I+D+T contrastive), global batch {b * world}"),
            "value": pairs_per_s, "unit": "paired samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": ("bf16" if not args.fp8_forward else "fp8 (e4m3) forward GEMMs + bf16") + ({"fp8": " + fp8 (e4m3) MLP / projection dgrad", "fp8-pooled": " + fp8 (e4m3) MLP / projection dgrad of the mean-pooled towers"}.get(args.dgrad, "")),
            "data": "synthetic (rand 224x224 images, random 660-nt barcodes = 133 5-mer tokens, random-init weights)",
            "config": {"workload": f"global batch {b * world} = {world} GPU x {b} (" + ("BASELINE.json metric config" if b * world == 2048 and not args.tri_modal and not args.full_finetune and not args.fp8_forward and args.dgrad in (None, "bf16") else "secondary config") + "): Image+DNA contrastive training step, ViT-B/16 + BarcodeBERT(BERT-base) " +
                                   ("FULL fine-tune (disable_lora)" if args.full_finetune else "LoRA r=4") +
                                   (", bf16 MFMA" if not args.fp8_forward else f", fp8-forward mode (BASELINE configs[4], towers={args.fp8_forward}): forward GEMMs of " + ({"pooled": "the mean-pooled towers (BarcodeBERT)", "pooled_ffn": "the MLP pair of the mean-pooled towers (BarcodeBERT)", "pooled_mlp": "the mean-pooled towers (BarcodeBERT) and the MLP pair of every ViT block"}.get(args.fp8_forward, "every tower")) + " on the fp8 MFMA, backward bf16") +
                                   (" + BERT-small text tower" if args.tri_modal else ""),
                       "per_gpu_batch": b, "global_batch": b * world, "parallelism": f"dp{world}", "image": "3x224x224", "dna_tokens": 133,
                       "loss": "soft-target InfoNCE over the all-gathered global batch", "optimizer": "fused AdamW",
                       "numerics": model.numerics(), "train_mode": bool(model.training)},
            "loss": loss_val, "roofline": roof,
        }
        hw_ms, hc_ms = sorted(1e3 * t for t in host_wall), sorted(1e3 * t for t in host_cpu)
        pick = lambda v, q: v[min(len(v) - 1, int(q * len(v)))]
        out["host_enqueue_ms"] = {
            "mean": sum(hw_ms) / len(hw_ms), "median": pick(hw_ms, 0.5), "p95": pick(hw_ms, 0.95),
            "cpu_mean": sum(hc_ms) / len(hc_ms), "cpu_median": pick(hc_ms, 0.5),
            "note": "rank 0, per timed step: wall time of Trainer.step() returning (no sync) and the CPU time this thread spent in it.  "
                    "wall > cpu means the host was BLOCKED (the HIP queue's back-pressure once it is a few steps ahead), not busy: the "
                    "host cost of a step is the cpu figure; it is on the critical path only if it exceeds ms_per_step"}
        if shared_gpu:
            out["invalid"] = "CLIBD_BENCH_SHARED_GPU: ranks shared a GPU over gloo (code-path check, not a measurement)"
        if forced:
            out["collectives"] = "CLIBD_FORCE_COLLECTIVES: the step's three collectives ran over a ONE-rank RCCL group (their launch path, no transfer)"
        if multi is not None:
            out.update(multi)
        if cfg4 is not None:
            out["configs4"] = cfg4
        if ref_num is not None:
            if ref_num.get("value") is None:
                ref_num.update(value=pairs_per_s, ms_per_step=ms_per_step)
            out["reference_numerics"] = ref_num
        if h2d is not None:
            out["h2d_inclusive"] = h2d
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        emit(out)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
