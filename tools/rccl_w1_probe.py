"""Latency of the step's three collectives over a ONE-rank RCCL group (1-GPU box): what the calls cost before any byte crosses xGMI.

    python tools/rccl_w1_probe.py
"""
import os, socket, sys, time
import torch, torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
b, D = 2048, 768
packed = torch.randn(2 * b * D + 2 * b, device=dev); gathered = torch.empty_like(packed)
send = torch.randn(2 * b * D, device=dev); recv = torch.empty_like(send)
flat = torch.randn(1_480_064, device=dev)


def timeit(name, fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_issue = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    os.write(fd, f"{name:34s} host issue {t_issue * 1e6:8.1f} us/call   wall {t_all * 1e6:8.1f} us/call   one call on the stream {e0.elapsed_time(e1) * 1e3:8.1f} us\n".encode())


timeit("all_gather_into_tensor 12.6 MB", lambda: dist.all_gather_into_tensor(gathered, packed))
timeit("reduce_scatter_tensor 12.6 MB", lambda: dist.reduce_scatter_tensor(recv, send))
timeit("all_reduce 5.9 MB", lambda: dist.all_reduce(flat))
timeit("all_reduce 5.9 MB async + wait", lambda: dist.all_reduce(flat, async_op=True).wait())
timeit("all_reduce of a 4-byte scalar", lambda: dist.all_reduce(flat[:1]))
timeit("copy_ 12.6 MB (for scale)", lambda: gathered.copy_(packed))
dist.destroy_process_group()
