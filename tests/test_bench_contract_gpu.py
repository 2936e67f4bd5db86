"""bench.py's output contract on a GPU box: ONE line on stdout, a JSON object with the driver's keys, also when a process group
exists (librccl prints a version banner to fd 1 when the group is created; bench.py points fd 1 at stderr and writes its line to
the saved descriptor).  Small per-GPU batch so the check takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("forced", ["0", "1"])
def test_bench_prints_exactly_one_json_line(forced):
    env = dict(os.environ, CLIBD_FORCE_COLLECTIVES=forced)
    # (the forced-collectives run skips the host-batch legs; the plain run keeps them: round 5's steady-state / uint8 legs are part of the line)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--per-gpu-batch", "32", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--configs4-batch", "32"] + (["--no-h2d"] if forced == "1" else []), capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    if forced == "1" and r.returncode != 0 and "init_process_group" in r.stderr:
        pytest.skip("one-rank RCCL process group unavailable on this box")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["peak"] == 2500.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    # round 4: what arithmetic produced the line, and what the host paid to enqueue it
    assert d["config"]["train_mode"] is True                      # the reference steps in model.train() (train_epoch.py:19)
    num = d["config"]["numerics"]
    assert set(num) == {"image_encoder", "dna_encoder"} and all(v["forward"] == "bf16" and v["residual_grad"] in ("bf16", "fp32") and v["ln_fold"] in ("off", "on") and v["dgrad"] == "bf16" for v in num.values())
    he = d["host_enqueue_ms"]
    assert all(k in he for k in ("mean", "median", "p95", "cpu_mean", "cpu_median")) and 0 < he["cpu_median"] <= he["p95"] * 1.5 + 1.0
    assert roof["step_frac_gflop_per_pair"] == 117.6
    # round 5: the step at the reference's backward numerics rides in the same line; the host-fed legs are steady state + uint8 images
    rn = d["reference_numerics"]
    assert rn["value"] > 0 and rn["numerics"] == {"residual_grad": "fp32", "gelu_grad": "bf16", "attn_bwd": "2phase", "ln_fold": "off", "dgrad": "bf16"}
    # round 6: BASELINE configs[4] rides in the default line as a side record — bf16 against the recommended fp8 mode at the same batch, with the
    # fp8 FLOP share and the in-run gradient cosine (here at a toy batch; the driver's line carries per-GPU batch 1024)
    c4 = d["configs4"]
    assert "error" not in c4, c4
    assert c4["per_gpu_batch"] == 32 and c4["bf16"]["ms_per_step"] > 0 and c4["fp8"]["ms_per_step"] > 0 and c4["speedup"] > 0
    assert 0.05 < c4["fp8"]["fp8_flop_share"] < 0.6                         # the DNA tower's fc1 / fc2 forward + its MLP / projection dgrads
    gc = c4["gradient_cosine_vs_bf16"]
    assert 0.5 < gc["train_batch"] <= 1.0 and 0.5 < gc["fresh_batch"] <= 1.0 and gc["spread_steps"] == 40
    assert 0.5 < gc["as_timed"]["train_batch"] <= 1.0 and 0.5 < gc["as_timed"]["fresh_batch"] <= 1.0
    assert gc["image_embedding_mutual_cosine"] < gc["as_timed"]["image_embedding_mutual_cosine"] <= 1.0    # the spreading phase spread the embeddings
    # ... and the fastest mode that holds the 0.98 gate: bf16 forward with the 8-bit dgrad on both towers
    da = c4["fp8_dgrad_all"]
    assert da["ms_per_step"] > 0 and da["speedup"] > 0 and "--dgrad fp8" in da["mode"]
    assert 0.5 < da["gradient_cosine_vs_bf16"]["train_batch"] <= 1.0 and 0.5 < da["gradient_cosine_vs_bf16"]["fresh_batch"] <= 1.0
    dc = c4["fp8_ffn_dgrad_all"]
    assert dc["ms_per_step"] > 0 and dc["speedup"] > 0 and 0.5 < dc["gradient_cosine_vs_bf16"]["fresh_batch"] <= 1.0
    assert "fastest_at_cosine_0.98" in c4 and (c4["fastest_at_cosine_0.98"] is None or c4["fastest_at_cosine_0.98"]["record"] in ("fp8", "fp8_dgrad_all", "fp8_ffn_dgrad_all"))
    num2 = d["config"]["numerics"]
    assert all(v["forward"] == "bf16" and v["dgrad"] == "bf16" for v in num2.values())   # the side record switched its mode off again
    if forced == "1":
        assert "collectives" in d
        # round 6: a multi-GPU line is self-diagnosing — per-collective HIP-event times, every rank's own step time, the skew, the group
        cm = d["collectives_ms"]
        assert set(cm) == {"all_gather", "reduce_scatter", "all_reduce"} and all(v > 0 for v in cm.values()), cm
        det = d["collectives_detail"]
        assert det["all_gather"]["calls_per_step"] == 1 and det["reduce_scatter"]["calls_per_step"] == 1 and det["all_reduce"]["calls_per_step"] == 1
        assert det["all_gather"]["bytes_per_step"] == (2 * 32 * 768 + 2 * 32) * 4     # packed: two modalities' embeddings + the labels as fp32 slots
        assert len(d["per_rank_ms"]) == 1 and d["per_rank_ms"][0] > 0 and d["rank_skew_ms"] == 0.0
        assert d["rccl_ranks"]["world_size"] == 1 and d["rccl_ranks"]["backend"] == "nccl"
    else:
        h = d["h2d_inclusive"]
        assert h["value"] > 0 and h["copy_at_top_of_step"]["value"] > 0 and h["uint8_images"]["value"] > 0
        assert h["uint8_images"]["host_bytes_per_step_per_gpu"] < h["host_bytes_per_step_per_gpu"] / 3


def test_bench_line_of_the_fp8_configuration():
    """BASELINE configs[4] as bench.py runs it (`--fp8-forward pooled --dgrad fp8`): the line says so in dtype / config, it is a secondary
    config (never the metric's), the numerics record carries dgrad = fp8 for both towers and the fp8 forward for the pooled tower only, and
    the roofline prices the fp8 launches (forward and 8-bit dgrad) at the fp8 peak."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--per-gpu-batch", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-h2d", "--fp8-forward", "pooled", "--dgrad", "fp8"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "fp8" in d["dtype"] and "dgrad" in d["dtype"] and "secondary config" in d["config"]["workload"]
    num = d["config"]["numerics"]
    assert num["image_encoder"]["dgrad"] == "fp8" and num["dna_encoder"]["dgrad"] == "fp8"
    assert num["image_encoder"]["forward"] == "bf16" and num["dna_encoder"]["forward"].startswith("fp8")
    assert d["value"] > 0 and d["loss"] == d["loss"] and "reference_numerics" not in d
    assert 0 < d["roofline"]["frac"] < 1 and 0.2 < d["roofline"]["fp8_flop_share"] < 0.9   # DNA forward + MLP / projection dgrads of both towers
