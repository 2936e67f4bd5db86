"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly what
include/clibd_hip.h declares (no compute calls here: there is no GPU on the authoring box)."""
import ctypes
import os
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "clibd_hip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(clibd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from clibd_amd import build

    return ctypes.CDLL(str(build.build(verbose=False)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("clibd_gemm_bf16_nt", "clibd_layernorm_fwd", "clibd_attention_fwd", "clibd_attention_bwd", "clibd_lora_wgrad",
                 "clibd_softce_rows_fwd", "clibd_softce_rows_bwd", "clibd_l2norm_fwd", "clibd_adamw_step"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in clibd_hip.h but not exported: {missing}"


def test_python_binding_covers_every_declared_symbol(lib):
    from clibd_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.load().clibd_abi_version() == _lib.ABI_VERSION == 5


def test_binding_refuses_a_library_built_from_other_sources(lib, monkeypatch):
    """clibd_build_hash() (baked in by clibd_amd/build.py) must equal the hash of the sources beside the library."""
    from clibd_amd import _lib, build

    L = _lib.load()
    assert L.clibd_build_hash().decode() == build.csrc_hash()
    monkeypatch.setattr(build, "csrc_hash", lambda: "0123456789abcdef")
    with pytest.raises(_lib.ClibdHipError, match="built from other kernel sources"):
        _lib._check_build_hash(L)


def test_epilogue_struct_layout_matches_header():
    from clibd_amd._lib import GemmEpilogue

    # 8 pointers + 8 int32 + 4 dropout words + (ABI 3) the three LN-fold pointers: 8*8 + 12*4 + 3*8 = 136 bytes
    assert ctypes.sizeof(GemmEpilogue) == 136
    assert [f[0] for f in GemmEpilogue._fields_] == ["bias", "rank_u", "rank_v", "aux_bf16", "residual_f32", "out_pre_bf16", "out_bf16",
                                                     "out_f32", "act", "ld_rank_u", "ld_aux", "ld_res", "ld_pre", "ld_out_bf16",
                                                     "ld_out_f32", "split_k", "drop_seed", "drop_thr16", "drop_scale", "drop_ld", "row_sums", "row_stats", "col_sum_w"]


def test_host_side_validation_needs_no_gpu(lib):
    """Bad shapes are rejected on the host before any launch."""
    from clibd_amd import _lib

    L = _lib.load()
    ep = _lib.GemmEpilogue()
    assert L.clibd_gemm_bf16_nt(None, 64, None, 64, 8, 16, 64, ctypes.byref(ep), None) == -1
    assert b"null" in L.clibd_last_error()
    assert L.clibd_attention_fwd(ctypes.c_void_p(16), 1, 300, 1, None, ctypes.c_void_p(16), 300, 300, None) == -1
    assert L.clibd_softce_workspace_bytes(32, 32, 768) > 32 * 32 * 4
    # round 6: the stream-K tail plan (256 CUs assumed without a device): 591 tiles = 2 rounds + 79 -> 3 K-slices; 399 tiles (tail 143) and K = 768: none
    assert L.clibd_gemm_tail_workspace_bytes(50432, 768, 3072) == 1024 + 79 * 2 * 256 * 256 * 4
    assert L.clibd_gemm_tail_workspace_bytes(50432, 768, 2304) == 1024 + 79 * 2 * 256 * 256 * 4
    assert L.clibd_gemm_tail_workspace_bytes(34048, 768, 3072) == 0 and L.clibd_gemm_tail_workspace_bytes(50432, 3072, 768) == 0
    assert L.clibd_gemm_tail_workspace_bytes(403456, 768, 3072) == 1024 + 120 * 1 * 256 * 256 * 4      # the metric's batch: 4 728 tiles = 18 rounds + 120 -> 2 slices
    assert L.clibd_gemm_tail_workspace_bytes(403456, 768, 3072) <= 48 * 1024 * 1024 + 1024


def test_product_path_has_no_cpu_fallback():
    import torch

    from clibd_amd import ops

    with pytest.raises(ValueError):
        ops.l2norm_fwd(torch.zeros(4, 8))
    for py in (ROOT / "clibd_amd").rglob("*.py"):
        assert "oracle" not in re.sub(r"#.*|\"\"\".*?\"\"\"", "", py.read_text(), flags=re.S), f"{py} references the oracle"


def test_gemm256_asynchronous_operands_are_not_touched_before_their_wait():
    """gemm256's epilogue bias is an inline-asm load awaited by an inline-asm s_waitcnt (hipcc does not know the registers are
    written asynchronously): no instruction of any instantiation may touch them in between (tools/check_gemm256_isa.py, hipcc -S)."""
    import importlib.util
    import os

    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed: the ISA cannot be generated here")
    spec = importlib.util.spec_from_file_location("check_gemm256_isa", os.path.join(os.path.dirname(__file__), "..", "tools", "check_gemm256_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_hot_kernels_use_no_scratch():
    """Tripwire (round 6): every kernel of the hot units keeps its working set in registers.  A generic-epilogue branch that took a local array by
    pointer once demoted the 256x256 GEMM's 128 accumulators to 528 bytes of scratch per lane — no 'VGPRs Spill' in the compiler's report, no test
    failing, and every generic launch (patch embedding, heads, the whole no-grad forward) 3 x slower.  hipcc cross-compiles without a GPU; the
    units compile in parallel (~75 s).  Allowed: the six attention instantiations whose 2-10 spilled registers were measured faster than their
    spill-free alternatives (profiles/r06_exp_attention_spills.log), none of them on the metric's path."""
    import concurrent.futures as cf
    import subprocess

    from clibd_amd import build

    allowed = {   # mangled-name fragment -> scratch bytes per lane it may use
        "attention_fwd_persistent_kernelILi14ELb1ELb1E": 20, "attention_fwd_persistent_kernelILi16ELb1ELb1E": 44, "attention_fwd_persistent_kernelILi16ELb0ELb1E": 12,
        "attention_bwd_kernelILi10ELb0ELb0ELb1ELi4ELi160E": 28, "attention_bwd_kernelILi10ELb0ELb0ELb0ELi4ELi160E": 28, "attention_bwd_kernelILi16ELb1ELb0ELb1ELi4ELi256E": 44,
    }

    def report(unit):
        r = subprocess.run([build._hipcc(), f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-c",
                            str(build.CSRC / f"{unit}.hip"), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out, name = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+(Function Name|ScratchSize \[bytes/lane\]):\s*(\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                name = m.group(2)
            elif name:
                out.append((name, int(m.group(2))))
        return out

    units = ["gemm256", "attention", "layernorm", "lora", "gemm", "gemm256_tn", "loss"]
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        rows = [r for res in ex.map(report, units) for r in res]
    assert sum(1 for n, _ in rows if "gemm256" in n) >= 40 and len(rows) >= 120, len(rows)
    bad = []
    for name, scratch in rows:
        cap = max([v for k, v in allowed.items() if k in name] or [0])
        if scratch > cap:
            bad.append((name, scratch, cap))
    assert not bad, bad
