"""bench.py's launch convention on CPU: `python bench.py --gpus N` with no launcher around it starts its own N ranks through
`torch.distributed.run` (the contract's external form, same arguments); under a launcher (RANK / WORLD_SIZE set) it never
re-launches.  The launched path itself runs on the GPU box (tests/test_gpu_bench_two_ranks.py)."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_gpus_without_launcher_starts_ranks(monkeypatch):
    bench = _bench()
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: (calls.append((cmd, env)), 7)[1])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                      # the launcher's exit code is handed back
    (cmd, env), = calls
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_under_a_launcher_no_relaunch(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("re-launched under a launcher"))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    # WORLD_SIZE = 1 under a launcher but --gpus 2: the old, explicit error — not a silent re-launch, not a wrong n_gpus
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        bench.main()


def test_profile_selection_prefers_the_stamp_of_the_current_sources(tmp_path, monkeypatch):
    """the PMC-derived fields of the record must not come from a stale profile while a fresh sibling
    exists — a profile whose `<tag>_meta.json` carries the library's current source hash wins over a 'newer' name; profiles of
    another config are never picked for C3."""
    import json
    bench = _bench()
    from ggrt_official_amd import _build
    prof = tmp_path / "profiles"
    prof.mkdir()
    now = _build.source_hash()
    for tag, meta in (("r09_v1", dict(source_hash=now, config="C3")), ("r09_v2", dict(source_hash="0" * 64, config="C3")),
                      ("r09_zz", dict(source_hash=now, config="C5p"))):
        (prof / f"{tag}_pmc_sq.json").write_text("{}")
        (prof / f"{tag}_meta.json").write_text(json.dumps(meta))
    (prof / "r01_v1_pmc_sq.json").write_text("{}")           # before the stamps: counts as C3, never as fresh
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert os.path.basename(bench.newest_profile("r*_pmc_sq.json", "C3")) == "r09_v1_pmc_sq.json"
    (prof / "r09_v1_meta.json").write_text(json.dumps(dict(source_hash="1" * 64, config="C3")))
    assert os.path.basename(bench.newest_profile("r*_pmc_sq.json", "C3")) == "r09_v2_pmc_sq.json"   # newest C3 by name
    assert os.path.basename(bench.newest_profile("r*_pmc_sq.json")) == "r09_zz_pmc_sq.json"          # no config: by name


def test_measured_quantities_come_from_profile_files():
    """bench.py quotes no measured quantity as a literal — the copy ceilings, the v_fma rate, the blend
    kernels' slot counts and the rocprof kernel averages are read from the committed files they were measured into."""
    import re
    import bench
    mc = bench.measured_constants()
    assert 100.0 < mc["fp32_fma_tflops"]["value"] < 160.0 and mc["fp32_fma_tflops"]["source"].endswith("r03_valu_peak.txt")
    assert 4000.0 < mc["copy_8in_8out_worst_GBps"]["value"] <= mc["copy_8in_8out_best_GBps"]["value"] < 8000.0
    row, src = bench.blend_slot_counts("C3")
    assert row["bwd_slots"] > 3_000_000 and row["bwd_valid_pairs"] == row["fwd_pairs_composited"] and "blend_slot_counts" in src
    assert bench.blend_slot_counts("no-such-config")[0] is None
    us, calls, stamp = bench.rocprof_kernel_us("C3", "blend_bwd_kernel")
    assert 250.0 < us < 500.0 and calls > 50 and stamp["file"].startswith("profiles/r") and "stale_profile" in stamp
    ev = bench.blend_evaluated_pairs("C3", {"fwd_blend_ms": 0.15, "bwd_blend_ms": 0.35}, 10_763_014)
    assert ev["bwd"]["slots"] == row["bwd_slots"] and 0.5 < ev["bwd"]["lane_utilisation"] < 0.65
    # … and the source holds none of the figures that used to be typed in
    src_text = open(bench.__file__).read()
    for literal in ("4885", "5680", "126.7 ", "37.0", "1.867e9", "3640000"):
        assert not re.search(r"(?<![\w.])" + re.escape(literal.strip()) + r"(?![\w])", re.sub(r"#.*", "", src_text)) or literal == "126.7 ", literal


def test_byte_model_of_the_built_kernels():
    """preprocess_bwd no longer reads the SH rows — its moved bytes are 156 in + (52 + 12 M) out per Gaussian."""
    import bench
    ab = bench.algorithmic_bytes(1_000_000, 10_000_000, 1920, 1080, 16, 16, 8_000_000)
    assert ab["bwd_preprocess"] == 1_000_000 * (40 + 192) + 1_000_000 * (52 + 192)
    assert ab["bwd_preprocess_moved"] == 1_000_000 * 156 + 1_000_000 * 244 == 400_000_000
    assert ab["tile_sort_compulsory"] == 8_000_000 * 12 and ab["tile_scatter_pairs_compulsory"] == 8_000_000 * 8 + 12_000_000
