"""bench.py's N > 1 control flow on the 1-GPU box (`-m gpu`): two ranks launched exactly as the driver launches
them (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 …`), sharing device 0 over gloo
(`--dist-backend gloo --device 0`; RCCL needs one GPU per rank).  Every rank renders its own frame, the step's one
exchange (stand-in parameter gradients + camera gradient in one flat buffer) runs in both modes, the timing
reductions and barriers execute, and rank 0 prints the contract line with `n_gpus == 2` and the `multi_gpu` legs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["overlap", "serial"])
def test_bench_two_ranks_share_one_gpu(mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--config", "C2", "--dist-backend", "gloo", "--device", "0", "--grad-buffer-floats", "300000",
           "--exchange-mode", mode, "--profile-steps", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["parallelism"] == "frames x2"
    mg = rec["multi_gpu"]
    assert mg["timed_loop_mode"] == mode
    for k in ("raster_ms", "allreduce_ms", "serial_ms_per_step", "overlapped_ms_per_step"):
        assert mg[k] > 0, k
    assert "cpu_baseline" not in rec and "secondary" not in rec   # rank 0 at N = 1 only
    # what tells a path-bound step from an exchange-bound one sits at the TOP level of the record
    assert rec["raster_only_mpix_s"] == mg["raster_only_mpix_s"] > 0 and rec["allreduce_ms"] == mg["allreduce_ms"]
    assert rec["allreduce_busbw_GBps"] >= 0 and rec["allreduce_over_raster"] == pytest.approx(mg["allreduce_ms"] / mg["raster_ms"], rel=1e-2)
    assert rec["exchange_bound"] == (mg["allreduce_ms"] > mg["raster_ms"])
    # … and every rank reported its device (here both ranks share GPU 0 on purpose: `--device 0`)
    assert [r["rank"] for r in rec["ranks"]] == [0, 1] and all(r["index"] == 0 for r in rec["ranks"])
    # the N > 1 record says what its figures mean before anybody computes an efficiency from `value`
    assert rec["scaling_basis"] == "raster_only_mpix_s" and "value_note" in rec
    assert rec["ranks_on_distinct_gpus"] is False           # (this dry run shares one GPU, and the record says so)
    sweep0 = [e for e in mg["grad_buffer_sweep"] if e["floats"] == 0][0]
    assert rec["value_raster_plus_camera_grad"] == sweep0["mpix_s"] > 0
    assert rec["ms_per_step_raster_plus_camera_grad"] == sweep0["overlapped_ms_per_step"]


@pytest.mark.timeout(900)
def test_bench_one_rank_over_rccl():
    """VERDICT r2 missing #4: the RCCL branch itself (`backend="nccl"`, `device_id=`, `ReduceOp.AVG`, the chunked
    asynchronous exchange) executed on the 1-GPU box: one rank under torch.distributed.run with a forced process group."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--config", "C2", "--force-dist", "--grad-buffer-floats", "1000000", "--profile-steps", "1",
           "--no-cpu-baseline", "--no-secondary", "--no-callsite", "--no-graph"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    mg = rec["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["world"] == 1 and mg["chunks"] == 8
    assert mg["allreduce_ms"] > 0 and mg["overlapped_ms_per_step"] > 0 and mg["serial_ms_per_step"] > 0
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["ranks"][0]["rank"] == 0 and rec["ranks"][0]["index"] == 0 and rec["allreduce_over_raster"] > 0
    # the record says how the device was brought to its sustained power state and carries the from-idle figure beside it
    ds = rec["device_state"]
    assert ds["prewarm_ms"] >= 60 and ds["prewarm_steps"] >= 1 and rec["warmup"] == 1
    assert ds["from_idle"]["steps"] == 2 and ds["from_idle"]["ms_per_step"] > 0


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE / RANK unset — the form the driver used
    for N = 1) starts its two ranks itself and prints ONE contract line with n_gpus == 2; the record carries the
    grad-buffer sweep that shows where the step turns exchange-bound."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--steps",
           "2", "--warmup", "1", "--config", "C2", "--grad-buffer-floats", "300000", "--profile-steps", "1"]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and [r["rank"] for r in rec["ranks"]] == [0, 1]
    sweep = rec["multi_gpu"]["grad_buffer_sweep"]
    assert [e["floats"] for e in sweep] == [0, 300000] and all(e["overlapped_ms_per_step"] > 0 for e in sweep)
