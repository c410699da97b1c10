"""bench.py's N-rank flow, run as the driver runs it.  A file of its own, first in the session: its children are the only processes on the GPU then."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_runs_its_two_rank_flow_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on THIS box's one GPU (--shared-gpu: gloo for the process
    group, the peer-write composer between the processes): the N-rank control flow of the file -- both legs' frames, the balancing rounds with their all-gathers,
    brmi_set_band / brmi_compose_set_bounds, composition of every frame, the max-over-ranks reduction, rank 0's ONE line -- runs end to end and the line has the
    contract's fields.  Its numbers are two processes sharing a GPU; nothing is asserted about them beyond being there."""
    import json
    import socket
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BRMI_BENCH_PEER_TIMEOUT_MS="10000")
    # Two processes' device-side waits for each other need both processes' queues ON the one GPU at the same time; when the scheduler time-slices them instead (seen once
    # in a session whose parent process held queues of 200 earlier tests) a wait runs into its timeout and bench.py refuses the run -- an artefact of sharing the GPU,
    # so the check is repeated rather than failed on that one message.
    for attempt in range(3):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--steps", "4", "--warmup", "3", "--balance-rounds", "2", "--balance-frames", "6"]
        done = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        text = done.stdout.decode(errors="replace")
        if done.returncode == 0 or "a wait for a peer's band" not in text:
            break
    assert done.returncode == 0, text[-4000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-4000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 3 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["metric"].startswith("shaded Mpixels/s") and out["unit"] == "Mpixels/s" and out["value"] > 0 and out["ms_per_step"] > 0
    assert "shared_gpu" in out
    assert out["config"]["baseline_config"].startswith("configs[2]")      # the line's own value: the headline scene, weak-scaled
    for leg in ("weak", "configs3_weak", "configs3_strong"):
        assert out[leg]["rank_ms_per_step"]["ranks"] == 2 and out[leg]["value"] > 0 and out[leg]["n1_reference"]["value"] > 0 and 0 < out[leg]["efficiency_vs_n1"]
    assert out["config"]["partition"].startswith("cost-balanced contiguous bands")


@pytest.mark.gpu
def test_bench_falls_back_to_peer_writes_when_rccl_refuses():
    """The RCCL composer of the N-GPU bench has never had more than one rank on this pool.  Should it fail on the driver's node, every rank must learn so together and the run
    must go on: here RCCL does refuse (two ranks on ONE device: 'Duplicate GPU detected'), the ranks agree, the peer-write composer takes over, and the line says what happened."""
    import json
    import socket
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BRMI_BENCH_PEER_TIMEOUT_MS="10000", BRMI_BENCH_KEEP_COMPOSER="1")
    for attempt in range(3):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--workload", "sponza", "--legs", "weak", "--steps", "3", "--warmup", "2", "--balance-rounds", "1", "--balance-frames", "4"]
        done = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        text = done.stdout.decode(errors="replace")
        if done.returncode == 0 or "a wait for a peer's band" not in text:
            break
    assert done.returncode == 0, text[-4000:]
    out = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "native composer failed" in out["config"]["workload"] and "peer" in out["config"]["workload"], out["config"]["workload"]
