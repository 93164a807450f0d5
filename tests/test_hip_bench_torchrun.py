"""GPU: the exact command the driver uses for the multi-GPU scaling bench -- ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`` -- end to end with two
ranks.  A dev box has ONE GPU, so both ranks share cuda:0 and exchange over gloo (MCVC_DIST_BACKEND=gloo; RCCL refuses
duplicate devices): what is exercised is bench.py's own choreography (env rendezvous, barriers, MAX-reduce of the timed
region, ONE JSON line from rank 0 only, clean process-group teardown), not the transport."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_two_ranks():
    env = dict(os.environ, MCVC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["warmup"] == 2 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["losses_finite"] and res["value"] > 0
    assert abs(res["value"] - 2 * 1e3 / res["ms_per_step"]) < 1e-6 * res["value"]        # whole-job rate = ranks x per-rank rate
    assert "cpu_baseline" not in res and "configs" not in res   # N=1 only
    assert res["dist"]["backend"] == "gloo" and res["dist"]["world"] == 2 and res["dist"]["rccl_ranks_seen"] == 2
    assert "degraded" not in res                   # (ranks sharing a GPU switch the persistent trunk off on purpose: not a degradation)


def test_multi_gpu_bench_fails_when_the_ranks_are_silently_degraded():
    """a SCALE line must not be quietly degraded.  Here the persistent trunk kernels are forced off their residency
    bound on both ranks (bench.py --test-force-residency 9: the engine claims 9 passes in flight = 576 workgroups > 256 compute units, what five
    free-running grouped passes did on data-parallel ranks after the identity cut-off before r5) while the ranks do NOT share a GPU as
    far as the job can tell (--test-distinct-gpus skips the shared-device switch): bench.py must print its line, say why, and exit
    non-zero; --allow-degraded turns that into a zero exit."""
    env = dict(os.environ, MCVC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-trace",
            "--test-force-residency", "9", "--test-distinct-gpus"]
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(lines[0])
    assert res["degraded"] and "per-layer" in res["degraded"][0] and res["schedule"]["trunk_persistent_possible"] and not res["schedule"]["trunk_persistent"]
    r = subprocess.run(base + ["--allow-degraded"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


def test_default_single_gpu_bench_carries_every_baseline_config():
    """The invocation the driver runs (`python bench.py --gpus 1 ...`) must measure BASELINE configs[2], the per-GPU shape of configs[3]
    and configs[4] besides the bs=1 headline, each as a nested record with its own roofline (CPU samples skipped here for time)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "3", "--cpu-iters", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["config"]["global_batch"] == 1 and res["n_gpus"] == 1 and res["steps"] == 5 and res["losses_finite"]
    assert res["roofline"]["frac"] > 0 and res["dist"] == {"backend": None, "world": 1, "rccl_ranks_seen": 1}
    ids = [c["config_id"] for c in res["configs"]]
    assert len(ids) == 3 and ids[0].startswith("configs[2]") and ids[1].startswith("configs[3]") and ids[2].startswith("configs[4]")
    assert [c["config"]["global_batch"] for c in res["configs"]] == [32, 8, 16]
    for c in res["configs"]:
        assert c["value"] > 0 and c["ms_per_step"] > 0 and 0 < c["roofline"]["frac"] < 1 and c["roofline"]["kernel"]
    assert res["configs"][2]["dtype"] == "bf16" and res["configs"][2]["outputs_finite"]
    # the two byte counts are named for what they are, and the reference-exact loss readback is priced in the same run
    assert res["hbm_bytes_per_step_launcher"] > 0 and "hbm_bytes_per_step_pmc" in res and "hbm_bytes_per_step" not in res
    sc = res["schedule"]
    assert sc["sync_losses_ms_per_step"] > 0 and -0.2 < sc["sync_losses_cost"] < 1.0 and "two step()s earlier" in sc["loss_readback"]
    # ... and so is the regime after the identity cut-off (train.py:314-315), labelled, beside the headline (which stays the dearer regime)
    post = res["after_identity_cutoff"]
    assert post["identity_loss_lambda"] == 0 and 0 < post["ms_per_step"] < 1.1 * res["ms_per_step"] and res["identity_loss_lambda"] == 5


def test_nccl_backend_refuses_fewer_gpus_than_ranks():
    import torch
    code = ("import os, sys; sys.path.insert(0, %r); "
            "os.environ.update(RANK='0', WORLD_SIZE='%d', LOCAL_RANK='0', LOCAL_WORLD_SIZE='%d', MASTER_ADDR='127.0.0.1', MASTER_PORT='29547'); "
            "from mask_cyclegan_vc.parallel import init_from_env; init_from_env(backend='nccl')")
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "-c", code % (os.path.join(ROOT, "maskcyclegan-vc_amd"), n, n)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "one GPU per local rank" in r.stderr
