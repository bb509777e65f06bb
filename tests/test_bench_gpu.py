"""bench.py as the driver starts it: ``python bench.py --gpus N`` launches N ranks by itself (the reference's
tools/dist_test.sh:9-10 -> tools/test.py:101-208 shape).  On a one-GPU box the ranks share the GPU through the
DMB_BENCH_BACKEND=gloo hook (RCCL refuses duplicate devices); everything else -- the sharding pair i -> rank i mod
world, the barrier-fenced timing, the MAX-over-ranks clock, the single SUM all-reduce of the EPE accumulator -- is the
code path of the N-GPU job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None, want_text=False):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras"] + list(flags),
                         capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # ONE JSON line, from rank 0
    if want_text:
        return json.loads(lines[0]), out.stdout + out.stderr
    return json.loads(lines[0])


def test_bench_gpus_flag_launches_the_ranks(dev):
    one = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2")
    two = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", env={"DMB_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["pairs"] == 4 and two["pairs"] == 4          # B * steps * world
    assert two["config"]["pairs_per_step_per_gpu"] == 1 and two["scaling"] == "weak"
    # same four (pair, step) evaluations -- pairs 0 and 1, twice -- however they are sharded: the dataset metrics agree
    for k in ("epe", "1px", "3px"):
        assert abs(one["epe_accumulator"][k] - two["epe_accumulator"][k]) <= 1e-9 * max(1.0, abs(one["epe_accumulator"][k]))
    assert two["value"] > 0 and two["ms_per_step"] > 0
    # every rank's own clock is in the line, the metric uses the slowest (MAX over ranks)
    assert len(two["per_rank_ms"]) == 2 and len(one["per_rank_ms"]) == 1
    assert abs(max(two["per_rank_ms"]) - two["ms_per_step"]) <= 1e-3 * two["ms_per_step"] + 2e-3
    assert "pinned" in two["host_affinity"]


def test_bench_one_rank_through_rccl(dev):
    """The exchange of tools/test.py:172-208 on the hardware that is there: a ONE-rank job forced through an ``nccl`` (= RCCL)
    process group with ``device_id`` -- RCCL initialisation, the FP64 accumulator all-reduce, the barrier fences and the MAX
    clock reduce all execute on the MI355X.  The RCCL banner (NCCL_DEBUG=VERSION) is captured as evidence."""
    plain = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2")
    rccl, text = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", want_text=True,
                        env={"DMB_BENCH_FORCE_PG": "1", "NCCL_DEBUG": "VERSION", "MASTER_ADDR": "127.0.0.1"})
    banner = [ln for ln in text.splitlines() if "NCCL version" in ln or "RCCL version" in ln]
    print("\n".join(banner[:3]))
    assert banner, "no RCCL version banner in the job's output:\n" + text[-2000:]
    assert rccl["n_gpus"] == 1 and rccl["pairs"] == plain["pairs"] == 4
    for k in ("epe", "1px", "3px"):   # the all-reduced accumulator of a one-rank group is the rank's own
        assert rccl["epe_accumulator"][k] == plain["epe_accumulator"][k]
    assert rccl["value"] > 0
    assert rccl.get("rccl_version") and "rccl_version" not in plain     # e.g. "2.26.6": read from the library the job ran on
