"""bench.py's launch contract: `--gpus N` must yield N ranks (self-launch when no launcher set WORLD_SIZE), a
mismatch must fail loudly, and the JSON line must keep the two scan modes apart."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "QADC_WGQ", "QADC_HEAD_LEVEL", "QADC_TEST_HOOKS"):
        env.pop(k, None)
    env.update({k: str(v) for k, v in kw.items()})
    return env


def test_world_size_mismatch_fails_before_touching_the_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(WORLD_SIZE=1),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0
    assert b"--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_self_launch_builds_one_rank_per_gpu(monkeypatch):
    """No WORLD_SIZE and --gpus 4: the parent starts torch.distributed.run with 4 ranks as a child process and relays
    rank 0's line (checked with a fake child; no GPU here)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, stderr=None):
            seen["cmd"] = cmd
            self.stdout = [b"some log line\n", b'{"metric": "pq_codes_scanned_per_sec", "n_gpus": 4}\n']

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    bench.main()
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--gpus") + 1] == "4" and cmd[cmd.index("--steps") + 1] == "3"


def test_bench_prepares_the_stream_set_before_any_communicator():
    """A process that creates a communicator BEFORE the library's stream set runs a rank's IVF batch 14-60 % slower (DESIGN.md
    section 5): in bench.py's main() the qadc_device_prepare call must come before init_process_group and before any rank-side
    index is created — checked on the source, no GPU needed."""
    import ast
    src = open(BENCH).read()
    main = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main")
    calls = {}
    for node in ast.walk(main):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
            calls.setdefault(node.func.attr, []).append(node.lineno)
    assert "device_prepare" in calls and "init_process_group" in calls and "Index" in calls
    assert min(calls["device_prepare"]) < min(calls["init_process_group"])
    assert min(calls["device_prepare"]) < min(calls["Index"])


@pytest.mark.gpu
def test_a_hung_multi_rank_leg_is_abandoned_with_a_nonzero_exit():
    """The N-rank IVF legs sit under a watchdog on every rank: when they do not finish in time (here: 0.2 s for legs that take
    seconds) rank 0 prints the headline with the error in `ivf` and every rank leaves with exit code 3 — the launcher returns
    non-zero instead of hanging."""
    env = _env(QADC_BENCH_BACKEND="gloo", QADC_BENCH_ONE_GPU=1, QADC_BENCH_IVF_TIMEOUT=0.2,
               **dict(SMALL, QADC_BENCH_IVF_CODES=int(4e6), QADC_BENCH_CODES=int(1e7)))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "abandoned" in line["ivf"]["error"]


SMALL = dict(QADC_BENCH_CODES=int(4e7), QADC_BENCH_CPU_SECONDS=0, QADC_BENCH_REAL_CODES=0, QADC_BENCH_IVF_CODES=0,
             QADC_BENCH_32X4=0, QADC_BENCH_LATENCY=0, QADC_BENCH_PMC=0)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,native", [(2, 1), (4, 1), (8, 1), (2, 0)])
def test_gpus_n_self_launch_runs_the_multi_rank_loop_on_one_gpu(ranks, native):
    """bench.py --gpus N end to end with N ranks on ONE GPU (gloo for the script's own barriers): the flat multi-rank loop —
    sliced pre-scan, one gather per step carrying streams + the next pre-scan values, host-share replay, second gather —
    through the library's NATIVE merge over the shared-memory transport (what an 8-GPU node runs over RCCL), and through
    the older torch.distributed path."""
    env = _env(QADC_BENCH_BACKEND="gloo", QADC_BENCH_ONE_GPU=1, QADC_BENCH_NATIVE_DIST=native, **SMALL)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(ranks), "--steps", "6", "--warmup", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["rccl_ranks"] == ranks
    assert line["multi_gpu_merge"].startswith("native" if native else "pyqadc/sharded.py")
    assert line["recall_at_100"] == 1.0
    # the N-rank line carries the metric's mode too: rank 0's shard, one query per pass, priced against HBM (no PMC child at N > 1)
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and rf["traffic"] is None
    assert rf["shard"] == {"rank": 0, "of": ranks, "codes": rf["shard"]["codes"]} and abs(rf["shard"]["codes"] - 4e7 / ranks) <= 64
    assert line["config"]["mode"] == "one query per pass"
    assert line["value_batched"] > 0 and line["roofline_batched"]["bound"] == "lds"


@pytest.mark.gpu
@pytest.mark.parametrize("placement", ["range", "whole"])
def test_gpus_2_ivf_legs_through_the_native_merge(placement):
    """The N-rank line carries `ivf` (BASELINE configs[2] shape, shrunk) measured through qadc_search_submit +
    qadc_dist_collect on every rank — here 2 ranks on one GPU over the shared-memory transport (RCCL needs a GPU per rank)."""
    env = _env(QADC_BENCH_BACKEND="gloo", QADC_BENCH_ONE_GPU=1, QADC_BENCH_IVF_PLACEMENT=placement,
               **dict(SMALL, QADC_BENCH_IVF_CODES=int(4e6), QADC_BENCH_CODES=int(1e7)))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    ivf = line["ivf"]
    assert ivf["rccl_ranks"] == 2 and ivf["placement"] == placement and ivf["us_per_query"] > 0
    assert ivf["batches_through_partition_major_second_phase"] > 0 and ivf["of_them_redone_on_the_level_path"] == 0
    assert "shared-memory" in ivf["merge"]


@pytest.mark.gpu
def test_single_gpu_line_keeps_the_two_modes_apart():
    env = _env(**dict(SMALL, QADC_BENCH_PMC=1, QADC_BENCH_IVF_CODES=int(4e6), QADC_BENCH_LATENCY=1, QADC_BENCH_32X4=1,
                      QADC_BENCH_SINGLE_QUERIES=8))
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and "scan_i8_kernel<16,2,nt,chunk>" in rf["kernel"]
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    # the headline triple comes from ONE timed region in the metric's mode: value = codes x queries / wall, the roofline's
    # region is those same steps, and the algorithmic bytes of a step / ms_per_step stay under the HBM peak
    assert line["config"]["mode"] == "one query per pass" and line["config"]["queries_per_step"] == 32
    assert rf["region_queries"] == 32 * 3 and rf["region_wall_s"] == pytest.approx(line["ms_per_step"] * 3e-3, rel=1e-6)
    assert line["value"] == pytest.approx(4e7 * 32 * 3 / rf["region_wall_s"], rel=1e-6)
    assert 4e7 * 8 * 32 / (line["ms_per_step"] * 1e-3) / 1e9 <= rf["peak"]
    assert rf["region_wall_s"] >= rf["launches"] * rf["avg_launch_ms"] * 1e-3 / 3     # (three steps in flight at most)
    # the batched mode is a separate figure with its own roof; so is the one-query-per-CALL leg
    assert line["roofline_batched"]["bound"] == "lds" and 0 < line["roofline_batched"]["frac"] <= 1.0
    assert line["value_batched"] > 0 and line["ms_per_step_batched"] > 0
    rc = line["roofline_one_query_per_call"]
    assert rc["bound"] == "hbm" and 0 < rc["frac"] <= 1.0 and rc["region_queries"] == 8
    assert 0 < line["roofline_32x4"]["frac"] <= 1.0
    assert line["ivf"]["us_per_query"] > 0 and line["latency_us_single_query"]["value"] > 0
    ri = line["roofline_ivf"]                                  # the IVF leg's own roofs: grouped scan vs LDS, head vs HBM
    assert ri["bound"] == "lds" and 0 < ri["frac"] <= 1.0 and 0 < ri["seat_fill"] <= 1.0
    assert ri["head"]["bound"] == "hbm" and 0 < ri["head"]["frac"] <= 1.0
    assert line["n_gpus"] == 1 and line["recall_at_100"] == 1.0
    # in-run PMC traffic: present when rocprofv3 exists on the box, and then close to the algorithmic bytes
    if rf["traffic"] is not None:
        assert 0.5 < rf["traffic_over_algorithmic"] < 1.5


@pytest.mark.gpu
@pytest.mark.parametrize("native", [1, 0])
def test_multi_rank_loop_with_one_rank_over_rccl(native):
    """bench.py's multi-rank loop (sliced pre-scan, one gather per step carrying streams + next pre-scan values) with
    the one RCCL rank a 1-GPU box allows: through the library's native merge (qadc_dist_collect) and through the
    torch.distributed path; recall 1.0 either way."""
    env = _env(QADC_BENCH_FORCE_DIST=1, QADC_BENCH_NATIVE_DIST=native, **SMALL)
    r = subprocess.run([sys.executable, BENCH, "--steps", "6", "--warmup", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["recall_at_100"] == 1.0 and line["rccl_ranks"] == 1
    assert line["multi_gpu_merge"].startswith("native" if native else "pyqadc/sharded.py")
