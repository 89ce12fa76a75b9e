"""R-independence on the real kernels: what DESIGN.md section 6 promises -- "results do not depend
on the number of GPUs" -- checked with two ranks running the HIP engine (two fresh child
processes on the one GPU of the box, gloo; tests/dist_gpu_worker.py) against one rank playing
all the games; and bench.py's own launcher (`--gpus N` without torch.distributed.run).
The reference's analogue of the exchange: list concatenation, src/train_rl.py:48-51."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def run_world(world, *args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, *[str(a) for a in args]], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        for p in procs:
            out, err = p.communicate(timeout=900)
            assert p.returncode == 0, err[-3000:]
    finally:
        # a rank that failed (or a timeout) leaves the others blocked in a gloo collective, holding the
        # GPU: end every rank that is still running, by PID
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait(timeout=60)


def canonical(npz, keys, by):
    order = np.lexsort(tuple(npz[k] for k in reversed(by)))
    return {k: npz[k][order] for k in keys}


def test_selfplay_two_ranks_equal_one_rank(tmp_path):
    """SelfPlayEngine (production defaults: hipGraph, look-ahead, value cache) on 2 x 32 games vs
    1 x 64: the gathered (s, pi, z) tuples are identical row for row -- Philox streams are keyed
    by the global game id, a net's output for a board does not depend on its batch."""
    one, two = str(tmp_path / "w1.npz"), str(tmp_path / "w2.npz")
    run_world(1, "selfplay", 64, 24, one)
    run_world(2, "selfplay", 64, 24, two)
    a, b = np.load(one), np.load(two)
    keys = ("own", "opp", "pi", "z", "move", "colour", "game", "turn")
    ca, cb = canonical(a, keys, ("game", "turn")), canonical(b, keys, ("game", "turn"))
    assert len(ca["z"]) > 64 * 40
    for k in keys:
        assert np.array_equal(ca[k], cb[k]), k
    # rank 0's rows first, then rank 1's (gather_tuples): games 0..31 before 32..63
    first_hi = int(np.argmax(b["game"] >= 32))
    assert np.all(b["game"][:first_hi] < 32) and np.all(b["game"][first_hi:] >= 32)
    assert np.array_equal(a["final_z"][np.argsort(a["final_game"])], b["final_z"][np.argsort(b["final_game"])])
    assert int(a["leaf_evals"]) > 0


def test_selfplay_five_ranks_equal_one_rank(tmp_path):
    """The same with FIVE ranks on the one GPU (this pool lets six processes of a job use the GPU at once, the
    test runner being one of them; the driver's node has 8 GPUs): 5 x 13 games = 1 x 65 games row for row, a
    five-way ragged gather."""
    one, six = str(tmp_path / "w1.npz"), str(tmp_path / "w5.npz")
    run_world(1, "selfplay", 65, 24, one)
    run_world(5, "selfplay", 65, 24, six)
    a, b = np.load(one), np.load(six)
    keys = ("own", "opp", "pi", "z", "move", "colour", "game", "turn")
    ca, cb = canonical(a, keys, ("game", "turn")), canonical(b, keys, ("game", "turn"))
    assert len(ca["z"]) > 65 * 40
    for k in keys:
        assert np.array_equal(ca[k], cb[k]), k
    # gather_tuples: rank r's rows before rank r + 1's -- the global game ids never decrease across blocks
    blocks = b["game"] // 13
    assert np.all(np.diff(blocks) >= 0) and set(blocks.tolist()) == set(range(5))
    assert np.array_equal(a["final_z"][np.argsort(a["final_game"])], b["final_z"][np.argsort(b["final_game"])])


def test_reinforce_two_ranks_equal_one_rank(tmp_path):
    """ReinforceTrainer.step() x 3 and one step_from_tuples() of PV-MCTS tuples, 2 ranks vs 1:
    same gathered tuples, same win rates, parameters allclose 1e-6, replicas bit-identical after
    the broadcast (the ranks start from DIFFERENT random replicas and seeds: rank 0's win)."""
    run_world(1, "reinforce", 3, tmp_path / "w1")
    run_world(2, "reinforce", 3, tmp_path / "w2")
    one = np.load(str(tmp_path / "w1.rank0.npz"))
    r0, r1 = np.load(str(tmp_path / "w2.rank0.npz")), np.load(str(tmp_path / "w2.rank1.npz"))
    for i in range(3):   # the gathered tuples of every set: the same rows in the same (canonical) order
        for src in (r0, r1):
            for k in ("own", "opp", "action", "z"):
                assert np.array_equal(one["set%d_%s" % (i, k)], src["set%d_%s" % (i, k)]), (i, k)
    assert np.array_equal(one["rates"], r0["rates"]) and np.array_equal(r0["rates"], r1["rates"])
    assert np.array_equal(one["n_tuples"], r0["n_tuples"])
    assert int(one["adam_t"]) == int(r0["adam_t"]) == 4
    assert int(one["mcts_tuples"]) == int(r0["mcts_tuples"]) > 0
    assert np.allclose(one["losses"], r0["losses"], rtol=1e-5, atol=1e-7)
    assert np.allclose(one["mcts_loss"], r0["mcts_loss"], rtol=1e-5, atol=1e-7)
    params = [k for k in one.files if "/" in k]
    assert len(params) == 18
    worst = max(float(np.max(np.abs(one[k] - r0[k]))) for k in params)
    print("max |param(1 rank) - param(2 ranks)| after 4 updates: %.3g" % worst)
    for k in params:
        assert np.array_equal(r0[k], r1[k]), k                       # replicas: bit-identical
        assert np.allclose(one[k], r0[k], rtol=0, atol=1e-6), k      # 2 ranks vs 1


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` as the driver runs it, without a launcher: the spawn path
    (forced with --spawn for N = 1, all this box has) starts the rank as a child process with
    RANK / WORLD_SIZE / MASTER_* set, that rank runs the whole bench over the nccl (= RCCL)
    backend and prints the ONE JSON line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "20",
                          "--warmup", "5", "--rollout-only"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env={k: v for k, v in os.environ.items()
                                           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 1e6
    assert "rccl all-gather" in line["config"]["tuple_allgather"]   # the rank ran under a process group
    # a rank that fails takes the launcher's exit status with it
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--boards", "-5",
                          "--rollout-only"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus 2 started the way the driver starts it (no launcher: it spawns its ranks),
    every leg, with both ranks on the one GPU of the box over gloo (IAGO_BENCH_BACKEND / _DEVICE:
    RCCL refuses two ranks on one device): the N > 1 code of the bench -- the headline's sharded engines
    (global game ids per rank), the all-gather of every batch's tuples inside its step, the max-over-ranks
    timing, the self-diagnosis keys (ranks seen, per-rank play time, "ranks played different games"), the
    nested rollout leg's side-stream gathers, the sharded REINFORCE leg -- runs and prints ONE line with
    whole-job values."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IAGO_BENCH_BACKEND="gloo", IAGO_BENCH_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--rollout-steps", "20", "--rollout-warmup", "5", "--large-boards", "0",
                          "--mcts-games", "128", "--nthr1-turns", "0", "--mcts400-turns", "2", "--train-iters", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["games_per_step"] == 2 * 128 and line["config"]["full_games"] is True
    assert "gloo all-gather" in line["config"]["tuple_allgather"]
    assert abs(line["value"] - 2 * 128 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["value"] == line["mcts_games_per_sec"]
    m = line["mcts"]
    assert m["steps"] == 2 and len(m["step_seconds"]) == 2 and m["gathered_tuples"] > 2 * 128 * 50
    assert line["leaf_evals_per_sec"] == m["leaf_evals_per_sec"] > 0
    assert line["ranks_seen"] == 2 and line["ranks_played_different_games"] is True
    assert m["ranks"]["ranks_reporting"] == [0, 1]
    assert 0 < line["rank_play_seconds_min"] <= line["rank_play_seconds_max"] and line["gather_ms"] > 0
    r = line["rollout"]
    assert r["config"]["games_per_step"] == 2 * 4096 and "gloo all-gather" in r["config"]["tuple_allgather"]
    assert abs(r["value"] - 2 * 4096 * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]
    assert line["reinforce"]["iters"] == 2 and "cpu_baseline" not in line   # CPU baselines are N = 1 figures
    assert line["mcts400_opening"]["leaf_evals"] == 2 * 128 * 400 * 2 and line["leaf_evals_per_sec_400_opening"] > 0
    assert line["reinforce"]["mcts_fed"]["rounds"] == 1 and line["reinforce_miopen_find_db"] in ("cold", "warm")


FIVE = ["--gpus", "5", "--steps", "2", "--warmup", "1", "--boards", "4096", "--rollout-steps", "20", "--rollout-warmup", "5",
        "--mcts-games", "32", "--nthr1-turns", "0", "--mcts400-turns", "0", "--train-iters", "1", "--large-boards", "0"]


def _rehearsal_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IAGO_BENCH_BACKEND="gloo", IAGO_BENCH_DEVICE="0")
    return env


def test_bench_five_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus N at the largest world this pool allows on one box (6 GPU processes, the test
    runner included; the driver's first 8-rank run cannot be debugged): self-spawned ranks, one
    rendezvous, five engines' pools and five MIOpen caches warming at once, five-way ragged gathers, the
    "ranks played different games" assertion five ways -- ONE line with whole-job values."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + FIVE, cwd=ROOT, capture_output=True,
                         text=True, timeout=1100, env=_rehearsal_env())
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 5 and line["scaling"] == "weak"
    assert line["config"]["games_per_step"] == 5 * 32 and "gloo all-gather" in line["config"]["tuple_allgather"]
    assert abs(line["value"] - 5 * 32 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["ranks_seen"] == 5 and line["ranks_played_different_games"] is True
    assert line["mcts"]["ranks"]["ranks_reporting"] == [0, 1, 2, 3, 4]
    assert line["mcts"]["gathered_tuples"] > 5 * 32 * 50         # every rank's rows of the last batch
    assert line["rollout"]["config"]["games_per_step"] == 5 * 4096
    assert line["reinforce"]["iters"] == 1 and line["reinforce"]["mcts_fed"]["rounds"] == 1
    assert line["reinforce"]["mcts_fed"]["tuples_per_round"] > 64 * 40      # all 64 games of the round, gathered
    assert "cpu_baseline" not in line
    # round 6: the check before anything is timed (a 1-element all-reduce: 0 + 1 + 2 + 3 + 4; the device check is skipped
    # in the rehearsal, where every rank sits on the one GPU), the per-rank rates, the REINFORCE set that scales
    assert line["preflight"]["allreduce_sum_of_ranks"] == 10 and "skipped" in line["preflight"]["device_check"]
    assert 0 < line["per_rank_games_per_sec_min"] <= line["per_rank_games_per_sec_max"]
    ws = line["reinforce"]["weak_scaling_set"]
    assert ws["games_per_set"] == 5 * 64 and ws["tuples_per_iter"] > 5 * 64 * 20 and ws["gather_ms"] > 0.0


def test_bench_launcher_propagates_a_killed_rank():
    """One of five running ranks is killed (SIGKILL, by PID): the launcher ends the other four -- they
    would wait in a collective forever -- and exits non-zero with nothing on stdout."""
    import signal
    import time
    # (a run long enough to be in full swing when the rank is killed: whole games, a long training leg)
    long_run = [x for x in FIVE]
    long_run[long_run.index("--steps") + 1] = "40"
    long_run[long_run.index("--mcts-games") + 1] = "256"
    long_run[long_run.index("--train-iters") + 1] = "400"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + long_run, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=_rehearsal_env())
    try:
        kids = []
        for _ in range(600):        # the ranks are this launcher's direct children
            time.sleep(0.1)
            kids = []
            for d in os.listdir("/proc"):
                if d.isdigit():
                    try:
                        with open("/proc/%s/stat" % d) as f:
                            if int(f.read().rsplit(")", 1)[1].split()[1]) == p.pid:
                                kids.append(int(d))
                    except (OSError, ValueError, IndexError):
                        pass
            if len(kids) == 5 or p.poll() is not None:
                break
        assert len(kids) == 5 and p.poll() is None
        time.sleep(12.0)            # let them get into the run (import, rendezvous, first launches)
        assert p.poll() is None
        os.kill(sorted(kids)[3], signal.SIGKILL)
        out, err = p.communicate(timeout=120)
        assert p.returncode != 0
        assert out.strip() == ""
        time.sleep(1.0)
        assert not [k for k in kids if os.path.exists("/proc/%d" % k)]     # no rank left behind
    finally:
        if p.poll() is None:
            p.kill()
