"""The N>1 path on CPU: world_size-2 gloo processes shard the games and
all-gather their finished tuples (iago_amd.dist), as bench.py / the self-play
engine do over RCCL on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iago_amd.dist import broadcast_object, broadcast_tensors, gather_tuples, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_games():
    for n, world in [(8192, 8), (10, 3), (5, 8), (0, 2), (4096, 1)]:
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_games = 11
        lo, hi = shard_range(n_games)
        # ragged: every game contributes (game id + 1) % 4 tuples
        rows = [(g, k) for g in range(lo, hi) for k in range((g + 1) % 4)]
        own = torch.tensor([g * 1000 + k for g, k in rows], dtype=torch.int64)
        pi = torch.tensor([[g + k + c for c in range(64)] for g, k in rows],
                          dtype=torch.int32).reshape(-1, 64)
        z = torch.tensor([(g % 3) - 1 for g, k in rows], dtype=torch.int8)
        out = gather_tuples(dict(own=own, pi=pi, z=z))
        q.put((rank, out["own"].tolist(), out["pi"].sum().item(), out["z"].tolist(),
               tuple(out["pi"].shape)))
        # an empty shard on one rank must work too
        e = gather_tuples(dict(x=torch.arange(3 if rank == 0 else 0, dtype=torch.float32)))
        q.put((rank, e["x"].tolist()))
        # replicas of a model take rank 0's parameters with one collective (train_rl.py)
        a = torch.full((3, 2), float(rank + 1))
        b = torch.full((5,), rank + 7, dtype=torch.int64)
        broadcast_tensors([a, b])
        assert torch.equal(a, torch.full((3, 2), 1.0)) and torch.equal(b, torch.full((5,), 7))
        assert broadcast_object(100 + rank) == 100
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_tuples_gloo(world):
    """world 8 = the driver's node: 11 games over 8 ranks (shards of 1 and 2 games, some of them
    without a tuple), an 8-way ragged gather, empty shards on 7 ranks, the replica broadcast."""
    port = 29000 + (os.getpid() + 7 * world) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = [(g, k) for g in range(11) for k in range((g + 1) % 4)]
    want_own = [g * 1000 + k for g, k in rows]
    want_z = [(g % 3) - 1 for g, k in rows]
    full = [x for x in got if len(x) == 5]
    assert len(full) == world
    for rank, own, pisum, z, shape in full:
        assert own == want_own and z == want_z
        assert shape == (len(rows), 64)
        assert pisum == sum(g + k + c for g, k in rows for c in range(64))
    for x in got:
        if len(x) == 2:
            assert x[1] == [0.0, 1.0, 2.0]


def test_gather_tuples_single_process_is_identity():
    t = dict(a=torch.arange(5), b=torch.ones(5, 3))
    out = gather_tuples(t)
    assert out["a"] is t["a"] and out["b"] is t["b"]


def test_bench_launcher_reports_a_failing_rank():
    """`python bench.py --gpus 2` without a launcher starts its ranks as child processes and exits
    with their status: on this GPU-less container every rank fails at its first HIP call, and the
    launcher must come back promptly with a non-zero status and nothing on stdout."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU (the -m gpu tests run the launcher for real)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rollout-only"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert out.stdout.strip() == ""


def test_bench_refuses_thin_kernel_profiles(tmp_path, monkeypatch):
    """mcts.roofline.kernels comes from a committed full-game profile and refuses kernels with fewer
    than 100 launches (a one-launch sample had been reported)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    k = {"calls": 500, "pmc_launches": 500, "avg_us": 90.0, "SQ_INSTS_MFMA": 1.8e6, "hbm_bytes_per_launch": 3e7}
    thin = dict(k, calls=1, pmc_launches=1)
    (prof / "r98_mcts_fullgame_pmc_summary.json").write_text(json.dumps(
        {"command": "x", "kernels": {"value_rollout_kernel": k, "policy_resident_kernel": k}}))
    (prof / "r99_mcts_fullgame_pmc_summary.json").write_text(json.dumps(
        {"command": "y", "kernels": {"value_rollout_kernel": thin, "policy_resident_kernel": k}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.net_kernel_profiles()
    assert got["profile"] == "r98_mcts_fullgame_pmc_summary.json"      # the newer one is refused
    v = got["value_rollout_kernel"]
    assert v["launches"] == 500 and abs(v["executed_tflops"] - 1.8e6 * 32768 / 90e-6 / 1e12) < 1e-6
    assert abs(v["frac"] - v["executed_tflops"] / 2500.0) < 1e-12
    # useful figures: boards per launch from the MFMA count, SURVEY 8(d)'s FLOPs per evaluation
    boards = 1.8e6 / bench.VALUE_MFMA_PER_BOARD
    assert abs(v["boards_per_launch"] - boards) < 1e-9
    assert abs(v["useful_tflops"] - boards * bench.VALUE_FLOP / 90e-6 / 1e12) < 1e-6
    assert abs(v["useful_frac_f16_peak"] - v["useful_tflops"] / 2500.0) < 1e-12
    # provenance: a profile without the hash of the kernel sources it was taken on is
    # STALE and says so; one that carries the current hash is `current`
    assert got["current"] is False and "STALE" in got["provenance"]
    (prof / "r97_mcts_fullgame_pmc_summary.json").write_text(json.dumps(
        {"command": "z", "csrc_sha16": bench.csrc_sha16(),
         "kernels": {"value_rollout_kernel": k, "policy_resident_kernel": k}}))
    os.remove(str(prof / "r98_mcts_fullgame_pmc_summary.json"))
    got = bench.net_kernel_profiles()
    assert got["profile"] == "r97_mcts_fullgame_pmc_summary.json" and got["current"] is True


def _preflight_worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("IAGO_BENCH_DEVICE", None)
    import time
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fd = os.open(os.path.join(out_dir, "rank%d.json" % rank), os.O_WRONLY | os.O_CREAT)
    if mode == "absent" and rank == 1:
        time.sleep(20)           # never enters the collective while rank 0 waits (its limit: 3 s)
        os._exit(0)
    uuid = "GPU-shared" if mode == "shared" else "GPU-%d" % rank
    got = bench.preflight(dist, world, rank, 0, fd, timeout_s=3.0, uuid=uuid, on="cpu")
    os.write(fd, (__import__("json").dumps(got) + "\n").encode())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "shared", "absent"])
def test_bench_preflight(mode, tmp_path):
    """bench.py's check before an N > 1 run is timed: a 1-element all-reduce (sum of the ranks) and the ranks' device
    uuids, polled under a time limit.  Two ranks on one device, or a rank that never joins the collective, end the
    run with ONE JSON line carrying "error" on rank 0's stdout and a non-zero exit -- no hang."""
    import json
    world = 2
    port = 31000 + (os.getpid() + {"ok": 0, "shared": 1, "absent": 2}[mode]) % 2000
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, mode, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    procs[0].join(timeout=120)
    assert procs[0].exitcode is not None
    line = json.loads(open(os.path.join(str(tmp_path), "rank0.json")).read().strip().splitlines()[-1])
    if mode == "ok":
        procs[1].join(timeout=60)
        assert procs[0].exitcode == 0 and procs[1].exitcode == 0
        assert line["allreduce_sum_of_ranks"] == 1 and line["distinct_devices"] == 2 and line["device_check"] == "ok"
    else:
        assert procs[0].exitcode == 3 and "error" in line
        assert ("share a device" in line["error"]) if mode == "shared" else ("did not complete" in line["error"])
        if mode == "absent":
            procs[1].kill()          # (this very process object: not a pattern)
        procs[1].join(timeout=60)


def test_bench_value_spread_and_variant_profiles(tmp_path, monkeypatch):
    """`value_spread`: min / max of `value` over the committed one-GPU lines of the same workload and this run;
    `variant_profile`: a variant leg's profile figures are quoted only from a profile taken on these kernel sources."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    line = {"metric": "self-play games/sec", "n_gpus": 1, "value": 2900.0,
            "config": {"games_per_gpu": 1024, "sims_per_move": 100, "full_games": True}}
    (prof / "r06a_bench.json").write_text(json.dumps(line) + "\n")
    (prof / "r06b_bench.json").write_text(json.dumps(dict(line, value=2850.0)) + "\n")
    (prof / "r06c_bench.json").write_text(json.dumps(dict(line, value=9999.0, n_gpus=8)) + "\n")          # not a one-GPU line
    (prof / "r06d_bench.json").write_text(json.dumps(dict(line, value=1.0, config=dict(line["config"], games_per_gpu=256))) + "\n")
    (prof / "r06e_bench.json").write_text("not json\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.value_spread(1024, 100, 2925.0)
    assert got["min"] == 2850.0 and got["max"] == 2925.0 and got["boxes"] == 3
    assert got["committed_lines"] == ["r06a_bench.json", "r06b_bench.json"]
    k = {"calls": 4, "avg_us": 1.7e6, "SQ_INSTS_MFMA": 1.29e11, "hbm_bytes_per_launch": 1.5e12, "l2_hit_rate": 0.9,
         "SQ_LDS_BANK_CONFLICT": 2.0, "SQ_LDS_IDX_ACTIVE": 10.0}
    (prof / "r06a_mcts400_persistent_pmc_summary.json").write_text(json.dumps(
        {"command": "x", "csrc_sha16": "0000", "kernels": {"search_kernel": k}}))
    stale = bench.variant_profile("mcts400")
    assert stale["profile"] == "r06a_mcts400_persistent_pmc_summary.json" and stale["profile_current"] is False
    assert "traffic" not in stale
    (prof / "r06b_mcts400_persistent_pmc_summary.json").write_text(json.dumps(
        {"command": "y", "csrc_sha16": bench.csrc_sha16(), "kernels": {"search_kernel": k}}))
    cur = bench.variant_profile("mcts400")
    assert cur["profile_current"] is True and cur["traffic"] == 1.5e12 and abs(cur["rocprof_kernel_avg_ms"] - 1700.0) < 1e-9
    assert abs(cur["executed_frac_pmc"] - 1.29e11 * 16384 / 1.7 / 1e12 / 2500.0) < 1e-9 and cur["lds_bank_conflict_share"] == 0.2
    assert bench.variant_profile("mctsnthr1") == {"profile": None, "profile_current": False}
