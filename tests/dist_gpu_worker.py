"""One rank of a world-size-W rehearsal on ONE GPU (tests/test_dist_gpu.py starts W fresh copies
of this script as child processes): the sharded PV-MCTS self-play engine and the REINFORCE
trainer exactly as an N-GPU run drives them -- shard_range, Philox streams keyed by the global
game id, gather_tuples, the replica broadcast -- with every rank on cuda:0 and the `gloo`
backend (RCCL refuses two ranks on one device; iago_amd.dist stages gloo's payloads through
host memory).  Not a test module: no test_ prefix.

    python tests/dist_gpu_worker.py selfplay <n_games> <n_sims> <out.npz>
    python tests/dist_gpu_worker.py reinforce <n_steps> <out-prefix>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shipped_rollout(ops):
    import json
    with open(os.path.join(ROOT, "tests", "golden", "simulate.json")) as f:
        g = json.load(f)
    return ops.RolloutWeights(g["shipped_w"], g["shipped_b"])


def selfplay(n_games, n_sims, out):
    import torch.distributed as dist
    from iago_amd import engine, network, ops
    from iago_amd.dist import gather_tuples, shard_range
    torch.manual_seed(0)                      # replicated weights: the same random init on every rank
    policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
    lo, hi = shard_range(n_games)
    m = engine.BatchedMCTS(hi - lo, policy, value, shipped_rollout(ops), n_thr=15, capacity=4096, seed=7,
                           game_id_base=lo, use_graph=True)
    res = engine.SelfPlayEngine(m).play(n_sims)
    tup = res.tuples()      # with the (global game id, turn) of every row: the comparison sorts by it
    B = res.valid.shape[1]
    got = gather_tuples(tup)
    zs = gather_tuples(dict(z=res.z, game=(torch.arange(B, device="cuda") + lo).to(torch.int32)))
    if dist.get_rank() == 0:
        np.savez(out, **{k: v.cpu().numpy() for k, v in got.items()},
                 final_z=zs["z"].cpu().numpy(), final_game=zs["game"].cpu().numpy(),
                 leaf_evals=np.asarray(m.n_leaf_evals))


def reinforce(n_steps, prefix):
    import torch.distributed as dist
    from iago_amd import network
    from iago_amd.train_rl import ReinforceTrainer
    # MIOpen's default weight-gradient kernels accumulate with atomics: two runs of the SAME rank
    # count already differ in the last bit of a gradient, and Adam's m / (sqrt(v) + eps) turns
    # that into +-alpha wherever a gradient entry is noise (measured: 2e-4 after 4 updates).  The
    # comparison of 1 vs 2 ranks therefore asks for deterministic convolution algorithms
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(100 + dist.get_rank())  # DIFFERENT initial replicas: sync_replicas must fix that
    tr = ReinforceTrainer(network.SLPolicy(), pool_dir=None, N=32, seed=3 + dist.get_rank())
    recorded = []
    play_set = tr.play_set

    def spy(model2):
        tup, wins = play_set(model2)
        recorded.append({k: v.cpu().numpy() for k, v in tup.items()})
        return tup, wins

    tr.play_set = spy
    outs = [tr.step() for _ in range(n_steps)]
    # PV-MCTS tuples into the same update (BASELINE configs[4]): each rank searches its shard
    from iago_amd import engine, ops
    from iago_amd.dist import shard_range
    lo, hi = shard_range(16)
    value = network.Value()
    torch.manual_seed(5)
    value.reset_parameters_chainer()
    value = value.cuda().eval()
    tr.model1.eval()
    m = engine.BatchedMCTS(hi - lo, tr.model1, value, shipped_rollout(ops), n_thr=15, capacity=2048, seed=9,
                           game_id_base=lo, use_graph=True)
    res = engine.SelfPlayEngine(m, max_turns=8).play(20)
    last = tr.step_from_tuples(res.tuples())
    params = {k: v for k, v in tr.model1.npz_dict().items()}
    np.savez("%s.rank%d.npz" % (prefix, dist.get_rank()), **params,
             rates=np.asarray([o["rate"] for o in outs]), losses=np.asarray([o["loss"] for o in outs]),
             n_tuples=np.asarray([o["n_tuples"] for o in outs]),
             mcts_loss=np.asarray(last["loss"]), mcts_tuples=np.asarray(last["n_tuples"]),
             adam_t=np.asarray(tr.opt.t),
             **{"set%d_%s" % (i, k): v for i, r in enumerate(recorded) for k, v in r.items()})


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                  # every rank on the one GPU of the box
    dist.init_process_group(os.environ.get("IAGO_TEST_BACKEND", "gloo"), rank=rank, world_size=world)
    try:
        if sys.argv[1] == "selfplay":
            selfplay(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        else:
            reinforce(int(sys.argv[2]), sys.argv[3])
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
