"""Lab tool: hunts the role split's rare clock-limit abort (round 6).  Fresh engines play whole batches with a SHORT clock
limit until a launch gives up, then the protocol's state is dumped: ring heads / tails, finished workgroups, and for every
game that never finished what it was waiting for (its reply tags against the ring's entries).

    IAGO_PERSISTENT_GAME_LIMIT_MS=3000 IAGO_SEARCH_SPLIT=16 python tools/debug_split_abort.py [batches=60] [games=1024]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iago_amd import _lib, engine, network, ops  # noqa: E402

os.environ.setdefault("IAGO_PERSISTENT_GAME_LIMIT_MS", "3000")
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
games = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w, b = bench.shipped_rollout_weights()
torch.manual_seed(0)
policy, value = network.SLPolicy().cuda().eval(), network.Value().cuda().eval()
for i in range(batches):
    m = engine.BatchedMCTS(games, policy, value, ops.RolloutWeights(w, b), n_thr=15, seed=7 + i,
                           capacity=engine.suggest_capacity(100, 15, moves=64))
    m.time_limit_ms = 1000
    eng = engine.SelfPlayEngine(m)
    if "replytimes" in os.environ.get("IAGO_HIP_LIB", ""):
        m.trace = torch.zeros((4096 + 512, 4), dtype=torch.int64, device="cuda")   # (the variant's per-game / per-workgroup rows)
    try:
        eng.play(100)
        torch.cuda.synchronize()
        m.close()
        continue
    except _lib.IagoError as e:
        print("batch %d: %s" % (i, str(e)[:120]))
    torch.cuda.synchronize()
    ps = m._ps
    ctl = ps["ctl"].cpu().numpy().view(np.uint32)
    print("ctl: finished %d abort %d net wgs %d idle %d | value ring head %d tail %d | policy ring head %d tail %d | playing %d"
          % (ctl[2], ctl[3], ctl[7], ctl[14], ctl[8], ctl[9], ctl[10], ctl[11], ctl[6]))
    print("split cus %d, game workgroups %d, net workgroups %d" % (m.split_cus, -(-games // m.games_per_workgroup), m.net_workgroups))
    tot = ps["totals"].cpu().numpy()
    print("totals", tot.tolist())
    q = ps["q_slots"].cpu().numpy().view(np.uint64).reshape(2, _lib.SEARCH_QUEUE_ENTRIES, 8)
    for ring, (h, t) in enumerate(((ctl[8], ctl[9]), (ctl[10], ctl[11]))):
        lo = max(0, int(min(h, t)) - 4)
        print("ring %d entries around head/tail (ticket: tag of word 0 .. 5, game):" % ring)
        for tk in list(range(lo, min(lo + 12, int(max(h, t)) + 3))) + list(range(max(lo + 12, int(max(h, t)) - 6), int(max(h, t)) + 3)):
            e = q[ring, tk % _lib.SEARCH_QUEUE_ENTRIES]
            print("   ticket %d: tags %s kind|game %#x" % (tk, [int(x >> np.uint64(32)) for x in e[:6]], int(e[0] & np.uint64(0xFFFFFFFF))))
    done = ps["done"].cpu().numpy()
    rep_v = ps["rep_v"].cpu().numpy().view(np.uint64)
    print("games' progress (turn x 100 + playouts) min %d max %d; reply tags of the value mailboxes of the 8 slowest games:" % (done.min(), done.max()))
    for g in np.argsort(done)[:8]:
        print("   game %d progress %d rep_v tag %d" % (g, done[g], int(rep_v[g] >> np.uint64(32))))
    # per game-workgroup: how many games lag
    gpw = m.games_per_workgroup
    lag = (done < np.median(done) - 200).reshape(-1, gpw).sum(1) if games % gpw == 0 else None
    print("lagging games per game workgroup:", None if lag is None else lag.tolist())
    # what the unfinished games wait for: their last request's reply tag (epoch) is not visible from outside; the mailboxes'
    # tags and the rings' unread entries are.  Entries of ring 0 between head and tail that name a game:
    h, t = int(ctl[8]), int(ctl[9])
    waiting = []
    for tk in range(min(h, t), max(h, t)):
        e = q[0, tk % _lib.SEARCH_QUEUE_ENTRIES]
        if int(e[0] >> np.uint64(32)) == tk + 1 and int(e[0] & np.uint64(0x7FFFFFFF)) != 0x7FFFFFFF:
            waiting.append((tk, int(e[0] & np.uint64(0x7FFFFFFF))))
    print("value ring: %d unread entries name a game (first: %s)" % (len(waiting), waiting[:10]))
    # the unfinished games' own account (written by the kernel when it gives up): state, the reply tag waited for
    names = ("READY", "WAIT_PRIOR", "PRIOR_READY", "ROLL", "ROLL_FRESH", "WAIT_VALUE", "HAVE_VALUE", "DONE", "TURN", "MOVE")
    cur, lv = m.cur_node.cpu().numpy(), m.leaf_value.cpu().numpy()
    rep_p = ps["rep_p"].cpu().numpy().view(np.uint64).reshape(games, 64)
    stuck = [int(g) for g in np.argsort(done)[:int(ctl[6]) + 3]]
    tr = m.trace.cpu().numpy() if getattr(m, "trace", None) is not None else None
    if tr is not None:
        base = tr[:games, 0][tr[:games, 0] > 0].min()
        wg = tr[4096:4096 + 512]
        held = {}
        for b in range(512):
            if wg[b, 0] or wg[b, 1]:
                held[(int(wg[b, 2] >> 32), int(wg[b, 2] & 0xFFFFFFFF))] = b
        stages = {}
        for b in range(512):
            if wg[b, 0] or wg[b, 1]:
                stages[int(wg[b, 3])] = stages.get(int(wg[b, 3]), 0) + 1
        print("net workgroups by last stage (1 loop top, 2 waiting for a ticket, 3 fetched, 23 gave up in fetch; + 100 left):", stages)
        for g in stuck:
            if tr[g, 2] == 0:
                continue
            key = (int(tr[g, 3] >> 32) & 1, int(tr[g, 3] & 0xFFFFFFFF))   # (the post-mortem overwrote [3] for unfinished games)
        # the games' request tickets are in rows [3] only for finished games; for stuck ones use the rings: who waited for what
        late = sorted(((wg[b, 1] - base) / 100.0, b, int(wg[b, 2] >> 32), int(wg[b, 2] & 0xFFFFFFFF), int(wg[b, 3]), (wg[b, 0] - base) / 100.0)
                      for b in range(512) if wg[b, 1])
        print("net workgroups that began their last wait EARLIEST (began waiting us, workgroup, ring, ticket, stage, last loop top us):")
        for row in late[:24]:
            print("    %12.1f  wg %3d  ring %d ticket %8d  stage %3d  loop top %12.1f" % row)
    for g in stuck:
        if tr is not None:
            print("   game %4d: last request at %.1f us, its reply written at %.1f us, workgroup left at %.1f us (since the first game's first request)"
                  % (g, (tr[g, 0] - tr[:, 0][tr[:, 0] > 0].min()) / 100.0, (tr[g, 1] - tr[:, 0][tr[:, 0] > 0].min()) / 100.0,
                     (tr[g, 2] - tr[:, 0][tr[:, 0] > 0].min()) / 100.0))
        st_ = int(lv[g]) if 0 <= lv[g] < 10 and lv[g] == int(lv[g]) else -1
        ptags = sorted(set(int(x >> np.uint64(32)) for x in rep_p[g]))
        print("   game %4d (workgroup %2d) progress %5d state %-11s waits for tag %6d | value mailbox tag %6d | prior mailbox tags %s"
              % (g, g // gpw, done[g], names[st_] if st_ >= 0 else "?", cur[g], int(rep_v[g] >> np.uint64(32)), ptags[-3:]))
    break
else:
    print("no abort in %d batches" % batches)
