"""Batched PV-MCTS and self-play on one GPU.

Two engines behind one class, same trees bit for bit: the PERSISTENT search (round 4; the default wherever the
split-f16 Value net and the three-piece SLPolicy apply) -- a whole search, or a whole batch of self-play games, as
ONE launch in which every game runs on its own clock (iago_mcts_search_persistent, csrc/search_kernel.hip) -- and
the per-playout launches described below (use_graph=True and the look-ahead options select them).

Host-side select/expand/backup loop over thousands of games (the reference runs
one game, one playout at a time: MCTS.py:105-147, game.py:117-142).  Every
tree operation, board update and the leaf rollout is a HIP kernel behind the C
ABI (include/iago_hip.h); the policy and value nets are PyTorch-ROCm modules
(or any callables on a CUDA planes tensor).  The reference's constants and
quirks are the defaults: lmbda=0.5, c_puct=1, n_thr=15,
P = prior + 0.1, U = c*P*sqrt(N)/(0.01+n), root evaluated itself for its first
n_thr simulations, no sign flip in backup, pass = action -1, subtree reuse.
The wall-clock budget (10 s per move, MCTS.py:142) becomes a simulation count.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import MctsTree, check

START_OWN = 0x0000000810000000  # colour 1 "X", moves first: (3,4), (4,3)  (game.py:26-30)
START_OPP = 0x0000001008000000  # colour 2 "O": (3,3), (4,4)
HANDICAP_CELLS = (2 * 8 + 4, 3 * 8 + 5, 4 * 8 + 2, 5 * 8 + 3)  # src/train_rl.py:45


def _p(t):
    return C.c_void_p(t.data_ptr())


_stream = ops._stream   # the current HIP stream as a void*

_SEARCH_STREAMS = {}    # (device, game CUs) -> iago_search_streams* (None: this runtime gives no CU-masked streams)


def _search_streams(game_cus):
    """The process's CU-masked streams of the role-split search on the current device (iago_mcts_search_streams_create:
    created once, kept for the life of the process), or None where they cannot be had."""
    key = (torch.cuda.current_device(), int(game_cus))
    if key not in _SEARCH_STREAMS:
        h = C.c_void_p()
        rc = _lib.lib().iago_mcts_search_streams_create(int(game_cus), C.byref(h))
        _SEARCH_STREAMS[key] = h if rc == 0 and h.value else None
    return _SEARCH_STREAMS[key]


def suggest_capacity(n_sims, n_thr=15, moves=64, branching=12):
    """Nodes per game that a whole self-play game needs without ever compacting the pools:
    every expansion adds ~`branching` children, a search adds at most
    n_sims / n_thr + 1 expansions.  The turn-by-turn loop (BatchedMCTS.search) compacts a pool that is
    half full -- TreePool.compact -- so there a smaller capacity only costs compaction passes, as long as
    one search's live tree fits in half of it.  The one-launch whole-game path of the persistent search
    never compacts: SelfPlayEngine.play takes it only for a pool of at least half this size (the nodes a
    game really leaves behind: 1,718 at 100 playouts per move) and replays the batch through the turn
    loop if a pool fills up all the same."""
    per_move = (n_sims // max(n_thr, 1) + 1) * branching
    cap = 1024
    while cap < per_move * moves:
        cap *= 2
    return cap


class TreePool(object):
    """Device memory of the per-game search trees (iago_mcts_tree): ONE array of 32-byte node
    records (iago_mcts_node); `n_visits`, `q`, `p`, `v`, `first_child`, `parent`, `action`,
    `n_children` are strided views of it (read them with .cpu(), fill them in place)."""

    def __init__(self, n_games, capacity, device="cuda", value_cache=False):
        if not torch.cuda.is_available():
            raise _lib.IagoError("TreePool needs a HIP device (no CPU fallback)")
        self.n_games, self.capacity = n_games, capacity
        self._alloc(device, value_cache)
        self.reset()

    def _alloc(self, device, value_cache):
        n = self.n_games * self.capacity
        kw = dict(device=device)
        self.nodes = torch.zeros((n, _lib.NODE_WORDS), dtype=torch.int32, **kw)
        f32, i8 = self.nodes.view(torch.float32), self.nodes.view(torch.int8)
        self.n_visits, self.q, self.p = self.nodes[:, 0], f32[:, 1], f32[:, 2]
        # value_func(node) once evaluated, NaN before (iago_mcts_fresh_leaves); None = no cache
        self.v = f32[:, 3] if value_cache else None
        self.first_child, self.parent = self.nodes[:, 4], self.nodes[:, 5]
        self.action, self.n_children = i8[:, 24], self.nodes.view(torch.uint8)[:, 25]
        self.n_nodes = torch.zeros(self.n_games, dtype=torch.int32, **kw)
        self.root = torch.zeros(self.n_games, dtype=torch.int32, **kw)
        self.overflow = torch.zeros(self.n_games, dtype=torch.int32, **kw)
        t = MctsTree()
        t.n_games, t.capacity, t.has_v = self.n_games, self.capacity, 1 if value_cache else 0
        t.nodes = self.nodes.data_ptr()
        t.n_nodes, t.root, t.overflow = self.n_nodes.data_ptr(), self.root.data_ptr(), self.overflow.data_ptr()
        self.c = t

    def ref(self):
        return C.byref(self.c)

    def reset(self, mask=None):
        check(_lib.lib().iago_mcts_reset(self.ref(), _p(mask) if mask is not None else None,
                                         _stream()), "iago_mcts_reset")
        for hook in getattr(self, "reset_hooks", ()):
            hook(mask)

    def bytes(self):
        return self.nodes.numel() * 4

    def compact(self, mask=None):
        """Garbage collection (iago_mcts_compact): the live subtree of every game (mask: uint8
        per game, None = all) re-laid from index 0; the nodes abandoned by subtree reuse are
        freed.  A second pool and an index array are allocated on first use."""
        if getattr(self, "_scratch", None) is None:
            sc = TreePool.__new__(TreePool)
            sc.n_games, sc.capacity = self.n_games, self.capacity
            sc._alloc(self.nodes.device, self.v is not None)
            self._scratch = sc
            self._order = torch.empty(self.n_games * self.capacity, dtype=torch.int32, device=self.nodes.device)
        check(_lib.lib().iago_mcts_compact(self.ref(), self._scratch.ref(), _p(self._order),
                                           _p(mask) if mask is not None else None, _stream()),
              "iago_mcts_compact")

    def dump(self, g, max_depth=6):
        """Host copy of game g's tree in the format of oracle.mcts_py.dump_tree."""
        lo, hi = g * self.capacity, g * self.capacity + int(self.n_nodes[g].item())
        rec = self.nodes[lo:hi].cpu()
        f32, i8 = rec.view(torch.float32), rec.view(torch.int8)
        arr = {"first_child": rec[:, 4].numpy(), "n_children": rec.view(torch.uint8)[:, 25].numpy(),
               "action": i8[:, 24].numpy(), "n_visits": rec[:, 0].numpy(), "q": f32[:, 1].numpy(),
               "p": f32[:, 2].numpy()}

        def rec(i, depth):
            d = dict(n=int(arr["n_visits"][i]), Q=float(arr["q"][i]), P=float(arr["p"][i]),
                     children={}, order=[])
            fc, k = int(arr["first_child"][i]), int(arr["n_children"][i])
            if fc >= 0:
                d["order"] = [int(arr["action"][fc + j]) for j in range(k)]
                if depth < max_depth:
                    for j in range(k):
                        d["children"][str(int(arr["action"][fc + j]))] = rec(fc + j, depth + 1)
            return d

        return rec(int(self.root[g].item()), 0)


class BatchedMCTS(object):
    """MCTS(lmbda, c_puct, n_thr) of MCTS.py:78-154 for n_games trees at once.

    policy_fn(planes) -> (L,64) probabilities, value_fn(planes) -> (L,) values,
    planes = (L,2,8,8) float32 CUDA tensor (GameFunctions.make_state_var
    layout).  rollout_weights: ops.RolloutWeights (None = uniform random
    rollouts).  rollout_hook(engine) runs after every rollout launch (tests record `engine.z`).

    Two ways through a playout (MCTS.py:105-133), same trees:
      * sync-free (default when policy_fn has `forward_counted(planes, n_dev)`, as
        network.SLPolicy does): the number of leaves that expand stays on the device
        (iago_mcts_pending -> n_dev of the policy kernels and of iago_mcts_expand), so a
        playout is a fixed sequence of launches with no host synchronisation --
        select, pending, planes, policy net, expand, continue-select, value net, rollout,
        leaf mix + backup.  use_graph=True captures that sequence once and replays it
        with ONE launch per playout.
      * host-counted (arbitrary callables, e.g. the stand-in nets of the parity tests):
        one host sync per playout tells how many leaves expand; the policy callable sees
        exactly those rows.  sync_free=True forces the first way for any callable (it
        is then evaluated on all n_games rows, those past the count being ignored).
    """

    def __init__(self, n_games, policy_fn, value_fn, rollout_weights, lmbda=0.5, c_puct=1.0,
                 n_thr=15, capacity=4096, seed=0, game_id_base=0, device="cuda", use_graph=False,
                 sync_free=None, lookahead=None, lookahead_slots=None, value_cache=None, lookahead_overlap=None,
                 z_log_rows=0, async_steps=None, async_parts=None, value_ahead=None, persistent=None,
                 net_workgroups=None, max_cus=None, split=None):
        if n_thr < 1:
            raise ValueError("n_thr must be >= 1")
        # (what the caller asked for explicitly, before the defaults below fill the options in: any of these selects
        # the per-playout launches unless `persistent` says otherwise)
        per_playout_asked = bool(use_graph or async_steps or value_ahead or lookahead is not None
                                 or lookahead_overlap is not None or sync_free is not None)
        self.n_games = n_games
        self.policy_fn, self.value_fn, self.rollout_weights = policy_fn, value_fn, rollout_weights
        self.lmbda, self.c_puct, self.n_thr = float(lmbda), float(c_puct), int(n_thr)
        self.seed, self.game_id_base = seed, game_id_base
        kw = dict(device=device)
        # Value cache (iago_mcts_fresh_leaves in include/iago_hip.h): the value net runs only on
        # the leaves it has not evaluated yet (~15 % of the playouts' leaves); every other visit
        # of a leaf takes the value stored in its node.  Same trees.  Default: on whenever the
        # value net can be fed a device-side list of boards.
        can_cache = (value_fn is not None and self.lmbda < 1.0
                     and getattr(value_fn, "forward_boards_counted", None) is not None
                     and getattr(value_fn, "split_f16", False))
        if value_cache is None:
            value_cache = can_cache
        if value_cache and not can_cache:
            raise ValueError("value_cache needs a value net with forward_boards_counted (split-f16 path) and lmbda < 1")
        self.value_cache = bool(value_cache)
        self.tree = TreePool(n_games, capacity, device, value_cache=self.value_cache)
        self._fresh_idx = torch.zeros(n_games, dtype=torch.int64, **kw)
        self._fresh_count = torch.zeros(1, dtype=torch.int32, **kw)
        self._value_total = torch.zeros(1, dtype=torch.int64, **kw)  # value-net evaluations, on the device
        self._value_key = None
        self.fused_leaf_eval = os.environ.get("IAGO_FUSED_LEAF_EVAL", "1") != "0"
        self.cur_node = torch.zeros(n_games, dtype=torch.int32, **kw)
        self.cur_own = torch.zeros(n_games, dtype=torch.int64, **kw)
        self.cur_opp = torch.zeros(n_games, dtype=torch.int64, **kw)
        self.needs_expand = torch.zeros(n_games, dtype=torch.uint8, **kw)
        self._pending = torch.zeros(n_games, dtype=torch.uint8, **kw)
        self._pend_idx = torch.zeros(n_games, dtype=torch.int64, **kw)
        self._pend_games = torch.zeros(n_games, dtype=torch.int32, **kw)
        self._pend_count = torch.zeros(1, dtype=torch.int32, **kw)
        self._pend_total = torch.zeros(1, dtype=torch.int64, **kw)  # policy evaluations, on the device
        self.legal = torch.zeros(n_games, dtype=torch.int64, **kw)
        self.leaf_value = torch.zeros(n_games, dtype=torch.float32, **kw)
        self.planes = torch.zeros((n_games, 2, 8, 8), dtype=torch.float32, **kw)
        self.z = torch.zeros(n_games, dtype=torch.int8, **kw)
        self.v = torch.zeros(n_games, dtype=torch.float32, **kw)
        self.move = torch.zeros(n_games, dtype=torch.int8, **kw)
        self.visits = torch.zeros((n_games, 64), dtype=torch.int32, **kw)
        self._policy_in = torch.zeros((max(n_games, 16), 2, 8, 8), dtype=torch.float32, **kw)
        self.stats = None             # optional (n_games, 2) int32: levels, children scored
        if sync_free is None:
            sync_free = getattr(policy_fn, "forward_counted", None) is not None
        self.sync_free = bool(sync_free)
        if use_graph and not self.sync_free:
            raise ValueError("use_graph needs the sync-free playout (a policy with forward_counted)")
        self.use_graph, self._graph, self._graph_key = bool(use_graph), None, None
        # look-ahead blocks (of 2 K playouts) per replay of the long graph (tuning knob: DESIGN.md)
        self.graph_blocks = max(1, int(os.environ.get("IAGO_GRAPH_BLOCKS", "4")))
        self._graph_long = None
        self.n_compactions = 0
        self._live_after_compaction = 0
        # Policy look-ahead (iago_mcts_lookahead in include/iago_hip.h): leaves are queued K
        # visits before they expand and the policy net runs on the queue every K playouts, off
        # the playouts' critical path.  Default: K = 4 whenever the sync-free playout with a
        # board-fed policy net applies and n_thr leaves room for it; 0 = the net runs inside the
        # playout that expands (the reference's order of evaluation).  Same trees either way.
        # Persistent search (iago_mcts_search_persistent in include/iago_hip.h): a whole search is ONE
        # launch in which every game runs on its own clock -- game workgroups (32 games each: descent,
        # rollout, backup) and net workgroups that serve a queue of positions with the one-board walks
        # of the value and the policy net.  The policy net runs exactly where the reference runs it (at
        # the expansion), so there is no look-ahead in this mode.  Same trees.
        # (the launch keeps every game workgroup resident and needs net workgroups beside them: at most half of the
        # device's 256 CUs for the games, i.e. 4096 games per launch; larger batches take the per-playout launches)
        # (tuning knobs: pacing of the leading games -- playouts a game may be ahead of the mean while requests queue,
        # 0 = the library's default, < 0 = off; 8 / 16 / 32 games per game workgroup)
        self.pace_margin = int(os.environ.get("IAGO_PERSISTENT_PACE", "0"))
        # (games per game workgroup: 32 to 60 game workgroups measure best at every batch size -- 256 games 6.6 M leaf-evals/s
        # at 8 per workgroup against 5.1 M at 32; 512 / 640 / 768 / 896 games 11.7 / 13.1 / 14.3 / 14.5 M at 16 against 9.7 /
        # 11.2 / 12.8 / 14.0 M at 32; 1024 games 15.5 M at 32 against 14.4 M at 16: a workgroup's iteration is as long as its
        # rollout passes of 16 boards, and every game workgroup is a net workgroup less)
        self.games_per_workgroup = int(os.environ.get("IAGO_PERSISTENT_GPW", "0")) or (
            8 if n_games <= 256 else 16 if n_games <= 960 else _lib.SEARCH_GAMES_PER_WORKGROUP)
        # The launch's grid follows the device (iago_mcts_search_capacity: CUs x workgroups of the search kernel per CU,
        # all of them resident from the start); max_cus / IAGO_PERSISTENT_CUS: the CUs the launch may count on when fewer
        # are free for it -- a CU-masked stream, a device shared with another job.  The games take at most half of them
        # (4096 games per launch on a whole MI355X); larger batches take the per-playout launches.
        self.max_cus = int(max_cus if max_cus is not None else os.environ.get("IAGO_PERSISTENT_CUS", "0"))
        self.resident_workgroups = self._search_capacity()
        n_gw = -(-n_games // self.games_per_workgroup)
        # Role split (iago_mcts_search_split): the game workgroups as a launch of their own, two per CU on `split` CUs
        # (a multiple of 8), the net workgroups on all the others -- two CU-masked streams, co-resident by construction.
        # Same trees.  Default ("auto"): wherever the games need more than 32 workgroups -- in the single launch every
        # game workgroup holds a CU alone, so beyond 32 of them each one is a net workgroup less; two per CU give the CUs
        # back (1536 / 2048 / 4096 games: 17.4 -> 17.9, 17.1 -> 18.3, 10.8 -> 15.1 M leaf-evals/s; 1024 games: no
        # difference, the single launch stays; LABNOTES.md, round 6).  split / IAGO_SEARCH_SPLIT: game CUs, 0 = always the single
        # launch.  Not with max_cus (the split owns the device); a runtime without CU-masked streams falls back to
        # the single launch.
        want = split if split is not None else os.environ.get("IAGO_SEARCH_SPLIT", "auto")
        if want == "auto":
            # (32-game workgroups only: the smaller workgroups of batches up to 960 games measure slower two per CU --
            # 640 / 768 / 896 games at 16 per workgroup: 14.9 / 16.3 / 16.6 M single, 13.5 / 15.1 / 16.3 M split)
            want_split = (8 * (-(-n_gw // 16)) if (n_gw > 32 and self.games_per_workgroup == _lib.SEARCH_GAMES_PER_WORKGROUP)
                          else 0)                                     # (two game workgroups per CU)
        else:
            want_split = int(want)
        self.split_cus, self._split = 0, None
        while want_split > 0 and 2 * want_split < n_gw:
            want_split += 8
        can_p = (self.resident_workgroups > 0 and 2 * n_gw <= self.resident_workgroups
                 and can_cache and getattr(value_fn, "search_args", None) is not None
                 and getattr(policy_fn, "search_args", None) is not None and getattr(policy_fn, "split3", False)
                 and rollout_weights is not None and not rollout_weights.log_form and 0.0 <= self.lmbda < 1.0)
        # Default: ON wherever it applies, unless the caller asks for the per-playout launches (use_graph,
        # look-ahead / asynchronous-step / value-look-ahead options, a rollout hook comes later); the environment
        # variable IAGO_PERSISTENT=0 / 1 overrides both (measurements: tools/time_value_ahead.py).
        env_p = os.environ.get("IAGO_PERSISTENT")
        if env_p in ("0", "1"):
            persistent = can_p and env_p == "1"
        elif persistent is None:
            persistent = can_p and not per_playout_asked
        if persistent and not can_p:
            raise ValueError("persistent needs the split-f16 value net and the three-piece policy net (modules with "
                             "search_args), product-form rollout weights, lmbda < 1 and at most %d games (the games' "
                             "workgroups may take half of the %d workgroups this device keeps resident)"
                             % (self.resident_workgroups // 2 * self.games_per_workgroup, self.resident_workgroups))
        self.persistent = bool(persistent)
        if self.persistent:
            if not self.value_cache:
                raise ValueError("persistent needs the value cache")
            if (want_split > 0 and want_split % 8 == 0 and self.max_cus <= 0
                    and want_split <= self.resident_workgroups // 2):
                self._split = _search_streams(want_split)     # (None: no CU-masked streams here -> the single launch)
                self.split_cus = want_split if self._split else 0
            lookahead, async_steps, value_ahead, use_graph = 0, False, False, False
            self.use_graph = False
        can = (self.sync_free and getattr(policy_fn, "forward_counted_boards", None) is not None)
        if lookahead is None:
            k = int(os.environ.get("IAGO_LOOKAHEAD", "4"))   # (tuning knob: tools/)
            lookahead = k if (can and self.n_thr > k + 1) else 0
        # lookahead_overlap = j: the policy batch of a group of K playouts runs on a second stream
        # BESIDE the first j playouts of the next group (0: at the end of its own group, on the
        # same stream); leaves are then queued j - 1 visits earlier so that their priors are
        # stored before any of them can expand.
        if lookahead_overlap is None:
            lookahead_overlap = int(os.environ.get("IAGO_LOOKAHEAD_OVERLAP", "2"))
        self.lookahead = int(lookahead)
        j = min(int(lookahead_overlap), max(self.lookahead - 1, 0)) if self.lookahead else 0
        while j > 0 and not self.n_thr > self.lookahead + j - 1:
            j -= 1   # n_thr leaves no room to queue the leaves that much earlier
        self.lookahead_overlap = j
        margin = self.lookahead + max(self.lookahead_overlap - 1, 0)
        if z_log_rows and not self.lookahead and not self.persistent:
            raise ValueError("z_log_rows needs the look-ahead playout (rollout_hook serves the other paths)")
        if self.lookahead and not (can and self.n_thr > margin):
            raise ValueError("lookahead needs the sync-free playout, a policy net with "
                             "forward_counted_boards and n_thr > lookahead (+ overlap - 1)")
        if self.lookahead:
            K = self.lookahead
            slots = int(lookahead_slots) if lookahead_slots else max(256, capacity // 8)
            Q = n_games * K
            self._la_next_seq = torch.zeros(n_games, dtype=torch.int32, **kw)
            self._la_cache_seq = torch.full((n_games, slots), -1, dtype=torch.int32, **kw)
            self._la_cache = torch.zeros((n_games, slots, 64), dtype=torch.float32, **kw)
            self._la_error = torch.zeros(1, dtype=torch.int32, **kw)
            # the nodes of a game's last descent (the one-launch descent records them, the backup
            # updates them side by side).  A path can be much longer than the plies left: at a
            # finished position every expansion adds one more pass child (MCTS.py:112-114), so the
            # buffer covers the descent's own bound of 512 levels
            self.PATH_STRIDE = 520
            self.fused_descent = os.environ.get("IAGO_FUSED_DESCENT", "1") != "0"
            use_path = self.fused_descent and os.environ.get("IAGO_BACKUP_PATH", "1") != "0"
            self._la_path = torch.zeros((n_games, self.PATH_STRIDE), dtype=torch.int32, **kw) if use_path else None
            self._la_path_len = torch.zeros(n_games, dtype=torch.int32, **kw) if use_path else None
            # two queues: the playouts of a group fill one while the other one's batch is in flight
            self._la_queues, self._la = [], []
            # diagnostic record of the parity tests: the z every playout of a game backed up, in
            # playout order (iago_mcts_lookahead.z_log); works in graph mode, unlike rollout_hook
            self.z_log = torch.zeros((z_log_rows, n_games), dtype=torch.int8, **kw) if z_log_rows else None
            self.z_log_n = torch.zeros(n_games, dtype=torch.int32, **kw) if z_log_rows else None
            for _ in range(2):
                q = dict(count=torch.zeros(1, dtype=torch.int32, **kw), own=torch.zeros(Q, dtype=torch.int64, **kw),
                         opp=torch.zeros(Q, dtype=torch.int64, **kw), game=torch.zeros(Q, dtype=torch.int32, **kw),
                         seq=torch.zeros(Q, dtype=torch.int32, **kw))
                a = _lib.MctsLookahead()
                a.trigger, a.slots, a.q_capacity = self.n_thr - margin, slots, Q
                a.next_seq, a.cache_seq = self._la_next_seq.data_ptr(), self._la_cache_seq.data_ptr()
                a.cache, a.q_count = self._la_cache.data_ptr(), q["count"].data_ptr()
                a.q_own, a.q_opp = q["own"].data_ptr(), q["opp"].data_ptr()
                a.q_game, a.q_seq = q["game"].data_ptr(), q["seq"].data_ptr()
                a.error = self._la_error.data_ptr()
                # the backup clears the fresh-leaf count that the next descent appends to
                a.clear_word = self._fresh_count.data_ptr() if self.value_cache else None
                if self._la_path is not None:
                    a.path, a.path_len = self._la_path.data_ptr(), self._la_path_len.data_ptr()
                    a.path_stride = self.PATH_STRIDE
                if self.z_log is not None:
                    a.z_log, a.z_log_n, a.z_log_rows = self.z_log.data_ptr(), self.z_log_n.data_ptr(), z_log_rows
                self._la_queues.append(q)
                self._la.append(a)
            self._la_cur = 0   # the queue the playouts fill
            prio = int(os.environ.get("IAGO_SIDE_PRIORITY", "0"))
            self._la_side = torch.cuda.Stream(device=device, priority=prio) if self.lookahead_overlap else None
            # Value look-ahead (iago_mcts_value_ahead in include/iago_hip.h): the descent queues every
            # node it expands; once per group of K playouts the children that have no value yet go
            # through the value net as ONE batch on the side stream and the results land in the
            # children's records, so that their first visits find a stored value instead of walking
            # the net on the playouts' critical path.  Same trees (a child visited before its value
            # has landed is evaluated in place as before).  Default OFF -- measured slower (round 4,
            # LABNOTES.md): in lockstep ONE game without a stored value still puts a 70 us one-board
            # walk on every playout's critical path (1024 games x 16 % fresh leaves x 53 % misses = ~85
            # per playout), and with game-asynchronous steps the extra evaluations (children that are
            # never visited, or visited before their value lands: +50 % rows) make the side stream the
            # bound.  value_ahead=True / IAGO_VALUE_AHEAD=1 selects it.
            can_va = bool(self.value_cache and self.fused_descent and self._la_side is not None
                          and getattr(value_fn, "forward_boards_batch", None) is not None)
            if value_ahead is None:
                value_ahead = can_va and os.environ.get("IAGO_VALUE_AHEAD", "0") == "1"
            if value_ahead and not can_va:
                raise ValueError("value_ahead needs the value cache, the one-launch descent, lookahead_overlap > 0 "
                                 "and a value net with forward_boards_batch")
            self.value_ahead = bool(value_ahead)
            if self.value_ahead:
                self.va_boards = int(os.environ.get("IAGO_VALUE_AHEAD_BOARDS", "2"))      # boards per workgroup
                self.va_grid = int(os.environ.get("IAGO_VALUE_AHEAD_GRID", "176"))        # workgroup cap of a batch
                xcap, rcap = n_games * K, max(4096, 8 * n_games * K)
                self._va_total = torch.zeros(1, dtype=torch.int64, **kw)   # rows the batches have evaluated
                self._va_row_count = torch.zeros(1, dtype=torch.int32, **kw)
                self._va_rows = dict(own=torch.zeros(rcap, dtype=torch.int64, **kw),
                                     opp=torch.zeros(rcap, dtype=torch.int64, **kw),
                                     node=torch.zeros(rcap, dtype=torch.int64, **kw),
                                     v=torch.zeros(rcap, dtype=torch.float32, **kw))
                self._va_x, self._va = [], []
                for a in self._la:
                    x = dict(count=torch.zeros(1, dtype=torch.int32, **kw), game=torch.zeros(xcap, dtype=torch.int32, **kw),
                             node=torch.zeros(xcap, dtype=torch.int32, **kw), own=torch.zeros(xcap, dtype=torch.int64, **kw),
                             opp=torch.zeros(xcap, dtype=torch.int64, **kw))
                    v = _lib.MctsValueAhead()
                    v.x_capacity, v.row_capacity = xcap, rcap
                    v.x_count, v.x_game, v.x_node = x["count"].data_ptr(), x["game"].data_ptr(), x["node"].data_ptr()
                    v.x_own, v.x_opp = x["own"].data_ptr(), x["opp"].data_ptr()
                    v.row_count = self._va_row_count.data_ptr()
                    v.row_own, v.row_opp = self._va_rows["own"].data_ptr(), self._va_rows["opp"].data_ptr()
                    v.row_node, v.row_v = self._va_rows["node"].data_ptr(), self._va_rows["v"].data_ptr()
                    v.total = self._va_total.data_ptr()
                    a.value_ahead = C.addressof(v)
                    self._va_x.append(x)
                    self._va.append(v)
                self._ev_priors = torch.cuda.Event()
                self._ev_rows = torch.cuda.Event()

            def reset_lookahead(mask):
                if mask is None:
                    self._la_next_seq.zero_()
                    self._la_cache_seq.fill_(-1)
                else:
                    m = mask.bool()
                    self._la_next_seq[m] = 0
                    self._la_cache_seq[m] = -1
            self.tree.reset_hooks = [reset_lookahead]
        self.value_ahead = bool(getattr(self, "value_ahead", False))
        if self.persistent:
            if net_workgroups is None:
                net_workgroups = int(os.environ.get("IAGO_PERSISTENT_NET", "0")) or max(32, 8 * n_games)
            # (an upper bound: the launch itself takes no more than fit beside the game workgroups)
            self.net_workgroups = max(1, min(int(net_workgroups), self.resident_workgroups - n_gw))
            if self.split_cus:   # (one net workgroup on every CU that is not the games', at most 7/8 of the device's CUs:
                # the library's cap -- beyond it the net launch's last workgroups were seen to stall, LABNOTES.md round 6)
                self.net_workgroups = max(1, min(int(net_workgroups), self.resident_workgroups
                                                 - max(self.split_cus, self.resident_workgroups // 8)))
            grid = n_gw + self.net_workgroups
            self.PATH_STRIDE = 520
            i64 = torch.int64
            self._ps = dict(
                path=torch.zeros((n_games, self.PATH_STRIDE), dtype=torch.int32, **kw),
                done=torch.zeros(n_games, dtype=torch.int32, **kw), roll=torch.zeros(n_games, dtype=torch.uint8, **kw),
                q_slots=torch.zeros(2 * _lib.SEARCH_QUEUE_ENTRIES * 8, dtype=i64, **kw), ctl=torch.zeros(16, dtype=torch.int32, **kw),
                rep_v=torch.zeros(n_games, dtype=i64, **kw), rep_p=torch.zeros(n_games * 64, dtype=i64, **kw),
                totals=torch.zeros(16, dtype=i64, **kw), wg_own=torch.zeros(4 * grid, dtype=i64, **kw),
                wg_opp=torch.zeros(4 * grid, dtype=i64, **kw), wg_v=torch.zeros(4 * grid, dtype=torch.float32, **kw),
                wg_probs=torch.zeros((4 * grid, 64), dtype=torch.float32, **kw))
            # position table of the value net (iago_mcts_search_args.vtable): 2^20 entries of 32 bytes, shared by the
            # games and kept across launches; zeroed when the value net's weights change
            slots = int(os.environ.get("IAGO_PERSISTENT_TABLE", str(1 << 20)))
            self._vtable = torch.zeros(4 * slots, dtype=i64, **kw) if slots > 0 else None
            self.z_log = torch.zeros((z_log_rows, n_games), dtype=torch.int8, **kw) if z_log_rows else None
            self.z_log_n = torch.zeros(n_games, dtype=torch.int32, **kw) if z_log_rows else None
            self.time_limit_ms = int(os.environ.get("IAGO_PERSISTENT_LIMIT_MS", "4000"))
        self.tree.reset_hooks = list(getattr(self.tree, "reset_hooks", ())) + [
            lambda mask: setattr(self, "_live_after_compaction", 0)]
        # Game-asynchronous steps (iago_mcts_async in include/iago_hip.h): a game whose leaf has a
        # stored value completes its playout in the step; a game whose leaf is fresh waits `parts`
        # steps while the value net walks its board piece by piece beside the other games' steps.
        # Same trees (tests/test_mcts_production_gpu.py runs both schedules against the oracle).
        # Default OFF: measured slower than lockstep playouts as long as every search ends in
        # lockstep -- a 100-playout search needs ~200 steps of ~105 us because the game with the most
        # fresh leaves sets the step count (LABNOTES.md, round 3); async_steps=True / IAGO_ASYNC=1
        # selects it.
        can_async = bool(self.lookahead and self.value_cache and getattr(self, "fused_descent", False)
                         and getattr(self, "_la_path", None) is not None and self.fused_leaf_eval
                         and 0.0 < self.lmbda < 1.0 and rollout_weights is not None and not rollout_weights.log_form
                         and getattr(value_fn, "forward_boards_async", None) is not None)
        if async_steps is None:
            async_steps = can_async and os.environ.get("IAGO_ASYNC", "0") == "1"
        if async_steps and not can_async:
            raise ValueError("async_steps needs the look-ahead playout with the value cache, the one-launch descent, "
                             "the path backup, the fused leaf evaluation (0 < lmbda < 1, product-form rollout "
                             "weights) and a value net with forward_boards_async")
        self.async_steps = bool(async_steps)
        self.n_steps = 0              # game-asynchronous steps run so far
        if self.async_steps:
            parts = int(async_parts if async_parts is not None else os.environ.get("IAGO_ASYNC_PARTS", "3"))
            if not 2 <= parts <= 4:
                raise ValueError("async_parts must be 2, 3 or 4")
            self.async_parts = parts
            self._a_wait = torch.zeros(n_games, dtype=torch.int32, **kw)
            self._a_done = torch.zeros(n_games, dtype=torch.int32, **kw)
            self._a_roll = torch.zeros(n_games, dtype=torch.uint8, **kw)
            self._a_fq_index = torch.zeros((parts, n_games), dtype=torch.int64, **kw)
            self._a_fq_count = torch.zeros(parts, dtype=torch.int32, **kw)
            self._a_step = torch.zeros(1, dtype=torch.int32, **kw)
            self._a_nsims = torch.zeros(1, dtype=torch.int32, **kw)
            self._a_scratch = torch.empty((parts, n_games, _lib.VALUE_IMAGE_BYTES), dtype=torch.uint8, **kw)
            y = _lib.MctsAsync()
            y.parts = parts
            y.wait, y.done, y.roll = self._a_wait.data_ptr(), self._a_done.data_ptr(), self._a_roll.data_ptr()
            y.fq_index, y.fq_count = self._a_fq_index.data_ptr(), self._a_fq_count.data_ptr()
            y.step, y.n_sims, y.scratch = self._a_step.data_ptr(), self._a_nsims.data_ptr(), self._a_scratch.data_ptr()
            self._async = y
            # the look-ahead state of the asynchronous steps: the lockstep one + the pointer
            self._la_async = []
            for a in self._la:
                b = _lib.MctsLookahead.from_buffer_copy(a)
                b.async_ = C.addressof(y)
                self._la_async.append(b)
            self._async_hint = {}     # n_sims -> steps the last such search needed
        self._g_own = torch.zeros(n_games, dtype=torch.int64, **kw)
        self._g_opp = torch.zeros(n_games, dtype=torch.int64, **kw)
        self._g_active = torch.zeros(n_games, dtype=torch.uint8, **kw)
        self._sim_dev = torch.zeros(1, dtype=torch.int32, **kw)  # Philox stream id on the device
        self.sim_counter = 0          # Philox stream id: one per simulation
        self.n_leaf_evals = 0
        self._n_policy_host = 0
        self.rollout_hook = None
        self._rollout_out = ops.RolloutResult()
        self._rollout_out.z = self.z

    def _search_capacity(self):
        """Workgroups of the persistent search this device keeps resident (0: the kernel cannot run here)."""
        cus, per = C.c_int32(0), C.c_int32(0)
        if _lib.lib().iago_mcts_search_capacity(C.byref(cus), C.byref(per)) != 0:
            return 0
        n = cus.value if self.max_cus <= 0 else min(self.max_cus, cus.value)
        return n * per.value

    # policy evaluations so far: counted on the host (host-counted playouts) and on the
    # device (sync-free playouts; reading it is a host sync)
    @property
    def n_policy_evals(self):
        if self.persistent:   # (+ the per-playout launches a rollout hook sends the searches through)
            return int(self._ps["totals"][1].item()) + self._n_policy_host + int(self._pend_total.item())
        return self._n_policy_host + int(self._pend_total.item())

    @property
    def n_value_evals(self):
        """Value-net evaluations executed so far: with the value cache the first visits of leaves that
        had no stored value (n_value_inline) + the rows of the value look-ahead's batches
        (n_value_ahead)."""
        return self.n_value_inline + self.n_value_ahead if self.value_cache else self.n_leaf_evals

    @property
    def n_value_inline(self):
        """Evaluations on the playouts' critical path (a leaf visited before it had a value)."""
        if self.persistent:
            return int(self._ps["totals"][0].item()) + int(self._value_total.item())
        return int(self._value_total.item()) if self.value_cache else self.n_leaf_evals

    @property
    def n_value_ahead(self):
        """Evaluations off the critical path: rows the value look-ahead's batches have evaluated (hits, duplicates and
        never-visited); with the persistent search the positions idle net workgroups walked for the position table
        ahead of their first visit (the children of expanding nodes)."""
        if self.persistent:
            return int(self._ps["totals"][11].item())
        return int(self._va_total.item()) if self.value_ahead else 0

    @n_policy_evals.setter
    def n_policy_evals(self, value):
        if value != 0:
            raise ValueError("n_policy_evals can only be reset to 0")
        self._n_policy_host = 0
        self._pend_total.zero_()
        if self.persistent:
            self._ps["totals"][1].zero_()

    def _bucket(self, n):
        """Smallest power-of-two batch >= n (min 16), capped at the pool size."""
        b = 16
        while b < n:
            b *= 2
        return min(b, self._policy_in.shape[0])

    def warmup(self):
        """Run the nets once per batch shape the search will use."""
        with torch.no_grad():
            if self.policy_fn is not None and self.sync_free:
                self._policy_counted(self._policy_in[:self.n_games], self._pend_count)
            elif self.policy_fn is not None:
                b = 16
                while True:
                    nb = min(b, self._policy_in.shape[0])
                    self.policy_fn(self._policy_in[:nb])
                    if nb == self._policy_in.shape[0]:
                        break
                    b *= 2
            if self.value_fn is not None and self.lmbda < 1.0:
                self.value_fn(self.planes)
        torch.cuda.synchronize()

    # -- one simulation = MCTS.playout for every active game (MCTS.py:105-133)
    def _select(self, own, opp, active, from_root):
        L = _lib.lib()
        check(L.iago_mcts_select(self.tree.ref(), _p(own), _p(opp), _p(active), self.c_puct,
                                 self.n_thr, 1 if from_root else 0, _p(self.cur_node),
                                 _p(self.cur_own), _p(self.cur_opp), _p(self.needs_expand),
                                 _p(self.legal), _p(self.stats) if self.stats is not None else None,
                                 _stream()), "iago_mcts_select")

    def _find_pending(self, active):
        check(_lib.lib().iago_mcts_pending(_p(self.needs_expand), _p(active), self.n_games,
                                           _p(self._pending), _p(self._pend_idx), _p(self._pend_games),
                                           _p(self._pend_count),
                                           _p(self._pend_total) if self.sync_free else None,
                                           _stream()), "iago_mcts_pending")

    def _expand_pending(self, own, opp, active):
        """Expansion branch of MCTS.playout (MCTS.py:109-121) for the games whose
        cursor sits on a leaf with n_visits >= n_thr, host-counted: one host sync."""
        L = _lib.lib()
        self._find_pending(active)
        n_exp = int(self._pend_count.item())  # the one host sync of a playout
        if n_exp == 0:
            return
        pending, idx, games = self._pending, self._pend_idx[:n_exp], self._pend_games[:n_exp]
        # MIOpen picks (and on first sight searches for) a kernel per input
        # shape: run the policy net on a few fixed bucket sizes only
        nb = self._bucket(n_exp)
        sub_planes = self._policy_in[:nb]
        ops.encode_planes_indexed(self.cur_own, self.cur_opp, idx, sub_planes)
        with torch.no_grad():
            probs = self.policy_fn(sub_planes).to(torch.float32).contiguous()
        self._n_policy_host += n_exp
        check(L.iago_mcts_expand(self.tree.ref(), _p(games), games.numel(), _p(self.cur_node),
                                 _p(self.legal), _p(probs), None, _stream()), "iago_mcts_expand")
        self._select(own, opp, pending, False)  # MCTS.py:121: recurse into the same node

    def _policy_counted(self, planes, n_dev):
        """(len(planes), 64) float32 probabilities, valid in the first *n_dev rows."""
        fc = getattr(self.policy_fn, "forward_counted", None)
        with torch.no_grad():
            if fc is not None:
                return fc(planes, n_dev)
            return self.policy_fn(planes).to(torch.float32).contiguous()

    def _expand_pending_counted(self, own, opp, active):
        """The same branch with the count left on the device: every launch is enqueued
        unconditionally and sized for n_games; items past *count exit at once."""
        L = _lib.lib()
        self._find_pending(active)
        n = self.n_games
        fb = getattr(self.policy_fn, "forward_counted_boards", None)
        if fb is not None:  # plane encoding fused into the net's first layer
            probs = fb(self.cur_own, self.cur_opp, self._pend_idx, n, self._pend_count)
        else:
            planes = self._policy_in[:n]
            ops.encode_planes_indexed(self.cur_own, self.cur_opp, self._pend_idx, planes,
                                      n_dev=self._pend_count)
            probs = self._policy_counted(planes, self._pend_count)
        check(L.iago_mcts_expand(self.tree.ref(), _p(self._pend_games), n, _p(self.cur_node),
                                 _p(self.legal), _p(probs), _p(self._pend_count), _stream()),
              "iago_mcts_expand")
        self._select(own, opp, self._pending, False)  # MCTS.py:121: recurse into the same node

    def _evaluate_and_backup(self, active, stream_id=0, stream_id_dev=None, counter=None, fresh_listed=False):
        """Leaf evaluation (MCTS.py:123-127) and Node.update_recursive."""
        L = _lib.lib()
        rolled = False  # the rollout has run inside the value net's launch
        if self.lmbda < 1.0 and self.value_cache:
            # the value net on the leaves without a stored value only (device-side list + count)
            if not fresh_listed:
                check(L.iago_mcts_fresh_leaves(self.tree.ref(), _p(active), _p(self.cur_node), _p(self._fresh_idx),
                                               _p(self._fresh_count), _p(self._value_total), _stream()),
                      "iago_mcts_fresh_leaves")
            # ... and, in the same launch, the rollouts of all leaves (two kinds of workgroups)
            both = (self.fused_leaf_eval and self.lmbda > 0.0 and self.rollout_hook is None
                    and self.rollout_weights is not None and not self.rollout_weights.log_form)
            ro = None
            if both:
                ro = ops.rollout_prepare(self.cur_own, self.cur_opp, self.rollout_weights, seed=self.seed,
                                         id_base=self.game_id_base, stream_id=stream_id,
                                         stream_id_dev=stream_id_dev, out=self._rollout_out)
            with torch.no_grad():
                self.value_fn.forward_boards_counted(self.cur_own, self.cur_opp, self._fresh_idx,
                                                     self._fresh_count, self.v, rollout=ro)
            rolled = both
        elif self.lmbda < 1.0:
            v = None
            fb = getattr(self.value_fn, "forward_boards", None)
            with torch.no_grad():
                if fb is not None:
                    v = fb(self.cur_own, self.cur_opp)  # plane encoding fused into the first layer
                if v is None:
                    ops.encode_planes(self.cur_own, self.cur_opp, out=self.planes)
                    v = self.value_fn(self.planes)
            self.v = v.to(torch.float32).contiguous()
        if self.lmbda > 0.0 and not rolled:
            ops.rollout(self.cur_own, self.cur_opp, self.rollout_weights, seed=self.seed,
                        id_base=self.game_id_base, stream_id=stream_id,
                        stream_id_dev=stream_id_dev, out=self._rollout_out)
            if self.rollout_hook is not None:
                self.rollout_hook(self)
        if self.lookahead:
            check(L.iago_mcts_mix_backup_lookahead(
                self.tree.ref(), _p(active), _p(self.cur_node), _p(self.cur_own), _p(self.cur_opp),
                _p(self.v) if self.lmbda < 1.0 else None, _p(self.z) if self.lmbda > 0.0 else None,
                self.lmbda, _p(self.leaf_value), _p(counter) if counter is not None else None,
                C.byref(self._la[self._la_cur]), _stream()), "iago_mcts_mix_backup_lookahead")
            return
        check(L.iago_mcts_mix_backup(self.tree.ref(), _p(active), _p(self.cur_node),
                                     _p(self.v) if self.lmbda < 1.0 else None,
                                     _p(self.z) if self.lmbda > 0.0 else None, self.lmbda,
                                     _p(self.leaf_value), _p(counter) if counter is not None else None,
                                     _stream()), "iago_mcts_mix_backup")

    def _playout_lookahead(self, own, opp, active, stream_id=0, stream_id_dev=None, counter=None):
        """One MCTS.playout for every active game with the priors of the expanding leaves taken
        from the look-ahead cache: select, expand, continue the descent, evaluate, back up (and
        queue the leaves that are K visits from expanding)."""
        L = _lib.lib()
        if self.fused_descent:
            # select, expand from the cache, continue, list the leaves without a value: one launch
            check(L.iago_mcts_descend(self.tree.ref(), _p(own), _p(opp), _p(active), self.c_puct, self.n_thr,
                                      _p(self.cur_node), _p(self.cur_own), _p(self.cur_opp), _p(self.legal),
                                      _p(self.stats) if self.stats is not None else None,
                                      C.byref(self._la[self._la_cur]),   # (the group's queue of expanded nodes)
                                      _p(self._fresh_idx) if self.value_cache else None,
                                      _p(self._fresh_count) if self.value_cache else None,
                                      _p(self._value_total) if self.value_cache else None, _stream()),
                  "iago_mcts_descend")
            self._evaluate_and_backup(active, stream_id=stream_id, stream_id_dev=stream_id_dev, counter=counter,
                                      fresh_listed=self.value_cache)
            return
        self._select(own, opp, active, True)
        check(L.iago_mcts_expand_cached(self.tree.ref(), _p(active), _p(self.needs_expand), _p(self.cur_node),
                                        _p(self.legal), C.byref(self._la[0]), _p(self._pending), _stream()),
              "iago_mcts_expand_cached")
        self._select(own, opp, self._pending, False)  # MCTS.py:121: recurse into the same node
        self._evaluate_and_backup(active, stream_id=stream_id, stream_id_dev=stream_id_dev, counter=counter)

    def _step_async(self, own, opp, active):
        """One game-asynchronous step for every game of `active`: descent of the games that are not
        waiting, leaf evaluation (rollouts of those games + one piece of the value net per queue of
        fresh leaves), backup of the games whose playout completes in this step."""
        L = _lib.lib()
        check(L.iago_mcts_descend(self.tree.ref(), _p(own), _p(opp), _p(active), self.c_puct, self.n_thr,
                                  _p(self.cur_node), _p(self.cur_own), _p(self.cur_opp), _p(self.legal),
                                  _p(self.stats) if self.stats is not None else None,
                                  C.byref(self._la_async[self._la_cur]),
                                  None, None, _p(self._value_total), _stream()), "iago_mcts_descend")
        ro = self.__dict__.get("_async_rollout")
        if ro is None or ro._keep[2] is not self.rollout_weights:
            # (marshalled once: the Philox stream of game g's playout is stream base + done[g], the
            # base in the device word the search sets)
            ro = self._async_rollout = ops.rollout_prepare(
                self.cur_own, self.cur_opp, self.rollout_weights, seed=self.seed, id_base=self.game_id_base,
                stream_id=0, stream_id_dev=self._sim_dev, out=self._rollout_out)
        with torch.no_grad():
            self.value_fn.forward_boards_async(self.cur_own, self.cur_opp, self.v, ro, C.byref(self._async))
        check(L.iago_mcts_mix_backup_lookahead(
            self.tree.ref(), _p(active), _p(self.cur_node), _p(self.cur_own), _p(self.cur_opp), _p(self.v), _p(self.z),
            self.lmbda, _p(self.leaf_value), None, C.byref(self._la_async[self._la_cur]), _stream()),
            "iago_mcts_mix_backup_lookahead")

    def _flush_lookahead(self, which=0):
        """The policy net on the leaves of queue `which` (one batch), its outputs into the prior
        cache; the queue is empty afterwards."""
        q = self._la_queues[which]
        n = q["own"].numel()
        probs = self.policy_fn.forward_counted_boards(q["own"], q["opp"], None, n, q["count"])
        check(_lib.lib().iago_mcts_store_priors(C.byref(self._la[which]), _p(probs), _p(self._pend_total),
                                                _stream()), "iago_mcts_store_priors")
        q["count"].zero_()

    def _flush_value_ahead(self, which, rows_event=None):
        """The value look-ahead's batch for the nodes queue `which` holds: one row per child without a
        value, the value net on the rows, the results into the children's records.  rows_event: recorded
        once the queue has been consumed (the next group may then append to it)."""
        L = _lib.lib()
        va, rows = self._va[which], self._va_rows
        self._va_row_count.zero_()
        check(L.iago_mcts_value_ahead_rows(self.tree.ref(), C.byref(va), _stream()), "iago_mcts_value_ahead_rows")
        self._va_x[which]["count"].zero_()
        if rows_event is not None:
            rows_event.record()
        with torch.no_grad():
            self.value_fn.forward_boards_batch(rows["own"], rows["opp"], self._va_row_count, rows["v"],
                                               self.va_boards, self.va_grid)
        check(L.iago_mcts_value_ahead_store(self.tree.ref(), C.byref(va), _stream()), "iago_mcts_value_ahead_store")

    def _lookahead_block(self, own, opp, active, stream_ids, async_=False, last=True):
        """Two groups of K playouts.  On entry queue 1 may hold the leaves of the previous block's
        second group and queue 0 is empty; on exit the same.  With lookahead_overlap = j > 0 the
        batch of the previous group runs on the side stream beside the first j playouts of a group
        (its leaves were queued early enough for that); with 0 every group ends with its own
        batch on the one stream.  stream_ids: None (the device word, graph capture / replay) or an
        iterator of 2 K Philox stream ids."""
        K, j = self.lookahead, self.lookahead_overlap
        main = torch.cuda.current_stream()
        va = self.value_ahead
        for grp in (0, 1):
            self._la_cur = grp
            if j:
                self._la_side.wait_stream(main)
                with torch.cuda.stream(self._la_side):
                    self._flush_lookahead(1 - grp)
                    if va:
                        # the value batch of the previous group's expansions goes on beside this whole
                        # group; the playouts only wait for the priors (playout j) and, before the next
                        # group appends to the queue, for the rows kernel that consumes it
                        self._ev_priors.record()
                        self._flush_value_ahead(1 - grp, self._ev_rows)
            for i in range(K):
                if j and i == j:
                    if va:
                        main.wait_event(self._ev_priors)
                    else:
                        main.wait_stream(self._la_side)
                if async_:
                    self._step_async(own, opp, active)
                elif stream_ids is None:
                    self._playout_lookahead(own, opp, active, stream_id=0, stream_id_dev=self._sim_dev,
                                            counter=self._sim_dev)
                else:
                    self._playout_lookahead(own, opp, active, stream_id=next(stream_ids))
            if not j:
                self._flush_lookahead(grp)
            if va:
                main.wait_event(self._ev_rows)
        if va and last:
            main.wait_stream(self._la_side)   # (a captured graph ends with every stream joined)
        self._la_cur = 0

    def _lookahead_tail(self, own, opp, active, n, stream_ids):
        """The rest of a search after its whole blocks: queue 1's batch, then n < 2 K playouts with
        a batch after every K of them and at the end (one stream); both queues end empty."""
        if self.lookahead_overlap:
            self._flush_lookahead(1)
        if self.value_ahead:
            self._flush_value_ahead(1)
        self._la_cur = 0
        for i in range(n):
            if stream_ids is None:
                self._playout_lookahead(own, opp, active, stream_id=0, stream_id_dev=self._sim_dev,
                                        counter=self._sim_dev)
            else:
                self._playout_lookahead(own, opp, active, stream_id=next(stream_ids))
            if (i + 1) % self.lookahead == 0 or i + 1 == n:
                self._flush_lookahead(0)
                if self.value_ahead:
                    self._flush_value_ahead(0)

    def simulate(self, own, opp, active, n_active=None):
        """One MCTS.playout for every active game (eager launches)."""
        if self.lookahead:
            self._la_cur = 0
            self._playout_lookahead(own, opp, active, stream_id=self.sim_counter)
            self._flush_lookahead(0)  # a lone playout flushes at once: the queues are empty between calls
            if self.value_ahead:
                self._flush_value_ahead(0)
            self.sim_counter = (self.sim_counter + 1) & 0xFFFFFFFF
            if n_active is not None:
                self.n_leaf_evals += n_active
            return
        self._select(own, opp, active, True)
        if self.sync_free:
            self._expand_pending_counted(own, opp, active)
        else:
            self._expand_pending(own, opp, active)
        self._evaluate_and_backup(active, stream_id=self.sim_counter)
        self.sim_counter = (self.sim_counter + 1) & 0xFFFFFFFF
        if n_active is not None:
            self.n_leaf_evals += n_active

    # -- hipGraph mode: the whole sync-free playout is captured once and replayed with a
    # single launch per playout; the rollout's Philox stream id is a device word the graph
    # itself increments.
    def _graph_state(self):
        """What the captured graph baked in: device pointers and versions of every weight
        (and of the layouts cached from them), the rollout table, the scalar arguments."""
        key = [self.lmbda, self.c_puct, self.n_thr, self.lookahead, self.lookahead_overlap, self.value_cache,
               self.async_steps, self.value_ahead, self.persistent,
               getattr(self, "fused_descent", False), self.fused_leaf_eval,
               self.stats.data_ptr() if self.stats is not None else 0,
               self.rollout_weights.table.data_ptr() if self.rollout_weights is not None else 0]
        for fn in (self.policy_fn, self.value_fn):
            params = getattr(fn, "parameters", None)
            if params is not None:
                key.extend((q.data_ptr(), q._version) for q in params())
                key.append(bool(getattr(fn, "training", False)))
                key.append(bool(getattr(fn, "split_f16", False)))
                key.append(bool(getattr(fn, "fused", False)))
                key.append((getattr(fn, "split3", None), getattr(fn, "split3_parts", None)))
            else:
                key.append(id(fn))
        return tuple(key)

    def close(self):
        """Drop the captured graphs (and with them their private memory pools) now.  An engine
        holds reference cycles (the trees' reset hooks), so without this its graphs live until the
        cyclic garbage collector runs -- which must not happen while ANOTHER engine captures:
        destroying a graph is not permitted while a stream of the process is capturing."""
        self._graph = self._graph_long = self._graph_key = None
        self._scratch_refs = None

    def _capture(self):
        """Record the playout launches into hipGraphs.  The cyclic garbage collector is held
        off for the duration: a collected torch.cuda.CUDAGraph of some other, unreferenced engine
        would be destroyed inside the capture, which HIP refuses (hipErrorStreamCaptureUnsupported)."""
        import gc
        self._graph = self._graph_long = None
        self._scratch_refs = None
        gc.collect()
        # (this engine's old graphs are gone and with them its hold on the scratch buffers of the
        # policy's multi-launch forward: the module drops its own references -- 205 MB per calling
        # stream that a re-capture-per-update loop would otherwise pile up -- and the warm-up below
        # allocates what the new capture needs.  Another engine that shares the module keeps the
        # buffers ITS graphs address alive through its own _scratch_refs)
        rel = getattr(self.policy_fn, "release_scratch", None)
        if rel is not None:
            rel()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_graphs()
            pool = getattr(self.policy_fn, "__dict__", {}).get("_split3_scratch_pool", {})
            self._scratch_refs = [b for k, b in pool.items() if k != "retired"] + list(pool.get("retired", ()))
        finally:
            if was_enabled:
                gc.enable()

    def _capture_graphs(self):
        if self.rollout_hook is not None:
            raise ValueError("rollout_hook is not available in graph mode")
        # one eager evaluation of both nets first: lazy one-time setup (kernel attributes,
        # MIOpen's choice for this shape, weight layouts cached per weight version) must not
        # happen under capture.  Neither touches the trees.
        if self.lmbda < 1.0 and self.value_cache:
            # (through the very entry point the capture records -- iago_value_rollout when the leaf
            # evaluation is fused -- with an empty fresh list: the value rows do nothing, the
            # rollouts write self.z, which every playout overwrites before it is read)
            self._fresh_count.zero_()
            both = (self.fused_leaf_eval and self.lmbda > 0.0 and self.rollout_weights is not None
                    and not self.rollout_weights.log_form)
            ro = ops.rollout_prepare(self.cur_own, self.cur_opp, self.rollout_weights, seed=self.seed,
                                     id_base=self.game_id_base, stream_id=0, stream_id_dev=self._sim_dev,
                                     out=self._rollout_out) if both else None
            with torch.no_grad():
                self.value_fn.forward_boards_counted(self.cur_own, self.cur_opp, self._fresh_idx,
                                                     self._fresh_count, self.v, rollout=ro)
        elif self.lmbda < 1.0:
            ops.encode_planes(self.cur_own, self.cur_opp, out=self.planes)
            with torch.no_grad():
                fb = getattr(self.value_fn, "forward_boards", None)
                if fb is None or fb(self.cur_own, self.cur_opp) is None:
                    self.value_fn(self.planes)
        if self.policy_fn is not None:
            # the entry point _expand_pending_counted / _flush_lookahead take: the board-fed forward
            # (three-piece kernel: weight split, argument template, scratch buffer, kernel
            # attributes) when the policy has one, the planes-fed float32 kernels otherwise
            self._pend_count.zero_()
            fb = getattr(self.policy_fn, "forward_counted_boards", None)
            if fb is not None:
                fb(self.cur_own, self.cur_opp, self._pend_idx, self.n_games, self._pend_count)
            else:
                self._policy_counted(self._policy_in[:self.n_games], self._pend_count)
        torch.cuda.synchronize()
        if self.lookahead:
            self._flush_lookahead(0)  # (queues empty: allocations and one-time setup only)
            self._flush_lookahead(1)
            if self.value_ahead:
                self._flush_value_ahead(0)
                with torch.cuda.stream(self._la_side):
                    self._flush_value_ahead(1)
            torch.cuda.synchronize()
        if self.async_steps:
            # the asynchronous step's own entry points, once, on empty queues and no game rolled
            self._a_wait.zero_()
            self._a_done.zero_()
            self._a_roll.zero_()
            self._a_fq_count.zero_()
            self._a_nsims.zero_()
            self._step_async(self._g_own, self._g_opp, torch.zeros_like(self._g_active))
            torch.cuda.synchronize()
        self._graph_long = None
        if self.lookahead and self.graph_blocks > 1:
            # the same block several times over: a replay costs tens of microseconds on the
            # device whatever it holds, so long searches replay the long graph and finish with
            # the one-block graph
            self._graph_long = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_long):
                for b in range(self.graph_blocks):
                    self._lookahead_block(self._g_own, self._g_opp, self._g_active, None, async_=self.async_steps,
                                          last=b + 1 == self.graph_blocks)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            if self.lookahead:
                # 2 K playouts (or steps) and the policy batches of the leaves they queue: ONE replay
                self._lookahead_block(self._g_own, self._g_opp, self._g_active, None, async_=self.async_steps)
            else:
                self._select(self._g_own, self._g_opp, self._g_active, True)
                self._expand_pending_counted(self._g_own, self._g_opp, self._g_active)
                self._evaluate_and_backup(self._g_active, stream_id=0, stream_id_dev=self._sim_dev,
                                          counter=self._sim_dev)

    def _search_graph(self, own, opp, active, n_sims, n_active):
        key = self._graph_state()
        if self._graph is None or key != self._graph_key:
            # first use, or the weights / rollout table changed since the capture (training
            # between searches, load_npz): the old graph holds pointers to freed layouts
            self._graph = None
            self._capture()
            self._graph_key = key
        self._g_own.copy_(own)
        self._g_opp.copy_(opp)
        self._g_active.copy_(active)
        self._sim_dev.fill_(self.sim_counter - (1 << 32) if self.sim_counter >= (1 << 31)
                            else self.sim_counter)
        if self.lookahead:
            block = 2 * self.lookahead
            n_blocks = n_sims // block
            if self._graph_long is not None:
                for _ in range(n_blocks // self.graph_blocks):
                    self._graph_long.replay()
                n_blocks %= self.graph_blocks
            for _ in range(n_blocks):
                self._graph.replay()
            # the same launches, not captured
            self._lookahead_tail(self._g_own, self._g_opp, self._g_active, n_sims % block, None)
        else:
            for _ in range(n_sims):
                self._graph.replay()
        self.sim_counter = (self.sim_counter + n_sims) & 0xFFFFFFFF
        self.n_leaf_evals += n_active * n_sims

    def _search_async(self, own, opp, active, n_sims, n_active):
        """n_sims playouts per active game as game-asynchronous steps: blocks of 2 K steps (one
        graph replay each, or the same launches eagerly) until every game has completed its
        playouts.  The number of steps depends on how often the games met fresh leaves, so the
        host reads one word back after the expected number of blocks and adds blocks while a
        game is behind; steps after the last game has finished do nothing."""
        if self.use_graph:
            key = self._graph_state()
            if self._graph is None or key != self._graph_key:
                self._graph = None
                self._capture()
                self._graph_key = key
            self._g_own.copy_(own)
            self._g_opp.copy_(opp)
            self._g_active.copy_(active)
            own, opp, act = self._g_own, self._g_opp, self._g_active
        else:
            act = active
        self._a_wait.zero_()
        self._a_done.zero_()
        self._a_roll.zero_()
        self._a_fq_count.zero_()
        self._a_nsims.fill_(n_sims)
        self._sim_dev.fill_(self.sim_counter - (1 << 32) if self.sim_counter >= (1 << 31) else self.sim_counter)
        block = 2 * self.lookahead
        want = self._async_hint.get(n_sims, n_sims + (self.async_parts - 1) * (n_sims // 5 + 1))
        steps = 0
        checks = 0
        while True:
            n_blocks = max(1, -(-(want - steps) // block))
            steps += n_blocks * block
            if self.use_graph:
                if self._graph_long is not None:
                    for _ in range(n_blocks // self.graph_blocks):
                        self._graph_long.replay()
                    n_blocks %= self.graph_blocks
                for _ in range(n_blocks):
                    self._graph.replay()
            else:
                for _ in range(n_blocks):
                    self._lookahead_block(own, opp, act, None, async_=True)
            checks += 1
            behind = bool(((self._a_done < n_sims) & (act != 0)).any().item())   # the search's host sync
            if not behind:
                break
            want = steps + block
        # one check: the estimate was enough (try one block less next time); more: remember the need
        self._async_hint[n_sims] = max(n_sims, steps - block) if checks == 1 else steps
        self.n_steps += steps
        self._lookahead_tail(own, opp, act, 0, None)   # the last group's policy batch: both queues end empty
        self.sim_counter = (self.sim_counter + n_sims) & 0xFFFFFFFF
        self.n_leaf_evals += n_active * n_sims

    def _search_persistent(self, own, opp, active, n_sims, n_active):
        """n_sims playouts per active game as ONE launch (iago_mcts_search_persistent)."""
        self._launch_persistent(own, opp, active, n_sims)
        self.sim_counter = (self.sim_counter + n_sims) & 0xFFFFFFFF
        self.n_leaf_evals += n_active * n_sims

    def _launch_persistent(self, own, opp, active, n_sims, game=None):
        """iago_mcts_search_persistent: one search from the roots (own, opp), or -- game = dict(max_turns, own,
        opp, n_turns, rec_own, rec_opp, rec_valid, rec_move, rec_pi) -- whole self-play games."""
        if self.rollout_hook is not None:
            raise ValueError("rollout_hook is not available in the persistent search (z_log_rows records the z)")
        ps = self._ps
        sid = self.sim_counter - (1 << 32) if self.sim_counter >= (1 << 31) else self.sim_counter
        ro = ops.rollout_prepare(self.cur_own, self.cur_opp, self.rollout_weights, seed=self.seed,
                                 id_base=self.game_id_base, stream_id=sid, out=self._rollout_out)
        with torch.no_grad():
            va, keep_v = self.value_fn.search_args(ps["wg_own"], ps["wg_opp"], ps["wg_v"])
            pa, keep_p = self.policy_fn.search_args(ps["wg_own"], ps["wg_opp"], ps["wg_probs"])
        a = _lib.MctsSearchArgs()
        a.tree = C.addressof(self.tree.c)
        if own is not None:
            a.root_own, a.root_opp = own.data_ptr(), opp.data_ptr()
        a.active = active.data_ptr()
        a.c_puct, a.lmbda, a.n_thr, a.n_sims = self.c_puct, self.lmbda, self.n_thr, int(n_sims)
        a.net_workgroups, a.time_limit_ms = self.net_workgroups, self.time_limit_ms
        a.games_per_workgroup = self.games_per_workgroup
        a.pace_margin = self.pace_margin
        a.max_cus = self.max_cus
        a.value, a.policy, a.rollout = C.addressof(va), C.addressof(pa), C.addressof(ro.args)
        a.cur_node, a.cur_own, a.cur_opp = self.cur_node.data_ptr(), self.cur_own.data_ptr(), self.cur_opp.data_ptr()
        a.path, a.path_stride = ps["path"].data_ptr(), self.PATH_STRIDE
        a.done, a.roll, a.leaf_value = ps["done"].data_ptr(), ps["roll"].data_ptr(), self.leaf_value.data_ptr()
        if self.z_log is not None:
            a.z_log, a.z_log_n, a.z_log_rows = self.z_log.data_ptr(), self.z_log_n.data_ptr(), self.z_log.shape[0]
        a.q_slots, a.ctl = ps["q_slots"].data_ptr(), ps["ctl"].data_ptr()
        a.rep_v, a.rep_p, a.totals = ps["rep_v"].data_ptr(), ps["rep_p"].data_ptr(), ps["totals"].data_ptr()
        a.stats = self.stats.data_ptr() if self.stats is not None else None
        a.wg_own, a.wg_opp = ps["wg_own"].data_ptr(), ps["wg_opp"].data_ptr()
        if game is not None:
            a.max_turns = int(game["max_turns"])
            # (a whole game of 400-playout searches takes seconds; IAGO_PERSISTENT_GAME_LIMIT_MS: the lab's shorter limit)
            a.time_limit_ms = max(self.time_limit_ms, int(os.environ.get("IAGO_PERSISTENT_GAME_LIMIT_MS", "60000")))
            a.game_own, a.game_opp, a.n_turns = game["own"].data_ptr(), game["opp"].data_ptr(), game["n_turns"].data_ptr()
            a.rec_own, a.rec_opp = game["rec_own"].data_ptr(), game["rec_opp"].data_ptr()
            a.rec_valid, a.rec_move, a.rec_pi = (game["rec_valid"].data_ptr(), game["rec_move"].data_ptr(),
                                                 game["rec_pi"].data_ptr())
        if self._vtable is not None:
            a.vtable, a.vtable_slots = self._vtable.data_ptr(), self._vtable.numel() // 4
        if getattr(self, "trace", None) is not None:   # (diagnostic: tools/exp_persistent_trace.py)
            a.trace, a.trace_rows = self.trace.data_ptr(), self.trace.shape[0]
        ev = getattr(self, "launch_events", None)   # (bench.py: HIP event pairs around the launches, on their stream)
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._split is not None:
            check(_lib.lib().iago_mcts_search_split(C.byref(a), self._split, _stream()), "iago_mcts_search_split")
        else:
            check(_lib.lib().iago_mcts_search_persistent(C.byref(a), _stream()), "iago_mcts_search_persistent")
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        self._ps_keep = (keep_v, keep_p, ro, va, pa, own, opp, active, game)   # alive until the next launch

    def search_counts(self, active):
        """Device tensor int64[2]: games in `active`, nodes of the fullest pool -- what search()
        reads back before it starts (a caller that batches its readbacks passes them in)."""
        return torch.stack([active.sum().to(torch.int64), self.tree.n_nodes.max().to(torch.int64)])

    def search(self, own, opp, active, n_sims, counts=None, check=True):
        """n_sims playouts from the current roots; (own, opp) = root positions
        with own = side to move; active: uint8 mask of participating games.
        counts: (games in `active`, nodes of the fullest pool) when the caller has read
        search_counts() back already; check=False: the error flags are not read back here (one
        host sync each) -- the caller reads error_flags() and calls raise_errors()."""
        n_active, used = (int(v) for v in (self.search_counts(active).tolist() if counts is None else counts))
        if n_active == 0:
            # (the playout counter advances all the same: a game's Philox streams are keyed by ITS turn and
            # playout, whatever the other games of the batch do at that turn)
            self.sim_counter = (self.sim_counter + n_sims) & 0xFFFFFFFF
            return
        if self.value_cache:
            # the stored values belong to the weights that computed them
            key = tuple((q.data_ptr(), q._version) for q in self.value_fn.parameters())
            if key != self._value_key:
                if self._value_key is not None:
                    self.tree.v.fill_(float("nan"))
                    if getattr(self, "_vtable", None) is not None:
                        self._vtable.zero_()
                self._value_key = key
            # (the descent appends to the fresh-leaf list through this count and the backup clears
            # it: a playout aborted between the two must not leave a stale count behind)
            self._fresh_count.zero_()
        if used > self.tree.capacity // 2 and used > self._live_after_compaction * 5 // 4:
            # a pool is half full: free the nodes that subtree reuse left behind (what the
            # reference's garbage collector does after MCTS.py:149-152) before this search adds
            # its own.  A pool that fills up all the same is reported below.  (Not again until
            # the pool has grown by a quarter over what the last pass left: a live tree that
            # itself fills half the pool would otherwise be re-laid before every search.)
            self.tree.compact()
            self.n_compactions += 1
            self._live_after_compaction = int(self.tree.n_nodes.max().item())
        if self.persistent and self.rollout_hook is None:
            self._search_persistent(own, opp, active, n_sims, n_active)
        elif self.async_steps and self.rollout_hook is None:
            self._search_async(own, opp, active, n_sims, n_active)
        elif self.use_graph:
            self._search_graph(own, opp, active, n_sims, n_active)
        elif self.lookahead:
            block = 2 * self.lookahead
            ids = iter([(self.sim_counter + i) & 0xFFFFFFFF for i in range(n_sims)])
            for _ in range(n_sims // block):
                self._lookahead_block(own, opp, active, ids)
            self._lookahead_tail(own, opp, active, n_sims % block, ids)
            self.sim_counter = (self.sim_counter + n_sims) & 0xFFFFFFFF
            self.n_leaf_evals += n_active * n_sims
        else:
            for _ in range(n_sims):
                self.simulate(own, opp, active, n_active)
        if check:
            self.raise_errors(self.error_flags().tolist())

    def error_flags(self):
        """Device tensor int64[5]: pools that overflowed, the look-ahead's error word, the saturation
        flags of the value and the policy net (0 where a net has none), the persistent search's
        gave-up word."""
        dev = self.cur_own.device
        zero = torch.zeros((), dtype=torch.int64, device=dev)
        parts = [self.tree.overflow.sum().to(torch.int64),
                 self._la_error[0].to(torch.int64) if self.lookahead else zero]
        for fn in (self.value_fn, self.policy_fn):
            f = getattr(fn, "__dict__", {}).get("_ovf") if hasattr(fn, "check_saturation") else None
            parts.append(f.reshape(-1)[0].to(torch.int64) if f is not None and f.device == dev else zero)
        parts.append(self._ps["ctl"][3].to(torch.int64) if self.persistent else zero)
        return torch.stack(parts)

    def raise_errors(self, flags):
        """The errors of a search from the host copy of error_flags()."""
        overflow, err, sat_v, sat_p, gave_up = (int(x) for x in flags)
        if gave_up:
            raise _lib.IagoError("the persistent search gave up at its clock limit (%d ms for a search, 60 s for whole games: a "
                                 "reply never came -- is another job on the device, or fewer CUs free than workgroups?%s); the "
                                 "trees are incomplete"
                                 % (self.time_limit_ms, "  This was the role split (two launches on CU-masked streams): "
                                    "split=0 / IAGO_SEARCH_SPLIT=0 selects the single launch" if self._split is not None else ""))
        if overflow != 0:
            raise _lib.IagoError("MCTS node pool exhausted (or a search path deeper than 512): "
                                 "raise `capacity` (%d nodes per game)" % self.tree.capacity)
        if err:
            self._la_error.zero_()
            raise _lib.IagoError("policy look-ahead: %s" % (
                "the queue overflowed" if err == 1 else
                "a leaf reached n_thr without cached priors (raise `lookahead_slots`, now %d per "
                "game; the look-ahead must be on from the reset of the trees and n_thr must "
                "not change)" % self._la[0].slots))
        for fn, flag in ((self.value_fn, sat_v), (self.policy_fn, sat_p)):
            if flag:
                fn.check_saturation()  # raises (and clears the flag): the split-f16 kernels clamp at 65000

    def enable_stats(self):
        self.stats = torch.zeros((self.n_games, 2), dtype=torch.int32, device=self.cur_own.device)

    def tree_bytes(self):
        """Algorithmic bytes moved on the tree arrays so far, for the 32-byte
        node records: the descent reads one record per level (the node's header) and one per child
        scored; the backup reads and writes (n_visits, Q) = 8 + 8 B per node of the path (the levels
        + the leaf itself)."""
        lv, ch = (int(x) for x in self.stats.to(torch.int64).sum(dim=0).tolist())
        sel = 32 * (lv + ch) + 32 * self.n_leaf_evals
        bak = 16 * (lv + self.n_leaf_evals)
        return {"select": sel, "backup": bak, "levels": lv, "children_scored": ch}

    def memory_bytes(self):
        """Device memory this engine holds, by part: the tree pools (32-byte node records;
        twice that once compact() has allocated its second pool), the look-ahead's
        prior cache ([game][slot][64] float32) and queues, the recorded paths, and the policy
        net's scratch for its multi-launch forward (network.SLPolicy.SPLIT3_SCRATCH_ROWS x 51,200 B
        = 205 MB per stream that calls it -- the search uses up to three: eager, capture, side
        stream -- bounded whatever n_games is; longer batches run in chunks)."""
        out = {"tree": self.tree.bytes()}
        if getattr(self.tree, "_scratch", None) is not None:
            out["tree_compaction_pool"] = out["tree"] + self.tree._order.numel() * 4
        if self.lookahead:
            out["prior_cache"] = self._la_cache.numel() * 4 + self._la_cache_seq.numel() * 4
            out["queues"] = sum(t.numel() * t.element_size() for q in self._la_queues for t in q.values())
            if self._la_path is not None:
                out["paths"] = self._la_path.numel() * 4
        if self.persistent:
            out["persistent_search"] = sum(t.numel() * t.element_size() for t in self._ps.values())
            if self._vtable is not None:
                out["position_table"] = self._vtable.numel() * 8
        if self.value_ahead:
            out["value_ahead"] = sum(t.numel() * t.element_size() for d in self._va_x + [self._va_rows] for t in d.values())
        pool = getattr(self.policy_fn, "__dict__", {}).get("_split3_scratch_pool", {})
        out["policy_scratch"] = sum(b.numel() for k, b in pool.items() if k != "retired") + \
            sum(b.numel() for b in pool.get("retired", ()))
        return out

    def best_move(self, active=None, want_visits=True):
        """argmax visit count of the root's children, first wins (MCTS.py:147)."""
        check(_lib.lib().iago_mcts_best_move(self.tree.ref(),
                                             _p(active) if active is not None else None,
                                             _p(self.move), _p(self.visits) if want_visits else None,
                                             _stream()), "iago_mcts_best_move")
        return self.move, self.visits

    def update_with_move(self, move, mask=None):
        """MCTS.update_with_move (MCTS.py:149-154); move int8 tensor, -1 = pass."""
        check(_lib.lib().iago_mcts_advance_root(self.tree.ref(),
                                                _p(mask) if mask is not None else None, _p(move),
                                                _stream()), "iago_mcts_advance_root")


class SelfPlayResult(object):
    """Training tuples of one self-play round, device resident.

    own/opp: (T, B) int64 positions before each searched move (own = mover),
    pi: (T, B, 64) int32 root visit counts, valid: (T, B) uint8 (the game moved
    at that turn), move: (T, B) int8, z: (B,) int8 result from colour 1's view,
    mover: (T,) colour to move at that turn (1 or 2)."""

    def tuples(self):
        """Flat (s, pi, z) rows of all searched moves; z from the mover's view."""
        m = self.valid.reshape(-1).bool()
        T, B = self.valid.shape
        sign = torch.tensor([1 if c == 1 else -1 for c in self.mover], dtype=torch.int8,
                            device=self.z.device).reshape(T, 1)
        zz = (self.z.reshape(1, B) * sign).reshape(-1)
        dev = self.z.device
        colour = torch.tensor(list(self.mover), dtype=torch.int8, device=dev).reshape(T, 1).expand(T, B)
        # (global game id, turn) of a row: the key that puts the gathered rows of any number of
        # ranks into ONE canonical order (train_rl.ReinforceTrainer.step_from_tuples)
        game = (torch.arange(B, dtype=torch.int32, device=dev) + int(getattr(self, "game_id_base", 0))).reshape(1, B)
        turn = torch.arange(T, dtype=torch.int32, device=dev).reshape(T, 1)
        return dict(own=self.own.reshape(-1)[m], opp=self.opp.reshape(-1)[m],
                    pi=self.pi.reshape(-1, 64)[m], z=zz[m], move=self.move.reshape(-1)[m],
                    colour=colour.reshape(-1)[m], game=game.expand(T, B).reshape(-1)[m],
                    turn=turn.expand(T, B).reshape(-1)[m])


class SelfPlayEngine(object):
    """Lockstep PV-MCTS self-play: both colours search the shared tree, moves
    are the most visited children, passes advance the tree with -1
    (game.py:117-142 turn structure, both sides driven by MCTS.get_move)."""

    def __init__(self, mcts, max_turns=_lib.IAGO_MAX_TURNS):
        self.mcts = mcts
        self.B = mcts.n_games
        self.max_turns = max_turns

    def _play_persistent(self, n_sims, own, opp, record):
        """The whole game of every board in ONE launch (iago_mcts_search_persistent with max_turns > 0): each
        game walks through its own turns -- search, most visited move, update_with_move, the stone, the books
        -- with no barrier between the games' turns.  Same moves, visit counts and results as the turn-by-turn
        loop below (tests/test_search_persistent_gpu.py)."""
        m, B, T = self.mcts, self.B, self.max_turns
        dev = own.device
        g = dict(max_turns=T, own=own, opp=opp, n_turns=torch.zeros(B, dtype=torch.int32, device=dev),
                 rec_own=torch.zeros((T, B), dtype=torch.int64, device=dev),
                 rec_opp=torch.zeros((T, B), dtype=torch.int64, device=dev),
                 rec_valid=torch.zeros((T, B), dtype=torch.uint8, device=dev),
                 rec_move=torch.full((T, B), -1, dtype=torch.int8, device=dev),
                 rec_pi=torch.zeros((T, B, 64), dtype=torch.int32, device=dev))
        active = torch.ones(B, dtype=torch.uint8, device=dev)
        if m.value_cache:
            key = tuple((q.data_ptr(), q._version) for q in m.value_fn.parameters())
            if key != m._value_key:
                if m._value_key is not None:
                    m.tree.v.fill_(float("nan"))
                    if m._vtable is not None:
                        m._vtable.zero_()
                m._value_key = key
        # (what the launch accumulates into, in case a pool fills up and the batch is replayed turn by turn)
        keep = [(t, t.clone()) for t in (m._ps["totals"], m.z_log_n, m.stats) if t is not None]
        m._launch_persistent(None, None, active, n_sims, game=g)
        back = torch.cat([m.error_flags(), m._ps["ctl"][4].to(torch.int64).reshape(1),
                          g["n_turns"].max().to(torch.int64).reshape(1),
                          g["rec_valid"].sum().to(torch.int64).reshape(1),
                          m._ps["ctl"][7].to(torch.int64).reshape(1)]).tolist()
        m.net_workgroups_launched = int(back[8])
        if back[0] and not back[4]:
            # a pool filled up (the launch cannot compact): nothing of this attempt counts
            for t, was in keep:
                t.copy_(was)
            return None
        m.raise_errors(back[:5])
        if back[5]:
            raise ValueError("a searched root has no children: n_sims is below the expansion threshold n_thr")
        t = int(back[6])
        m.sim_counter = (m.sim_counter + t * n_sims) & 0xFFFFFFFF
        m.n_leaf_evals += int(back[7]) * n_sims
        res = SelfPlayResult()
        res.game_id_base = m.game_id_base
        res.n_turns = t
        res.mover = [1 if k % 2 == 0 else 2 for k in range(t)]
        # a game's boards after n_turns[g] swaps of sides; colour 1's stones are `own` after an even number
        even = (g["n_turns"] % 2 == 0)
        p1, p2 = torch.where(even, own, opp), torch.where(even, opp, own)
        res.z = ops.judge(p1, p2)
        res.final_p1, res.final_p2 = p1, p2
        res.game_turns = g["n_turns"]    # (B,) the turn at which each game ended (the batch's last: n_turns)
        if record:
            turn = torch.arange(t, device=dev).reshape(t, 1)
            played = turn < g["n_turns"].reshape(1, B)
            # (the turn-by-turn loop records a finished game's boards, still swapping sides, until the last game
            # of the batch is over: the same rows here)
            tw = (turn % 2 == 0)
            res.own = torch.where(played, g["rec_own"][:t], torch.where(tw, p1.reshape(1, B), p2.reshape(1, B)))
            res.opp = torch.where(played, g["rec_opp"][:t], torch.where(tw, p2.reshape(1, B), p1.reshape(1, B)))
            res.valid, res.move, res.pi = g["rec_valid"][:t], g["rec_move"][:t], g["rec_pi"][:t]
        return res

    def play(self, n_sims, handicap=None, record=True):
        m, B = self.mcts, self.B
        dev = m.cur_own.device
        own = torch.full((B,), START_OWN, dtype=torch.int64, device=dev)
        opp = torch.full((B,), START_OPP, dtype=torch.int64, device=dev)
        if handicap is not None:  # (B,) int64 bit masks of extra colour-2 stones
            opp = opp | handicap
        m.tree.reset()
        # The one-launch whole-game path wherever the persistent search applies and the pools can hold a whole game
        # without compaction; else -- and when a pool fills up all the same -- the turn-by-turn loop below, whose
        # searches (one persistent launch per turn) compact a pool that is half full.  Same games either way.
        if (getattr(m, "persistent", False) and m.rollout_hook is None
                and os.environ.get("IAGO_PERSISTENT_GAMES", "1") != "0"
                and 2 * m.tree.capacity >= suggest_capacity(n_sims, m.n_thr, moves=min(self.max_turns, 64))):
            res = self._play_persistent(n_sims, own.clone(), opp.clone(), record)
            if res is not None:
                return res
            self.n_replayed = getattr(self, "n_replayed", 0) + 1   # batches replayed through the turn loop
            m.tree.reset()
        stone_num = torch.full((B,), 4, dtype=torch.int32, device=dev)  # game.py:32
        pass_flg = torch.zeros(B, dtype=torch.uint8, device=dev)
        done = torch.zeros(B, dtype=torch.uint8, device=dev)
        res = SelfPlayResult()
        res.game_id_base = m.game_id_base
        T = self.max_turns
        if record:
            res.own = torch.zeros((T, B), dtype=torch.int64, device=dev)
            res.opp = torch.zeros((T, B), dtype=torch.int64, device=dev)
            res.pi = torch.zeros((T, B, 64), dtype=torch.int32, device=dev)
            res.valid = torch.zeros((T, B), dtype=torch.uint8, device=dev)
            res.move = torch.full((T, B), -1, dtype=torch.int8, device=dev)
        res.mover = []
        legal = ops.legal_moves(own, opp)
        active = (legal != 0).to(torch.uint8)
        legal_next, active_next = torch.empty_like(legal), torch.empty_like(active)
        t = 0
        counts = m.search_counts(active).tolist()
        while t < T:
            # ONE readback per move (below): the flags of this move's search, the check of its
            # moves, the end-of-game test and the counts the next search starts from
            m.search(own, opp, active, n_sims, counts=counts, check=False)
            move, visits = m.best_move(active)
            mv = torch.where(active.bool(), move, torch.full_like(move, -1))
            if record:
                res.own[t], res.opp[t], res.valid[t], res.move[t] = own, opp, active, mv
                res.pi[t] = visits * active.reshape(B, 1).to(torch.int32)
            res.mover.append(1 if t % 2 == 0 else 2)
            m.update_with_move(mv, done ^ 1)  # game.py:84,108,140 (the games not yet done)
            # the move, stone_num / pass_flg, `while game.stone_num < 64` once per pair of turns
            # (game.py:117-142,253-255), the swap of sides, the next mover's legal moves
            ops.play_turn(own, opp, mv, active, stone_num, pass_flg, done, t % 2 == 1, legal_next, active_next)
            legal, legal_next = legal_next, legal
            active, active_next = active_next, active
            t += 1
            back = torch.cat([m.error_flags(), (mv == -2).any().to(torch.int64).reshape(1),
                              done.all().to(torch.int64).reshape(1), m.search_counts(active)]).tolist()
            m.raise_errors(back[:5])
            back = back[1:]
            if back[4]:
                # what max() over an empty children dict raises in MCTS.get_move (MCTS.py:147)
                raise ValueError("a searched root has no children: n_sims is below the "
                                 "expansion threshold n_thr")
            if t % 2 == 0 and back[5]:
                break
            counts = back[6:8]
        # colour 1's stones are `own` after an even number of turns
        p1, p2 = (own, opp) if t % 2 == 0 else (opp, own)
        res.z = ops.judge(p1, p2)
        res.final_p1, res.final_p2 = p1, p2
        res.n_turns = t
        res.game_turns = None   # (per-game end turns: the one-launch path records them)
        if record:
            for name in ("own", "opp", "pi", "valid", "move"):
                setattr(res, name, getattr(res, name)[:t])
        return res
