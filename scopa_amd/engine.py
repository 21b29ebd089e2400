"""Glue between the reference-shaped Python classes and one HIP context (one deal = one flat tree on device)."""
import numpy as np

from . import _lib


class Engine:
    """A scopa_ctx with the deal of `game` loaded, plus host copies of the (small) infoset metadata."""

    def __init__(self, game, device=0, stream=None):
        perm = getattr(game, "perm", None)
        if perm is None:
            perm = _lib.deal_py_seed(getattr(game, "seed", 42))
        self.game = game
        self.ctx = _lib.Context(device, stream=stream)  # raises ScopaError(ENODEV) without a GPU: no fallback
        self.n_infosets = self.ctx.set_deal(perm)
        t = self.ctx.tree_export()
        self.keys = [_lib.key_to_string(k) for k in t["infoset_key"]]          # information_state_string per id
        self.key_to_id = {k: i for i, k in enumerate(self.keys)}
        self.nlegal = t["infoset_nlegal"].astype(np.int64)
        self.legal = t["infoset_legal"].astype(np.int64)                        # [I][4] action ids, hand order
        self.player = np.array([int(k[1]) for k in self.keys], np.int64)        # "P0:..." / "P1:..."
        self.node_infoset = t["infoset"]                                        # reference DFS order

    def close(self):
        self.ctx.close()

    def visited_order(self):
        """Infoset ids visited so far, in first-visit order (= the reference's dict insertion order)."""
        seq = self.ctx.visited_get()
        ids = np.nonzero(seq)[0]
        return ids[np.argsort(seq[ids], kind="stable")]
