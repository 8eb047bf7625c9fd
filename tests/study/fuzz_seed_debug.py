"""Investigation aid for tests/test_gpu_fuzz.py: replay one seed's call sequence on the HIP engine and the oracle and, at the first
call after which the visible types differ, print what is known about the entries that differ (GPU box).
usage: python tests/study/fuzz_seed_debug.py <seed>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_fuzz as F  # noqa: E402
from tests.common import make_pair  # noqa: E402


def main(seed):
    from dynslam_amd.engine import OutOfBlocksError
    rng = np.random.default_rng(1000 + seed)
    W, H, kw, kind = F._draw_settings(rng)
    sc, g, o = make_pair(W=W, H=H, scene_kw=dict(noise_px=float(rng.choice([0.0, 0.4, 0.8]))), **kw)
    frame, fed = int(rng.integers(0, 4)), 0
    hist = []
    for step in range(int(rng.integers(10, 18))):
        op = rng.choice(["frame", "frame", "frame", "decay", "render", "reset"], p=[0.3, 0.2, 0.15, 0.15, 0.15, 0.05])
        if fed == 0:
            op = "frame"
        before = (g.dump_visible_types().copy(), g.dump_visible_list().copy(), g.dump_hash_table().copy(), g.get_stats())
        if op == "frame":
            frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 5, -3])))
            prepare = bool(rng.random() < 0.8)
            rgba, d, T, _ = sc.frame(frame)
            if rng.random() < 0.15:
                d = d.copy(); d[:, : W // 3] = 0
            for e in (g, o):
                e.update_view(rgba, d); e.set_pose_inv_m(T)
                try:
                    e.process_frame()
                except OutOfBlocksError:
                    print("  out of blocks", type(e).__name__)
                if prepare:
                    e.prepare()
            fed += 1
            what = ("frame", frame, prepare)
            if prepare:
                rg, ro = g.dump_render_state(False), o.dump_render_state(False)
                for k in ("minmax", "raycast_result", "raycast_image", "points", "normals"):
                    if not np.array_equal(rg[k], ro[k]):
                        print("  render state", k, "differs at", int((rg[k] != ro[k]).sum()), "elements; visible blocks", g.get_stats().no_visible_blocks, o.get_stats().no_visible_blocks)
        elif op == "decay":
            args = (int(rng.choice([1, 2, 5, 100])), int(rng.choice([0, 0, 1, 3])), bool(rng.random() < 0.25))
            for e in (g, o):
                e.decay(*args)
            what = ("decay",) + args
        elif op == "render":
            rng.integers(-2, 3); rng.normal(0, 0.05, 3); rng.integers(len(F.RENDER_TYPES))
            what = ("render (skipped)",)
        else:
            for e in (g, o):
                e.reset_scene()
            fed = 0
            what = ("reset",)
        tg, to = g.dump_visible_types(), o.dump_visible_types()
        lg, lo = g.dump_visible_list(), o.dump_visible_list()
        hg, ho = g.dump_hash_table(), o.dump_hash_table()
        sg, so = g.get_stats(), o.get_stats()
        print(step, what, "visible", len(lg), len(lo), "free block head", sg.last_free_block_id, so.last_free_block_id,
              "types differ", int((tg != to).sum()), "table differs", int((hg != ho).sum()), "lists equal", np.array_equal(lg, lo))
        if (tg != to).any():
            bad = np.nonzero(tg != to)[0]
            print("  entries", bad[:12], "of", len(bad))
            print("  type here / there", tg[bad][:12], to[bad][:12])
            print("  type before this call (here)", before[0][bad][:12])
            print("  in the visible list before this call", np.isin(bad, before[1])[:12], "after", np.isin(bad, lg)[:12])
            print("  ptr now", hg["ptr"][bad][:12], "ptr before", before[2]["ptr"][bad][:12], "offset", hg["offset"][bad][:12])
            print("  visible before", len(before[1]), "decayed", sg.decayed_block_count, so.decayed_block_count)
            break
    g.close(); o.close()


if __name__ == "__main__":
    main(int(sys.argv[1]))
