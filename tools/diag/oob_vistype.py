"""diagnostic: where do the visible types differ after frames that exhaust the block array?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.common import make_pair
from dynslam_amd.engine import OutOfBlocksError
for env in ({}, {"DSR_NO_PUBLISHED_STATUS": "1"}):
    os.environ.update(env)
    for do_prepare in (False, True):
        sc, g, o = make_pair(sdf_local_block_num=1500)
        for i in range(4):
            rgba, d, T, _ = sc.frame(i)
            for e in (g, o):
                e.update_view(rgba, d); e.set_pose_inv_m(T)
                try:
                    e.process_frame()
                except OutOfBlocksError:
                    pass
                if do_prepare:
                    e.prepare()
            a, b = g.dump_visible_types(), o.dump_visible_types()
            bad = np.nonzero(a != b)[0]
            vg, vo = g.dump_visible_list(), o.dump_visible_list()
            print(env, "prepare", do_prepare, "frame", i, "visible", len(vg), len(vo), "list equal", np.array_equal(vg, vo), "types differ at", len(bad), flush=True)
            if len(bad):
                ht = g.dump_hash_table(); ho = o.dump_hash_table()
                for t in bad[:6]:
                    print("   entry", int(t), "hip", int(a[t]), "oracle", int(b[t]), "in list", bool((vg == t).any()), "hip entry", ht[t], "oracle entry", ho[t], flush=True)
        g.close(); o.close()
