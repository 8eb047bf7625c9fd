"""TEST INFRASTRUCTURE (GPU box): the device layer bench.py loads when DSR_BENCH_TEST_BACKEND names this module — the REAL HIP
engines, every rank on cuda:0, gloo for the collectives.  A one-GPU box cannot host an RCCL group of two ranks (RCCL refuses two
ranks on one device), but it can host two PROCESSES that each drive their own HIP engines, device-resident masks, renders into
the exchange slots and the HIP composite: everything of the N > 1 path except the RCCL call itself (which runs at world size 1
under torchrun).  Never used by the product."""
import torch

DIST_BACKEND = "gloo"


def device(local_rank):
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def engine_factory(kinds, calib, local_rank):
    from dynslam_amd.engine import EngineCore, default_settings

    def make_engine(kind):
        kw = dict(kinds[kind])
        kw["sdf_local_block_num"] = min(kw["sdf_local_block_num"], 1 << 19)  # two ranks share one GPU's memory
        return EngineCore(default_settings(**kw, device=0, sync_status=0), calib)
    return make_engine


def host_api():
    return None
