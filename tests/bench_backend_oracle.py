"""TEST INFRASTRUCTURE: the stand-in device layer bench.py loads when DSR_BENCH_TEST_BACKEND names this module
(bench._test_backend): CPU tensors, gloo, and the CPU oracle as the engine — so that bench.py's real command line, its rank
spawning (`--gpus N` without a launcher), the collectives and the JSON line run in the CPU suite.  Never used by the product."""
import torch

DIST_BACKEND = "gloo"


def device(local_rank):
    return torch.device("cpu")


def engine_factory(kinds, calib, local_rank):
    from oracle.oracle import OracleEngine, oracle_settings

    def make_engine(kind):
        kw = dict(kinds[kind])
        # the oracle keeps every array in host RAM and the CPU suite must stay fast: small tables (the CLI test uses 5 cm voxels)
        kw["sdf_local_block_num"] = min(kw["sdf_local_block_num"], 40000)
        kw["hash_bucket_num"] = min(kw["hash_bucket_num"], 0x10000)
        kw["excess_list_size"] = min(kw["excess_list_size"], 0x4000)
        return OracleEngine(oracle_settings(**kw), calib)
    return make_engine


def host_api():
    from oracle.oracle import load_api
    return load_api()
