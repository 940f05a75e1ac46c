"""Loader for tests/golden/*.npz (written by oracle/gen_golden.py from the reference build)."""
import glob
import json
import os

import numpy as np

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    d = meta["param"]
    prm = orc.Param(**{k: (np.float32(v) if k.startswith("pen_") else int(v)) for k, v in d.items()})
    return dict(name=os.path.basename(path)[:-4], a=z["a"], f=z["f"], p=z["p"].astype(np.int64), u=z["u"], a_out=z["a_out"],
                prm=prm, meta=meta)


def all_cases(prefix=""):
    return sorted(glob.glob(os.path.join(GOLD, prefix + "*.npz")))


def case_ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


def load_rmq(path):
    """tests/golden/rmq/*.npz (oracle/gen_golden_rmq.py): the reference's mg_lchain_rmq on one read."""
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    d = meta["param"]
    prm = orc.RmqParam(**{k: (np.float32(v) if k.startswith("pen_") else int(v)) for k, v in d.items()})
    return dict(name=os.path.basename(path)[:-4], a=z["a"], f=z["f"], p=z["p"].astype(np.int64), u=z["u"], a_out=z["a_out"], prm=prm, tied=meta["tied"])


def rmq_cases():
    return sorted(glob.glob(os.path.join(GOLD, "rmq", "*.npz")))


def load_seeds(path):
    """tests/golden/seeds/*.npz (oracle/gen_golden_seeds.py): what the reference's collect_seed_hits was given and returned."""
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return dict(name=os.path.basename(path)[:-4], seeds=z["seeds"], hits=z["hits"], hit_off=z["hit_off"], a=z["a"], flag=int(meta["flag"]),
                qlen=int(meta["qlen"]), q_rank=int(meta.get("q_rank", 0)), ref_rank=meta.get("ref_rank"), ref_len=meta.get("ref_len"),
                rep_len=int(meta["rep_len"]), mini_pos=z["mini_pos"], source=meta["source"], record=int(meta["record"]))


def seed_cases():
    return sorted(glob.glob(os.path.join(GOLD, "seeds", "*.npz")))
