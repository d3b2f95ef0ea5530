"""Multi-process test of the sharded path on CPU: world_size 2, gloo backend.  The GPU law is
replaced by an oracle-backed stand-in (tests may use the oracle); what is under test is the
shard plan, the in-place gather layout and that sharded == unsharded."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_oracle as O

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


class OracleLaw:
    """Duck-typed IncrSmallStrainModel on CPU torch tensors (shares memory with NumPy)."""

    def evaluate(self, t, del_t, grad, stress, tangent, history):
        h = None if history is None else {k: v.numpy() for k, v in history.items()}
        O.von_mises_3d(VM_P, t, del_t, grad.numpy(), stress.numpy(), tangent.numpy(), h)


def make_inputs(n):
    rng = np.random.default_rng(42)
    scale = np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    return (rng.normal(size=9 * n) * scale, rng.normal(scale=30.0, size=6 * n),
            {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)})


def worker(rank, world, port, n, out_dir, direct=False):
    from fenics_constitutive_amd.sharded import ShardedEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, s, h = make_inputs(n)
        ev = ShardedEvaluator(OracleLaw(), n)
        per = ev.plan.per_rank
        # gathered buffers; this rank's slice starts as the committed stress (in-place semantics)
        sg = torch.zeros(6 * per * world, dtype=torch.float64)
        tg = torch.zeros(36 * per * world, dtype=torch.float64)
        sg[6 * per * rank : 6 * per * rank + 6 * ev.n_local] = torch.from_numpy(ev.local_view(s, 6).copy())
        hl = {"eps_n": torch.from_numpy(ev.local_view(h["eps_n"], 6).copy()),
              "alpha": torch.from_numpy(ev.local_view(h["alpha"], 1).copy())}
        gl = torch.from_numpy(ev.local_view(g, 9).copy())
        s_all, t_all = ev.evaluate_and_gather(0.0, 1.0, gl, sg, tg, hl, direct=direct)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), stress=s_all.numpy(), tangent=t_all.numpy(),
                 alpha=hl["alpha"].numpy(), lo=ev.lo, hi=ev.hi)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,direct", [(2, False), (2, True), (3, True)])
@pytest.mark.parametrize("n", [1000, 128, 65])
def test_sharded_equals_unsharded(n, world, direct, tmp_path):
    mp.spawn(worker, args=(world, free_port(), n, str(tmp_path), direct), nprocs=world, join=True)
    g, s, h = make_inputs(n)
    t = np.zeros(36 * n)
    O.von_mises_3d(VM_P, 0.0, 1.0, g, s, t, h)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["stress"], s), f"rank {r} gathered stress"
        assert np.array_equal(z["tangent"], t), f"rank {r} gathered tangent"
        assert np.array_equal(z["alpha"], h["alpha"][int(z["lo"]) : int(z["hi"])])  # history stays sharded
