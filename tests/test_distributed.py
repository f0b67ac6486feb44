"""Multi-rank parity of the distributed path.

CPU (`not gpu`), world_size 2 over gloo: the partition tables and the exchange plan of
aspect_amd.partition plus the gloo transport of aspect_amd.parallel, checked by running the serial
CPU oracle on each rank's local sub-mesh and compressing the interface sums over gloo.

GPU: N ranks sharing cuda:0, each running the HIP library on its subdomain with the exchange
going through the library's communicator callbacks (tests/dist_worker.py), against the serial
oracle on the global problem."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, script_args, port, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
    assert r.returncode == 0, tail
    return r.stdout


def test_partition_plan_is_consistent():
    """send lists of the owner and ghost groups of the neighbour describe the same nodes."""
    from aspect_amd import cases, partition
    g = cases.nsinker_box(2, n_sinkers=4, ratio=1e4)
    g.reps, g.extents = [2, 2, 1], [2.0, 2.0, 1.0]
    g.build()
    size = 4
    loc = [partition.LocalProblem(g, r, size) for r in range(size)]
    for l in range(g.hier.n_levels):
        levs = [p.hier.levels[l] for p in loc]
        for sp, nn in (("v", g.hier.levels[l].n_nodes_v), ("p", g.hier.levels[l].n_nodes_p)):
            owned = np.concatenate([getattr(lv, sp).l2g[:getattr(lv, sp).n_owned] for lv in levs])
            assert np.array_equal(np.sort(owned), np.arange(nn))       # every node owned exactly once
            for a, la in enumerate(levs):
                so = getattr(la, f"send_offsets_{sp}")
                sn = getattr(la, f"send_nodes_{sp}")
                for k, b in enumerate(la.neighbors):
                    lb = levs[b]
                    kb = lb.neighbors.index(a)                          # symmetric neighbour lists
                    ro = getattr(lb, f"recv_offsets_{sp}")
                    nb_owned = getattr(lb, sp).n_owned
                    mine = getattr(la, sp).l2g[sn[so[k]:so[k + 1]]]
                    theirs = getattr(lb, sp).l2g[nb_owned + ro[kb]:nb_owned + ro[kb + 1]]
                    assert np.array_equal(mine, theirs)


def test_world_size_2_gloo_partition_and_transport():
    out = _launch(2, [os.path.join("tests", "dist_cpu_worker.py")], 29541)
    assert "[dist-cpu] ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case,wire", [(2, "nsinker", "gloo"), (4, "nsinker_harmonic", "gloo"), (8, "box3d", "gloo"),
                                             (2, "deformed", "gloo"), (4, "box2d", "gloo"),
                                             (2, "nsinker_idr", "gloo"), (4, "nsinker_idr", "gloo-dev"),
                                             (2, "free_surface", "gloo"), (8, "free_surface", "gloo-dev"),
                                             # configs[3]: the shell's 96 trees over 8 ranks (12 each) and over 2
                                             (8, "shell", "gloo"), (2, "shell", "gloo-dev"),
                                             # several blocks per rank: the interior ones run under the ghost import
                                             (2, "nsinker_r3", "gloo"), (2, "nsinker_r3", "gloo-dev"), (4, "box3d_r3", "gloo-dev"),
                                             # rank-local set-up: every rank generates only its own subdomain
                                             (2, "nsinker_r3_ranklocal", "gloo-dev"), (8, "box3d_ranklocal", "gloo"),
                                             # configs[4] in small (one r4 cube per rank, per-q viscosity), rank-local set-up
                                             (2, "box3d_r4_ranklocal", "gloo-dev"), (8, "box3d_r4_ranklocal", "gloo"),
                                             # the coarsest levels held whole by every rank (no exchange on them, one sum over ranks in the
                                             # restriction onto the last of them)
                                             (8, "box3d_rep1", "gloo"), (2, "nsinker_r3_rep2", "gloo"), (4, "box3d_r3_rep2", "gloo-dev"),
                                             (4, "box2d_rep2", "gloo"), (2, "box3d_r4_ranklocal_rep3", "gloo-dev"), (2, "nsinker_idr_rep1", "gloo"),
                                             # device-pointer protocol of the communicator (what RCCL uses)
                                             (2, "nsinker_harmonic", "gloo-dev"), (4, "box3d", "gloo-dev")])
def test_distributed_hip_matches_serial_oracle(nproc, case, wire):
    # (the shell's iteration count to the deck's tolerance is not pinned on this geometry, tests/test_oracle_goldens.py:
    # everything up to the preconditioner is compared, then 40 iterations of the solve against the oracle's)
    what = "ops" if case == "shell" else ("vmult" if (case, nproc) == ("box3d_r4_ranklocal", 8) else "all")
    out = _launch(nproc, [os.path.join("tests", "dist_worker.py"), case, wire, what], 29550 + nproc)
    assert "[dist]" in out


@pytest.mark.gpu
def test_transport_device_callbacks_on_a_one_rank_rccl_group():
    """The RCCL leg of aspect_amd.parallel.Transport on the one GPU a test box has: a 1-rank
    "nccl" group; the all-reduce callback gets a raw device pointer and the library-stream handle."""
    out = _launch(1, [os.path.join("tests", "dist_nccl1_worker.py")], 29549)
    assert "[nccl1] ok" in out


@pytest.mark.gpu
def test_distributed_without_overlap_gives_the_same_result(monkeypatch):
    """AGMG_NO_OVERLAP=1: ghost import first, then every block (the order of round 1)."""
    monkeypatch.setenv("AGMG_NO_OVERLAP", "1")
    out = _launch(2, [os.path.join("tests", "dist_worker.py"), "nsinker_r3", "gloo-dev", "all"], 29571)
    assert "[dist]" in out and "0 overlapped" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case,wire", [(2, "box3d_r4_ranklocal", "gloo-dev"), (4, "box3d_r3", "gloo"), (2, "nsinker_r3", "gloo"),
                                             (8, "box3d_r3_rep2", "gloo")])
def test_distributed_brick_levels_match_serial_oracle(nproc, case, wire, monkeypatch):
    """The quad-per-cell brick kernel on the levels of a distributed context (AGMG_BRICKS=1 puts every level that tiles
    into 8^3 cubes on it; by default the levels of 512 cubes or more per rank, i.e. BASELINE configs[4]'s r7 cube per GPU):
    ghost nodes are owned by no brick, their sums travel in the compress step, the fix-up of the owned rows adds the
    received sums.  Operators, V-cycles, preconditioner and solve against the serial oracle as above."""
    monkeypatch.setenv("AGMG_BRICKS", "1")
    out = _launch(nproc, [os.path.join("tests", "dist_worker.py"), case, wire, "all"], 29580 + nproc)
    assert "[dist]" in out and "brick levels" in out


@pytest.mark.gpu
def test_distributed_two_round_mass_smoother_gives_the_same_result(monkeypatch):
    """AGMG_NO_SYM=1: the mass operator of the Schur smoother with ghost import + compress per application (the
    reference's two rounds) instead of the symmetric exchange (one round); both against the serial oracle."""
    monkeypatch.setenv("AGMG_NO_SYM", "1")
    out = _launch(4, [os.path.join("tests", "dist_worker.py"), "box3d_r3_rep2", "gloo-dev", "all"], 29591)
    assert "[dist]" in out
