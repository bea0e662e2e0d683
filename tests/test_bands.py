"""Sort-first row bands of one frame: the band assignment, and the gather of finished
bands to rank 0 with world sizes 2 and 3 over gloo on CPU.  The renderer plugged in here is
the CPU checker (test infrastructure); on GPUs the same gather is fed by the HIP path
(tests/test_gpu_bands.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracing_amd import bands as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("height", [1, 31, 32, 33, 480, 1080, 2160, 16384])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8, 64])
def test_bands_partition_the_rows(height, world):
    bands = B.all_bands(world, height)
    rows = [y for y0, y1 in bands for y in range(y0, y1)]
    assert rows == list(range(height))                     # every row once, in rank order
    for y0, y1 in bands:
        assert y0 == y1 == height or (y0 % 32 == 0 and (y1 % 32 == 0 or y1 == height))
    sizes = [((y1 - y0) + 31) // 32 for y0, y1 in bands]
    assert max(sizes) - min(sizes) <= 1


def test_band_alignment_argument():
    assert B.all_bands(3, 10, align=1) == [(0, 4), (4, 7), (7, 10)]
    assert B.all_bands(4, 64, align=32) == [(0, 32), (32, 64), (64, 64), (64, 64)]


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import torch.distributed as dist
import oracle.port as P
from raytracing_amd import bands as B, scenes as S
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H = 160, 120
sc = S.gen_c2(800, W, H, 7, radius=12, seed=21)
orc = P.Oracle(W, H)
orc.clear(0x203040, 255, 255, 255)
orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
y0, y1 = B.band_rows(rank, world, H)
c = torch.full((H, W), -1, dtype=torch.int32)
z = torch.full((H, W), -1, dtype=torch.int32)
c[y0:y1] = torch.from_numpy(orc.cbuf()[y0:y1].copy())      # this rank owns only its band
z[y0:y1] = torch.from_numpy(orc.zbuf()[y0:y1].copy())
B.gather_bands([c, z], rank, world, H)
if rank == 0:
    np.savez({out!r}, c=c.numpy(), z=z.numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_gather_bands_gloo(tmp_path, world):
    import oracle.port as P
    from raytracing_amd import scenes as S
    out = str(tmp_path / "frame.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    got = np.load(out)
    sc = S.gen_c2(800, 160, 120, 7, radius=12, seed=21)
    orc = P.Oracle(160, 120)
    orc.clear(0x203040, 255, 255, 255)
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    assert (got["c"] == orc.cbuf()).all() and (got["z"] == orc.zbuf()).all()


# ---- routed set-up: which polygons go to which band, and the exchange of the index lists

@pytest.mark.parametrize("n,world", [(0, 3), (1, 4), (10, 3), (1000, 8), (1_000_003, 7)])
def test_shares_partition_the_polygons(n, world):
    parts = [B.share(r, world, n) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == n
    for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [hi - lo for lo, hi in parts]
    assert max(sizes) - min(sizes) <= 1


def _random_rows(n, height, seed):
    """{first, last} per polygon: mostly short runs of rows, some empty ({1, 0}), some the whole screen"""
    rng = np.random.default_rng(seed)
    lo = rng.integers(0, height, n)
    hi = np.minimum(lo + rng.integers(0, 40, n), height - 1)
    rows = np.stack([lo, hi], axis=1).astype(np.int32)
    kind = rng.integers(0, 10, n)
    rows[kind == 0] = (1, 0)
    rows[kind == 1] = (0, height - 1)
    return rows


def _expected_lists(rows, world, height):
    out = []
    for y0, y1 in B.all_bands(world, height):
        out.append(np.array([i for i, (lo, hi) in enumerate(rows) if lo <= hi and lo < y1 and hi >= y0 and y0 < y1], np.int32))
    return out


@pytest.mark.parametrize("world", [1, 2, 5])
def test_lists_by_band(world):
    import torch
    height, n = 480, 3000
    rows = _random_rows(n, height, 5)
    want = _expected_lists(rows, world, height)
    got = [np.zeros(0, np.int32) for _ in range(world)]
    for k in range(world):                       # every rank its share; lists joined in rank order
        lo, hi = B.share(k, world, n)
        for r, l in enumerate(B.lists_by_band(torch.from_numpy(rows[lo:hi]), lo, world, height)):
            assert l.dtype == torch.int32
            got[r] = np.concatenate([got[r], l.numpy()])
    for r in range(world):
        assert np.array_equal(got[r], want[r])
        assert np.all(np.diff(got[r]) > 0)       # ascending: draw order kept


ROUTE_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import torch.distributed as dist
from raytracing_amd import bands as B
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rows = np.load({rows!r})
lo, hi = B.share(rank, world, rows.shape[0])
lists = B.lists_by_band(torch.from_numpy(rows[lo:hi]), lo, world, {height})
mine = B.exchange_lists(lists, rank, world)
np.save({out!r} + str(rank) + ".npy", mine.numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_lists_gloo(tmp_path, world):
    height, n = 1080, 5000
    rows = _random_rows(n, height, 11 + world)
    if world == 3:
        rows[: n // 3] = (1, 0)                 # a rank with nothing to send
    np.save(tmp_path / "rows.npy", rows)
    script = tmp_path / "worker.py"
    script.write_text(ROUTE_WORKER.format(root=ROOT, rows=str(tmp_path / "rows.npy"), out=str(tmp_path / "mine"), height=height))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29561 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    want = _expected_lists(rows, world, height)
    for r in range(world):
        got = np.load(str(tmp_path / "mine") + f"{r}.npy")
        assert got.dtype == np.int32 and np.array_equal(got, want[r])
