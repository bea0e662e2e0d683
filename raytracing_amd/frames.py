"""Frame-parallel sharding of a batch of independent frames over ranks (one process per
GPU).  Frames share nothing -- G_clear resets all state (reference GraphicsBase.cpp:99-111)
-- so rank r renders frames r, r+world, ... and the only exchange is a gather of per-frame
digests (or of the frames themselves) when a caller wants them in one place.  There is no
collective on the data path.
"""
from typing import Callable, Dict, Iterable, Tuple

from . import scenes as S
from .transform import make_matrix


def frames_for_rank(n_frames: int, rank: int, world: int) -> range:
    return range(rank, n_frames, world)


def frame_matrix(f: int, translation=(0, 0, 2048)):
    """Animated-batch placement of frame f (configuration 4): object angles
    (f, 2f, 3f) & 255, fixed translation, identity camera."""
    return make_matrix((0, 0, 0, 0, 0, 0), (f & 255, (2 * f) & 255, (3 * f) & 255) + tuple(translation))


def render_frames(render: Callable[[int], Tuple], frame_ids: Iterable[int]) -> Dict[int, Tuple[str, str]]:
    """render(f) -> (colour int32[H][W], depth int32[H][W]); returns {f: (digest_c, digest_z)}."""
    out = {}
    for f in frame_ids:
        c, z = render(f)
        out[f] = (S.frame_digest(c), S.frame_digest(z))
    return out


def gather_digests(local: Dict[int, Tuple[str, str]], world: int) -> Dict[int, Tuple[str, str]]:
    """All ranks receive every frame's digests (torch.distributed must be initialised when world > 1)."""
    if world == 1:
        return dict(local)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
