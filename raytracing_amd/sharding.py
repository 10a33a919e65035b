"""Host logic of the multi-GPU modes (one process per GPU, torch.distributed for the exchange).

Tile mode (one image over N GPUs): the image is cut into 8x8-pixel tiles, tile t = ty*tilesX + tx
goes to rank t % N, and a rank keeps its tiles contiguous (local tile j = global tile j*N + rank,
64 RGBA float pixels per tile). The only exchange is a gather of the equal-sized shards to one
rank, which scatters them back into image order (Renderer.assemble on the GPU; assemble_numpy
below is the same mapping on the host, used by the CPU tests).

Frame mode (a camera path over N GPUs): pose k goes to rank k % N; nothing is exchanged while
tracing, finished 8-bit frames are gathered for writing.
"""
import numpy as np

TILE = 8


def tiles(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def shard_tiles(width, height, world):
    tx, ty = tiles(width, height)
    return (tx * ty + world - 1) // world


def shard_floats(width, height, world):
    return shard_tiles(width, height, world) * 64 * 4


def extract_shard_numpy(image_rgba, rank, world):
    """Row-major (H, W, 4) image -> the flat tile-major shard of `rank` (padding tiles/pixels are 0)."""
    H, W, _ = image_rgba.shape
    tx, ty = tiles(W, H)
    per = shard_tiles(W, H, world)
    out = np.zeros((per, 64, 4), dtype=np.float32)
    padded = np.zeros((ty * TILE, tx * TILE, 4), dtype=np.float32)
    padded[:H, :W] = image_rgba
    t = padded.reshape(ty, TILE, tx, TILE, 4).transpose(0, 2, 1, 3, 4).reshape(ty * tx, 64, 4)   # [tile][lane]
    mine = t[rank::world]
    out[:len(mine)] = mine
    return out.reshape(-1)


def assemble_numpy(gathered, width, height, world):
    """Flat buffer of `world` shards back to back (rank order) -> row-major (H, W, 4) image."""
    tx, ty = tiles(width, height)
    per = shard_tiles(width, height, world)
    g = np.asarray(gathered, dtype=np.float32).reshape(world, per, 64, 4)
    t = np.zeros((tx * ty, 64, 4), dtype=np.float32)
    for r in range(world):
        n = len(t[r::world])
        t[r::world] = g[r, :n]
    img = t.reshape(ty, tx, TILE, TILE, 4).transpose(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, 4)
    return img[:height, :width].copy()


def poses_of_rank(n_poses, rank, world):
    return list(range(rank, n_poses, world))
