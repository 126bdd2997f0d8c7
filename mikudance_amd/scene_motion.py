"""Camera + depth -> per-frame 2-channel scene-motion flow at latent resolution (host side, float64 numpy, once per
clip).  Behavioural mirror of reference tools/scene_motion_tracking.py:14-67 as called from
scripts/inference_video.py:185-189 (K = [3.2, 3.2, 1.6, 1.6], istrain=False); pinned by tests/golden/g2_scene_motion.npz.
"""
import numpy as np


def intrinsics(K):
    """3x4 pin-hole projection from [fx, fy, cx, cy]."""
    fx, fy, cx, cy = K
    P = np.zeros((3, 4))
    P[0, 0], P[1, 1], P[0, 2], P[1, 2], P[2, 2] = fx, fy, cx, cy, 1.0
    return P


def camera_to_scene_motion(w2cs, c2ws, K, depth_map, width, height, istrain=True):
    """Returns flow (T, 2, height, width): flow[0] = 0, flow[t] = where the depth-lifted latent grid of frame t-1 lands
    in camera t minus where it was, clipped to mean +- 3 std over the clip; all zeros if anything is non-finite.
    (`istrain` gates the clip on |std| < 10 in the reference, but both of its branches do the same thing.)"""
    T = len(w2cs)
    P = intrinsics(K)
    n = width * height
    col = np.arange(-width // 2, width // 2, 1)
    row = np.arange(-height // 2, height // 2, 1)
    gx, gy = np.meshgrid(col, row)
    grid = np.empty((T, n, 4))
    grid[..., 0] = gx.reshape(1, n)
    grid[..., 1] = gy.reshape(1, n)
    grid[..., 2] = 100 - np.asarray(depth_map, dtype=np.float64).reshape(1, n) * 50
    grid[..., 3] = 1.0
    Pt = np.broadcast_to(P, (T, 3, 4))
    before = np.einsum("tij,taj->tai", Pt, grid)
    before /= before[..., 2:3]
    flow = np.zeros((T, 2, height, width))
    if T < 2:
        return flow
    world = np.einsum("tij,taj->tai", np.stack(c2ws, axis=0), grid)
    moved = np.einsum("tij,taj->tai", np.stack(w2cs, axis=0)[1:], world[:-1])
    after = np.einsum("tij,taj->tai", Pt[1:], moved)
    after /= after[..., 2:3]
    delta = (after[..., :2] - before[:-1, :, :2]).transpose(0, 2, 1).reshape(T - 1, 2, height, width)
    if np.isfinite(delta).all():
        mu, sd = np.mean(delta), np.std(delta)
        flow[1:] = np.clip(delta, mu - 3 * sd, mu + 3 * sd)
    return flow
