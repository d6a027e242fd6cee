#!/usr/bin/env python3
"""Keypoint matching -> triangulation on one MI355X with spectavi_amd, the device-side
counterpart of steps 2-4 of the reference's example/ex01_essential_estimation.py
(match keypoints :89-106, estimate/score cameras :138-160, triangulate :171-173).

Input: two SIFT tables in the reference's format (float32 [n,132] = x, y, sigma, angle +
128 descriptor values, src/Sift.h:13,115-123).  With no arguments the script builds a
synthetic second view from the reference's golden table stored in tests/golden.

    python examples/match_pipeline.py [table0.npy table1.npy]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spectavi_amd import device  # noqa: E402


def synthetic_pair():
    table = np.load(os.path.join(ROOT, "tests", "golden", "sift_sur_ogre_table.npz"))["table"]
    rng = np.random.default_rng(0)
    n = table.shape[0]
    # a scene: 3-D points seen by two cameras; keypoint (x, y) = projections, descriptors shared
    Xw = np.hstack([rng.uniform(-1, 1, (n, 2)), rng.uniform(4, 8, (n, 1)), np.ones((n, 1))])
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    a = 0.1
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    P1 = np.hstack([R, np.array([[-0.5], [0.02], [0.05]])])
    t0, t1 = table.copy(), table[rng.permutation(n)].copy()
    x0 = Xw @ P0.T
    t0[:, :2] = (x0[:, :2] / x0[:, 2:3]).astype(np.float32)
    perm = rng.permutation(n)
    t1 = t0[perm].copy()
    x1 = Xw[perm] @ P1.T
    t1[:, :2] = (x1[:, :2] / x1[:, 2:3]).astype(np.float32)
    t1[:, 4:] = np.clip(t1[:, 4:] + rng.integers(-3, 4, (n, 128)), 0, 255)
    return t0, t1, P0, P1


def main():
    if len(sys.argv) == 3:
        t0, t1 = np.load(sys.argv[1]).astype(np.float32), np.load(sys.argv[2]).astype(np.float32)
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        P1 = None
    else:
        t0, t1, P0, P1 = synthetic_pair()
    dev = torch.device("cuda")
    g0, d0 = device.split_sift_table(torch.from_numpy(t0).to(dev))      # database image
    g1, d1 = device.split_sift_table(torch.from_numpy(t1).to(dev))      # query image
    idx, dist = device.l1k2(d0, d1)                                     # exact L1 2-NN
    matches, count = device.ratio_test(idx, dist, 1.75)                 # ex01's default min_ratio
    x0, x1 = device.match_coordinates(g0, g1, matches, count)
    n = int(count.item())
    print("keypoints %d / %d -> %d matches after the ratio test" % (t0.shape[0], t1.shape[0], n))
    if P1 is None:
        return
    # score a few candidate cameras (the true one + perturbed ones) the way RANSAC does, then
    # triangulate the inliers of the winner
    rng = np.random.default_rng(1)
    cands = np.stack([P1] + [P1 + rng.normal(0, 0.05, (3, 4)) for _ in range(7)])
    counts, mask = device.dlt_score_hypotheses(P0, torch.from_numpy(cands).to(dev), x0[:n].contiguous(),
                                               x1[:n].contiguous(), 1e-3, want_mask=True)
    best = int(counts.argmax().item())
    inl = mask[best].bool()
    X = device.dlt_triangulate(P0, cands[best], x0[:n][inl].contiguous(), x1[:n][inl].contiguous())
    torch.cuda.synchronize()
    print("inlier counts per candidate camera:", counts.cpu().numpy().tolist(), "-> best", best)
    Xe = (X[:, :3] / X[:, 3:4]).cpu().numpy()
    print("triangulated %d points, depth range %.2f .. %.2f" % (len(Xe), Xe[:, 2].min(), Xe[:, 2].max()))


if __name__ == "__main__":
    main()
