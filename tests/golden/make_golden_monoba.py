#!/usr/bin/env python3
"""tests/golden/monoba.npz: the reference's own bundle-adjustment acceptance test (TestMonoBA,
A/15-vio-backend/app/TestMonoBA.cpp; its printout is published in A/15-vio-backend/README.md:21-73) as a fixture.

Inputs.  TestMonoBA draws its scene from std::default_random_engine with libstdc++'s distributions: three cameras on a
quarter arc (radius 8, noisy initial poses), 20 points, every point seen from every camera with 1e-3 of noise, the initial
inverse depths from 1 / (z + N(0, 1)).  The small C++ program below (ours: it states the same draws in the same order, with
<random> and nothing else) reproduces the scene of the source as it is in the tree.

Expected outputs.  The program itself cannot be run here (A/15's problem.cc includes glog, which this image lacks), so what
the reference gives as its result is its README's table: per landmark `ground truth`, `with noise`, `opt` (4 decimals) and
the camera translations after optimisation, with the verdict "the estimation has converged to ground truth".  That table
was printed by an earlier revision of the generator (its points come from a different place in the random stream: the rows
from the fifth on are this generator's points from the third on), so it pins two things only: the depth-noise draws of
main() — a fresh engine, N(0,1): 1/noisy - 1/gt of the README's rows are this program's draws, up to the 0.05 of camera 0's
height noise — and the acceptance level: |opt - gt| <= 3e-4 in inverse depth, camera translations equal to the ground truth
to the printed digit, on observations without noise.  Both are stored with the scene.

  python tests/golden/make_golden_monoba.py        (needs g++ and /root/reference)
"""
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
README = "/root/reference/workspace/assignments/15-vio-backend/README.md"

GEN = r"""
#include <cmath>
#include <cstdio>
#include <random>
int main() {
    const int NP = 3, NF = 20;
    const double radius = 8;
    std::default_random_engine gen;
    double th_gt[NP], th_obs[NP], t_gt[NP][3], t_obs[NP][3];
    for (int n = 0; n < NP; ++n) {
        std::uniform_real_distribution<double> xyz(-0.05, +0.05), th(-0.105, +0.105);
        th_gt[n] = n * 2 * M_PI / (NP * 4);
        th_obs[n] = th_gt[n] + th(gen);
        t_gt[n][0] = radius * std::cos(th_gt[n]) - radius; t_gt[n][1] = radius * std::sin(th_gt[n]); t_gt[n][2] = std::sin(2 * th_gt[n]);
        for (int k = 0; k < 3; ++k) t_obs[n][k] = t_gt[n][k] + xyz(gen);
        std::printf("cam %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", th_gt[n], t_gt[n][0], t_gt[n][1], t_gt[n][2], th_obs[n], t_obs[n][0], t_obs[n][1], t_obs[n][2]);
    }
    std::normal_distribution<double> pix(0., 1. / 1000.);
    double pw[NF][3];
    for (int j = 0; j < NF; ++j) {
        std::uniform_real_distribution<double> xy(-4, 4.0), z(4., 8.);
        // (the reference passes the three draws as constructor arguments: g++ evaluates them right to left)
        pw[j][2] = z(gen); pw[j][1] = xy(gen); pw[j][0] = xy(gen);
        std::printf("pt %.17g %.17g %.17g", pw[j][0], pw[j][1], pw[j][2]);
        for (int i = 0; i < NP; ++i) {
            // p_c = R_gt^T (p_w - t_gt), R_gt a rotation about z
            const double c = std::cos(th_gt[i]), s = std::sin(th_gt[i]);
            const double d0 = pw[j][0] - t_gt[i][0], d1 = pw[j][1] - t_gt[i][1], d2 = pw[j][2] - t_gt[i][2];
            const double x = c * d0 + s * d1, y = -s * d0 + c * d1, zc = d2;
            double u = x / zc, v = y / zc;
            u += pix(gen);
            v += pix(gen);
            std::printf(" %.17g %.17g", u, v);
        }
        std::printf("\n");
    }
    // main(): a fresh engine; the initial inverse depth of point j from the noisy camera 0
    std::default_random_engine gen2;
    std::normal_distribution<double> dn(0, 1.);
    for (int j = 0; j < NF; ++j) {
        const double c = std::cos(th_obs[0]), s = std::sin(th_obs[0]);
        const double d0 = pw[j][0] - t_obs[0][0], d1 = pw[j][1] - t_obs[0][1], d2 = pw[j][2] - t_obs[0][2];
        (void)c; (void)s; (void)d0; (void)d1;
        const double noise = dn(gen2);
        std::printf("invd %.17g\n", 1. / (d2 + noise));
    }
    return 0;
}
"""


def main():
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "gen.cpp"), os.path.join(tmp, "gen")
        open(src, "w").write(GEN)
        subprocess.check_call(["g++", "-O0", "-o", exe, src])
        lines = subprocess.check_output([exe], text=True).splitlines()
    cams = np.array([[float(x) for x in l.split()[1:]] for l in lines if l.startswith("cam")])
    pts = np.array([[float(x) for x in l.split()[1:]] for l in lines if l.startswith("pt")])
    invd0 = np.array([float(l.split()[1]) for l in lines if l.startswith("invd")])
    text = open(README).read()
    rows = re.findall(r"ground truth :([\d.]+)\s+with noise :([\d.]+)\s+opt ([\d.]+)", text)
    table = np.array([[float(a), float(b), float(c)] for a, b, c in rows[:20]])
    cam_opt = np.array([[float(x) for x in m] for m in re.findall(r"optimized:\s+(-?[\d.]+)\s+(-?[\d.]+)\s+(-?[\d.]+)", text)[:3]])
    assert table.shape == (20, 3) and cam_opt.shape == (3, 3)
    # the depth-noise draws of main() are the README's: z_noisy - z_gt per row (camera 0's height noise is within 0.05)
    mine = 1.0 / invd0 - pts[:, 2]
    theirs = 1.0 / table[:, 1] - 1.0 / table[:, 0]
    assert np.abs(mine - theirs).max() <= 0.06, np.abs(mine - theirs).max()
    # ... and from the fifth row on the README's points are this generator's from the third on
    assert np.abs(1.0 / pts[2:18, 2] - table[4:20, 0]).max() <= 5.1e-5
    np.savez(os.path.join(HERE, "monoba.npz"), theta_gt=cams[:, 0], t_gt=cams[:, 1:4], theta_obs=cams[:, 4], t_obs=cams[:, 5:8],
             points=pts[:, 0:3], obs=pts[:, 3:].reshape(20, 3, 2), inv_depth_init=invd0,
             readme_inv_depth_gt=table[:, 0], readme_inv_depth_noisy=table[:, 1], readme_inv_depth_opt=table[:, 2], readme_cam_t_opt=cam_opt)
    print("monoba.npz written; depth-noise draws against the README: max difference %.3f" % np.abs(mine - theirs).max())


if __name__ == "__main__":
    main()
